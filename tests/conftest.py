import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLD = os.path.join(REPO, "tests", "golden")

TINY = dict(embed_dim=56, num_heads=1, stages=(2, 2, 3, 3), global_att_blocks=(5,))
BPLUS = dict(embed_dim=112, num_heads=2)
# Hiera-L (configs/sam2.1/sam2.1_hiera_l.yaml:11-16: head_dim 72, window_spec (8,4,16,8)) and a 10-block miniature of it
LARGE = dict(embed_dim=144, num_heads=2, stages=(2, 6, 36, 4), global_att_blocks=(23, 33, 43), window_spec=(8, 4, 16, 8))
TINY_L = dict(embed_dim=72, num_heads=1, stages=(2, 2, 3, 3), global_att_blocks=(5,), window_spec=(8, 4, 16, 8))
# miniature of the reference's small model (sam2_model.py:92-99, stages [1,2,11,2]): blocks[-2:] holds the stage-4 entry block
TINY_S = dict(embed_dim=56, num_heads=1, stages=(1, 2, 3, 2), global_att_blocks=(4,))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.device_count() > 0:     # counting devices does not initialise the GPU in this process (is_available() does)
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def rel_err(a, b) -> float:
    """||a-b||_2 / ||b||_2 over the whole tensor (SURVEY.md section 8c metric)."""
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).double().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def load_golden(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))
