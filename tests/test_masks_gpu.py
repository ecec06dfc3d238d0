"""North-star mask criterion: masks predicted from the HIP encoder's outputs reach IoU >= 0.99 against the
REFERENCE's masks (tests/golden/masks_bplus_r4.npz, produced by the reference's own SAM2Base on the same
synthetic frames and weights).  The SAM heads are not part of the hot path: they run as the CPU
restatement oracle/sam_heads.py (pinned to the reference in tests/test_masks_cpu.py) on the encoder
outputs copied back from the GPU."""
import pytest
import torch

from conftest import BPLUS
from test_masks_cpu import golden_masks
from test_trunk_gpu import make_encoder
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,bar", [(torch.float32, 0.999), (torch.bfloat16, 0.99)])
def test_mask_iou_vs_reference(dtype, bar):
    from image_segmentation_amd.synth import synth_frames
    from oracle import sam_heads as SH
    g = load_golden("masks_bplus_r4")
    n = g["coords"].shape[0]
    enc = make_encoder(BPLUS)
    enc.trunk.compute_dtype = dtype
    with torch.no_grad():
        fpn = enc(synth_frames(n, 1024, 0, device="cuda"))["backbone_fpn"]
    out = SH.forward_sam_heads(SH.make_head_params(0), fpn, torch.from_numpy(g["coords"]), torch.from_numpy(g["labels"]))
    assert (out["best"].numpy() == g["best"]).all()
    assert (out["object_score_logits"] > 0).all()
    hi, lo = golden_masks(g, "high_bits", 1024), golden_masks(g, "low_bits", 256)
    ious = [SH.mask_iou(out["high_res_masks"][i] > 0, hi[i]) for i in range(n)]
    ious_lo = [SH.mask_iou(out["low_res_masks"][i] > 0, lo[i]) for i in range(n)]
    print(f"mask IoU vs reference ({dtype}): high-res {ious}, low-res {ious_lo}")
    assert min(ious) >= bar and min(ious_lo) >= bar, (ious, ious_lo)
