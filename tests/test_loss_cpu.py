"""Pin the loss restatement (oracle/loss_oracle.py) to the reference's own training/loss_fns.py outputs and gradients
(tests/golden/losses.npz, written by oracle/make_loss_goldens.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

CONFIGS = {
    "default": dict(weight_dict={"loss_mask": 20, "loss_dice": 1, "loss_iou": 1, "loss_class": 1}),
    "repo": dict(weight_dict={"loss_mask": 20, "loss_dice": 1, "loss_iou": 1, "loss_class": 1}, supervise_all_iou=True,
                 iou_use_l1_loss=True, pred_obj_scores=True, focal_gamma_obj_score=0.0, focal_alpha_obj_score=-1.0),
}
KEYS = ("multistep_pred_multimasks_high_res", "multistep_pred_ious", "multistep_object_score_logits")


def golden_case(g, device="cpu"):
    """(outs with leaf tensors, targets, the leaves in the golden's gradient order)."""
    targets = torch.from_numpy(g["targets"]).to(device)
    outs, leaves = [], []
    for ti in range(targets.shape[0]):
        d = {}
        for k in KEYS:
            d[k] = []
            si = 0
            while f"in.{ti}.{k}.{si}" in g:
                d[k].append(torch.from_numpy(g[f"in.{ti}.{k}.{si}"]).to(device).requires_grad_())
                si += 1
            leaves += d[k]
        outs.append(d)
    return outs, targets, leaves


@pytest.mark.parametrize("name", ["default", "repo"])
def test_loss_restatement_matches_reference(name):
    from oracle import loss_oracle as LO
    g = load_golden("losses")
    outs, targets, leaves = golden_case(g)
    losses = LO.multistep_losses(outs, targets, **CONFIGS[name])
    for k in ("loss_mask", "loss_dice", "loss_iou", "loss_class", "core_loss"):
        assert abs(float(losses[k]) - float(g[f"{name}.{k}"])) <= 1e-6 * max(1.0, abs(float(g[f"{name}.{k}"]))), k
    losses["core_loss"].backward()
    for i, x in enumerate(leaves):
        got = x.grad.numpy() if x.grad is not None else np.zeros_like(g[f"{name}.grad.{i}"])
        assert np.allclose(got, g[f"{name}.grad.{i}"], rtol=1e-5, atol=1e-9), i
    assert float(g["repo.loss_class"]) > 0 and float(g["default.loss_class"]) == 0
