"""The product losses (image_segmentation_amd/loss_fns.py: lsam_mask_loss_sums / _bwd + [N, M] arithmetic) against the
REFERENCE's own outputs and gradients (tests/golden/losses.npz): the five losses to 1e-5 relative, the gradient of the
core loss with respect to every prediction to 1e-4 (L2, per tensor)."""
import pytest
import torch

from conftest import load_golden, rel_err
from test_loss_cpu import CONFIGS, golden_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["default", "repo"])
def test_losses_vs_reference(name):
    from image_segmentation_amd.loss_fns import CORE_LOSS_KEY, MultiStepMultiMasksAndIous
    g = load_golden("losses")
    outs, targets, leaves = golden_case(g, "cuda")
    cfg = dict(CONFIGS[name])
    cfg["weight_dict"] = dict(cfg["weight_dict"])
    losses = MultiStepMultiMasksAndIous(**cfg)(outs, targets)
    for k in ("loss_mask", "loss_dice", "loss_iou", "loss_class", CORE_LOSS_KEY):
        ref = float(g[f"{name}.{k}"])
        assert abs(float(losses[k]) - ref) <= 1e-5 * max(1.0, abs(ref)), (k, float(losses[k]), ref)
    losses[CORE_LOSS_KEY].backward()
    for i, x in enumerate(leaves):
        ref = torch.from_numpy(g[f"{name}.grad.{i}"])
        got = x.grad.cpu() if x.grad is not None else torch.zeros_like(ref)
        if float(ref.abs().max()) == 0:
            assert float(got.abs().max()) == 0, i
        else:
            assert rel_err(got, ref) < 1e-4, (i, rel_err(got, ref))


def test_loss_functions_standalone_and_errors():
    from image_segmentation_amd import loss_fns as LF
    from image_segmentation_amd._C import LsamError
    from oracle import loss_oracle as LO
    g = torch.Generator().manual_seed(3)
    x = 3 * torch.randn(4, 3, 32, 32, generator=g)
    t1 = (torch.randn(4, 1, 32, 32, generator=g) > 0).float()
    t, tc = t1.expand(4, 3, 32, 32), t1.cuda().expand(4, 3, 32, 32)      # ONE target expanded over the candidates (loss_fns.py:240)
    pi = torch.rand(4, 3, generator=g)
    for mm in (True, False):
        if mm:
            a = LF.dice_loss(x.cuda(), tc, 2.0, True)
            b = LO.dice_loss(x, t, 2.0, True)
            assert rel_err(a, b) < 1e-5
            assert rel_err(LF.sigmoid_focal_loss(x.cuda(), tc, 2.0, loss_on_multimask=True), LO.sigmoid_focal_loss(x, t, 2.0, loss_on_multimask=True)) < 1e-5
            assert rel_err(LF.iou_loss(x.cuda(), tc, pi.cuda(), 2.0, True, True), LO.iou_loss(x, t, pi, 2.0, True, True)) < 1e-5
    o, to = torch.randn(5, 1, generator=g), (torch.rand(5, 1, generator=g) > 0.5).float()
    assert rel_err(LF.sigmoid_focal_loss(o.cuda(), to.cuda(), 3.0, alpha=-1, gamma=0.0), LO.sigmoid_focal_loss(o, to, 3.0, alpha=-1, gamma=0.0)) < 1e-5
    with pytest.raises(LsamError):
        LF.dice_loss(x, t, 2.0, True)                                  # CPU tensors: no fallback
    with pytest.raises(LsamError):
        LF.dice_loss(x.cuda(), t.cuda().contiguous(), 2.0, True)       # materialised per-candidate targets are refused
