"""Loss goldens from the REFERENCE's own training/loss_fns.py (TEST INFRASTRUCTURE ONLY; run once in the build container:
  python oracle/make_loss_goldens.py ).

loss_fns.py imports `CORE_LOSS_KEY` from training.trainer, which pulls in hydra / tensordict / fvcore (absent offline):
as for `hydra` and `iopath` in oracle/make_goldens.py, an arithmetic-free stub module supplies that one string constant
(training/trainer.py:57); `training.utils.distributed` imports as it is.  The reference's MultiStepMultiMasksAndIous then
runs on seeded synthetic predictions (two frames, three objects of which one has an empty target, a 3-candidate step and a
1-candidate step) under its default settings and under the repository's config.yaml:26-37 settings; the inputs, the five
losses and the gradients of the core loss with respect to every prediction are stored in tests/golden/losses.npz."""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("LSAM_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

CONFIGS = {
    "default": dict(weight_dict={"loss_mask": 20, "loss_dice": 1, "loss_iou": 1, "loss_class": 1}),
    "repo": dict(weight_dict={"loss_mask": 20, "loss_dice": 1, "loss_iou": 1, "loss_class": 1}, supervise_all_iou=True,
                 iou_use_l1_loss=True, pred_obj_scores=True, focal_gamma_obj_score=0.0, focal_alpha_obj_score=-1.0),
}


def synthetic_case(seed=0, T=2, N=3, H=64):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(H), indexing="ij")
    targets = torch.zeros(T, N, H, H)
    for t in range(T):
        for n in range(N - 1):                                  # the last object's target stays empty (target_obj = 0)
            cy, cx, r = 16 + 12 * n + 5 * t, 20 + 10 * n, 8 + 3 * n
            targets[t, n] = (((yy - cy) ** 2 + (xx - cx) ** 2) < r * r).float()
    outs = []
    for t in range(T):
        steps_m = (3, 1)
        masks, ious, objs = [], [], []
        for m in steps_m:
            base = (targets[t].unsqueeze(1) * 2 - 1) * 3.0
            masks.append(base + 2.5 * torch.randn(N, m, H, H, generator=g))
            ious.append(torch.rand(N, m, generator=g))
            objs.append(2.0 * torch.randn(N, 1, generator=g))
        outs.append({"multistep_pred_multimasks_high_res": masks, "multistep_pred_ious": ious,
                     "multistep_object_score_logits": objs})
    return outs, targets


def import_reference_losses():
    tr = types.ModuleType("training.trainer")
    tr.CORE_LOSS_KEY = "core_loss"                              # training/trainer.py:57
    sys.path.insert(0, os.path.join(REF, "SAM2", "sam2"))
    import training                                             # the reference package itself
    sys.modules["training.trainer"] = tr
    training.trainer = tr
    from training import loss_fns
    return loss_fns


def run(loss_impl, cfg, outs, targets):
    leaves = []
    outs_g = []
    for o in outs:
        d = {}
        for k, lst in o.items():
            d[k] = [x.clone().requires_grad_() for x in lst]
            leaves += d[k]
        outs_g.append(d)
    losses = loss_impl(outs_g, targets, cfg)
    losses["core_loss"].backward()
    return {k: float(v) for k, v in losses.items()}, [x.grad if x.grad is not None else torch.zeros_like(x) for x in leaves]


def main():
    L = import_reference_losses()
    outs, targets = synthetic_case()
    gold = {"targets": targets.numpy()}
    for ti, o in enumerate(outs):
        for k, lst in o.items():
            for si, x in enumerate(lst):
                gold[f"in.{ti}.{k}.{si}"] = x.numpy()
    for name, cfg in CONFIGS.items():
        ref = lambda og, tg, c: dict(L.MultiStepMultiMasksAndIous(**{**c, "weight_dict": dict(c["weight_dict"])})(og, tg))
        losses, grads = run(ref, cfg, outs, targets)
        for k, v in losses.items():
            gold[f"{name}.{k}"] = np.float64(v)
        for i, gr in enumerate(grads):
            gold[f"{name}.grad.{i}"] = gr.numpy()
        print(name, losses)
    path = os.path.join(REPO, "tests", "golden", "losses.npz")
    np.savez_compressed(path, **gold)
    print("wrote", path, os.path.getsize(path))
    # the restatement agrees with what was just stored
    from oracle import loss_oracle as LO
    for name, cfg in CONFIGS.items():
        mine = lambda og, tg, c: LO.multistep_losses(og, tg, **c)
        losses, grads = run(mine, cfg, outs, targets)
        for k, v in losses.items():
            assert abs(v - float(gold[f"{name}.{k}"])) <= 1e-6 * max(1.0, abs(v)), (name, k, v, gold[f"{name}.{k}"])
        for i, gr in enumerate(grads):
            assert np.allclose(gr.numpy(), gold[f"{name}.grad.{i}"], rtol=1e-5, atol=1e-9), (name, i)
    print("oracle/loss_oracle.py agrees with the reference")


if __name__ == "__main__":
    main()
