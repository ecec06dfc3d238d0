"""Drop-in mirror of the reference's mask losses (SAM2/sam2/training/loss_fns.py): `dice_loss`, `sigmoid_focal_loss`,
`iou_loss` and `MultiStepMultiMasksAndIous` with the reference's names, arguments, defaults and returned dictionary
(`loss_mask`, `loss_dice`, `loss_iou`, `loss_class`, `core_loss`).

The work over the pixels -- the only part with any size: [N, M, H, W] logits at 1024^2 -- is ONE liblsam pass per
prediction step (`lsam_mask_loss_sums`: per candidate mask the focal sum, sum p t, sum p, sum t and the two areas) and one
elementwise pass in the backward (`lsam_mask_loss_bwd`); what is left is arithmetic on [N, M] tensors (ratios, the
arg-min over the candidates, the weighted sum), which torch does on the device, autograd included.
Binary {0, 1} targets are ASSUMED, as every caller of the reference provides them, and NOT checked (a check would be one
more pass over the targets, or a host synchronisation as in the reference): the reference's `targets.sigmoid()` branch for
targets outside [0, 1] (loss_fns.py:35-36) is not built, so results for such targets are undefined here."""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List

import torch
import torch.nn as nn

from . import heads_autograd as HA
from ._C import LsamError

CORE_LOSS_KEY = "core_loss"   # training/trainer.py:57


def _sums(inputs: torch.Tensor, targets: torch.Tensor, alpha: float, gamma: float) -> torch.Tensor:
    """[N, M, 6] sums for logits [N, M, ...] against targets [N, ...] or [N, M, ...] whose M copies are the same tensor
    (loss_fns.py:240 expands one target over the candidates)."""
    if not inputs.is_cuda:
        raise LsamError("the mask losses run on the MI355X only (no CPU or PyTorch fallback)")
    if targets.dim() == inputs.dim():
        if targets.shape[1] != 1 and targets.stride(1) != 0:
            raise LsamError("mask losses: per-candidate targets are not built (the reference expands ONE target, loss_fns.py:240)")
        targets = targets[:, 0]
    return HA.MaskLossSumsFn.apply(inputs.float().contiguous(), targets.float().contiguous(), float(alpha), float(gamma))


def dice_loss(inputs, targets, num_objects, loss_on_multimask=False):
    """loss_fns.py:19-55.  `loss_on_multimask`: [N, M, H, W] inputs and targets, one dice per (n, m).  Otherwise the reference
    flattens the inputs to [N, K] and multiplies by the targets AS GIVEN (loss_fns.py:47-48), which only works for targets
    that are already [N, K]: that form is reproduced, one dice per n; other shapes fail in the reference and raise here."""
    if loss_on_multimask:
        if inputs.dim() != 4 or targets.dim() != 4:
            raise LsamError("dice_loss(loss_on_multimask=True) takes [N, M, H, W] inputs and targets (loss_fns.py:41)")
        x, t = inputs, targets
    else:
        if targets.dim() != 2 or inputs.shape[0] != targets.shape[0] or inputs[0].numel() != targets.shape[1]:
            raise LsamError("dice_loss(loss_on_multimask=False): the reference's form needs targets of shape [N, K] matching the "
                            "flattened inputs (loss_fns.py:47-48); other shapes fail there too")
        x, t = inputs.reshape(inputs.shape[0], 1, -1), targets.unsqueeze(1)
    s = _sums(x, t, -1.0, 0.0)
    loss = 1 - (2 * s[..., 1] + 1) / (s[..., 2] + s[..., 3] + 1)
    return loss / num_objects if loss_on_multimask else loss.sum() / num_objects


def sigmoid_focal_loss(inputs, targets, num_objects, alpha: float = 0.25, gamma: float = 2, loss_on_multimask=False):
    """loss_fns.py:58-98."""
    if loss_on_multimask:
        assert inputs.dim() == 4
        s = _sums(inputs, targets, alpha, gamma)
        return s[..., 0] / float(inputs[0, 0].numel()) / num_objects
    # [N, K] form (the object-score loss: K = 1): mean over dim 1, sum over the batch
    n, k = inputs.shape[0], inputs[0].numel()
    s = _sums(inputs.reshape(n, 1, k), targets.reshape(n, 1, k), alpha, gamma)
    return (s[..., 0] / float(k)).sum() / num_objects


def iou_loss(inputs, targets, pred_ious, num_objects, loss_on_multimask=False, use_l1_loss=False):
    """loss_fns.py:101-129."""
    assert inputs.dim() == 4 and targets.dim() == 4
    s = _sums(inputs, targets, -1.0, 0.0).detach()
    actual = s[..., 4] / torch.clamp(s[..., 5], min=1.0)
    d = pred_ious - actual
    loss = d.abs() if use_l1_loss else d * d
    return loss / num_objects if loss_on_multimask else loss.sum() / num_objects


class MultiStepMultiMasksAndIous(nn.Module):
    """loss_fns.py:132-313.  One `lsam_mask_loss_sums` pass per prediction step serves the focal, dice and IoU terms."""

    def __init__(self, weight_dict, focal_alpha=0.25, focal_gamma=2, supervise_all_iou=False, iou_use_l1_loss=False,
                 pred_obj_scores=False, focal_gamma_obj_score=0.0, focal_alpha_obj_score=-1):
        super().__init__()
        self.weight_dict = weight_dict
        self.focal_alpha = focal_alpha
        self.focal_gamma = focal_gamma
        assert "loss_mask" in self.weight_dict
        assert "loss_dice" in self.weight_dict
        assert "loss_iou" in self.weight_dict
        if "loss_class" not in self.weight_dict:
            self.weight_dict["loss_class"] = 0.0
        self.focal_alpha_obj_score = focal_alpha_obj_score
        self.focal_gamma_obj_score = focal_gamma_obj_score
        self.supervise_all_iou = supervise_all_iou
        self.iou_use_l1_loss = iou_use_l1_loss
        self.pred_obj_scores = pred_obj_scores

    def forward(self, outs_batch: List[Dict], targets_batch: torch.Tensor):
        assert len(outs_batch) == len(targets_batch)
        num_objects = torch.tensor((targets_batch.shape[1]), device=targets_batch.device, dtype=torch.float)
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():   # loss_fns.py:179 (one scalar all-reduce)
            torch.distributed.all_reduce(num_objects)
            world = torch.distributed.get_world_size()
        num_objects = torch.clamp(num_objects / world, min=1).item()
        losses = defaultdict(int)
        for outs, targets in zip(outs_batch, targets_batch):
            for k, v in self._forward(outs, targets, num_objects).items():
                losses[k] += v
        return losses

    def _forward(self, outputs: Dict, targets: torch.Tensor, num_objects):
        target_masks = targets.unsqueeze(1).float()
        assert target_masks.dim() == 4
        src_masks_list = outputs["multistep_pred_multimasks_high_res"]
        ious_list = outputs["multistep_pred_ious"]
        object_score_logits_list = outputs["multistep_object_score_logits"]
        assert len(src_masks_list) == len(ious_list)
        assert len(object_score_logits_list) == len(ious_list)
        losses = {"loss_mask": 0, "loss_dice": 0, "loss_iou": 0, "loss_class": 0}
        for src_masks, ious, object_score_logits in zip(src_masks_list, ious_list, object_score_logits_list):
            self._update_losses(losses, src_masks, target_masks, ious, num_objects, object_score_logits)
        losses[CORE_LOSS_KEY] = self.reduce_loss(losses)
        return losses

    def _update_losses(self, losses, src_masks, target_masks, ious, num_objects, object_score_logits):
        hw = float(src_masks[0, 0].numel())
        s = _sums(src_masks, target_masks, self.focal_alpha, self.focal_gamma)               # [N, M, 6], one pass
        loss_multimask = s[..., 0] / hw / num_objects
        loss_multidice = (1 - (2 * s[..., 1] + 1) / (s[..., 2] + s[..., 3] + 1)) / num_objects
        if not self.pred_obj_scores:
            loss_class = torch.tensor(0.0, dtype=loss_multimask.dtype, device=loss_multimask.device)
            target_obj = torch.ones(loss_multimask.shape[0], 1, dtype=loss_multimask.dtype, device=loss_multimask.device)
        else:
            target_obj = (s[:, :1, 3].detach() > 0).float()                                   # any(target > 0): sum t > 0 for {0, 1} targets
            loss_class = sigmoid_focal_loss(object_score_logits, target_obj, num_objects, alpha=self.focal_alpha_obj_score,
                                            gamma=self.focal_gamma_obj_score)
        sd = s.detach()
        actual_ious = sd[..., 4] / torch.clamp(sd[..., 5], min=1.0)
        d = ious - actual_ious
        loss_multiiou = (d.abs() if self.iou_use_l1_loss else d * d) / num_objects
        assert loss_multimask.dim() == 2 and loss_multidice.dim() == 2 and loss_multiiou.dim() == 2
        if loss_multimask.size(1) > 1:
            loss_combo = loss_multimask * self.weight_dict["loss_mask"] + loss_multidice * self.weight_dict["loss_dice"]
            best_loss_inds = torch.argmin(loss_combo, dim=-1)
            batch_inds = torch.arange(loss_combo.size(0), device=loss_combo.device)
            loss_mask = loss_multimask[batch_inds, best_loss_inds].unsqueeze(1)
            loss_dice = loss_multidice[batch_inds, best_loss_inds].unsqueeze(1)
            if self.supervise_all_iou:
                loss_iou = loss_multiiou.mean(dim=-1).unsqueeze(1)
            else:
                loss_iou = loss_multiiou[batch_inds, best_loss_inds].unsqueeze(1)
        else:
            loss_mask, loss_dice, loss_iou = loss_multimask, loss_multidice, loss_multiiou
        loss_mask = loss_mask * target_obj
        loss_dice = loss_dice * target_obj
        loss_iou = loss_iou * target_obj
        losses["loss_mask"] += loss_mask.sum()
        losses["loss_dice"] += loss_dice.sum()
        losses["loss_iou"] += loss_iou.sum()
        losses["loss_class"] += loss_class

    def reduce_loss(self, losses):
        reduced_loss = 0.0
        for loss_key, weight in self.weight_dict.items():
            if loss_key not in losses:
                raise ValueError(f"{type(self)} doesn't compute {loss_key}")
            if weight != 0:
                reduced_loss += losses[loss_key] * weight
        return reduced_loss
