"""Drop-in mirror of the reference's ImageEncoder / FpnNeck
(SAM2/sam2/sam2/modeling/backbones/image_encoder.py) and PositionEmbeddingSine
(SAM2/sam2/sam2/modeling/position_encoding.py:16-124, the `_pe` path only).

The 1x1 lateral convs run on the NHWC trunk outputs as `lsam_linear` GEMMs (bias fused),
with an input-gradient-only autograd rule (the neck is frozen by wrap_image_encoder_lora,
lora_qkv.py:159-160).  The nearest-x2 top-down add stays a PyTorch elementwise op in this
round (SURVEY.md section 8f row 1 lists the fully native neck as the next widening step).
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._C import LsamError
from .hieradet import _compute_dtype


class PositionEmbeddingSine(nn.Module):
    """Sine positional encoding, cached per (H, W) (position_encoding.py:89-124)."""

    def __init__(self, num_pos_feats, temperature: int = 10000, normalize: bool = True,
                 scale: Optional[float] = None, warmup_cache: bool = True, image_size: int = 1024,
                 strides=(4, 8, 16, 32)):
        super().__init__()
        assert num_pos_feats % 2 == 0, "Expecting even model width"
        self.num_pos_feats = num_pos_feats // 2
        self.temperature = temperature
        self.normalize = normalize
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.scale = 2 * math.pi if scale is None else scale
        self.cache = {}

    @torch.no_grad()
    def _pe(self, B, device, H, W):
        key = (H, W, str(device))
        if key not in self.cache:
            y = torch.arange(1, H + 1, dtype=torch.float32, device=device).view(-1, 1).expand(H, W)
            x = torch.arange(1, W + 1, dtype=torch.float32, device=device).view(1, -1).expand(H, W)
            if self.normalize:
                eps = 1e-6
                y = y / (y[-1:, :] + eps) * self.scale
                x = x / (x[:, -1:] + eps) * self.scale
            dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=device)
            dim_t = self.temperature ** (2 * (dim_t // 2) / self.num_pos_feats)
            px = x[:, :, None] / dim_t
            py = y[:, :, None] / dim_t
            px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
            py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
            self.cache[key] = torch.cat((py, px), dim=2).permute(2, 0, 1).contiguous()
        return self.cache[key][None].expand(B, -1, -1, -1)

    @torch.no_grad()
    def forward(self, x: torch.Tensor):
        return self._pe(x.shape[0], x.device, x.shape[-2], x.shape[-1])

    @torch.no_grad()
    def forward_as(self, x: torch.Tensor):
        """The encoding in x's dtype, as a batch-EXPANDED view of one cached [C,H,W] tensor per
        (H, W, device, dtype).  The reference materialises `pos.repeat(B,1,1,1).to(x.dtype)` per call
        (position_encoding.py:117-124, image_encoder.py:130): 0.5 GB of constant bf16 at level 0 of a
        16-frame batch; values and shape are the same here, only the batch stride is 0."""
        B, _, H, W = x.shape
        key = (H, W, str(x.device), x.dtype)
        if key not in self.cache:
            self.cache[key] = self._pe(1, x.device, H, W)[0].to(x.dtype).contiguous()
        return self.cache[key][None].expand(B, -1, -1, -1)


class _Conv1x1Fn(torch.autograd.Function):
    """y[NHWC] = x[NHWC] . W^T + b on MFMA; backward gives dx only (W, b frozen)."""

    @staticmethod
    def forward(ctx, x_nhwc, weight, bias, dt):
        N, H, W, Cc = x_nhwc.shape
        co = weight.shape[0]
        if dt == torch.bfloat16 and x_nhwc.dtype == torch.float32 and ops.lateral_conv_supported(Cc, co):
            # high-resolution levels: one kernel from the fp32 stage output (no bf16 copy of x in memory)
            out = ops.lateral_conv(x_nhwc.reshape(-1, Cc), weight, bias)
        else:
            xt = ops.cast(x_nhwc.reshape(-1, Cc), dt)
            wt = ops.pack_matrix(weight.detach().reshape(co, Cc), dt)
            out = ops.linear(xt, wt, bias.detach())
        ctx.save_for_backward(weight)
        ctx.dt, ctx.shape = dt, (N, H, W, Cc)
        return out.view(N, H, W, co)

    @staticmethod
    def backward(ctx, g):
        (weight,) = ctx.saved_tensors
        N, H, W, Cc = ctx.shape
        co = weight.shape[0]
        gt = ops.cast(g.reshape(-1, co).float().contiguous(), ctx.dt)
        wT = ops.pack_matrix(weight.detach().reshape(co, Cc).t(), ctx.dt)   # [C, co]
        dx = ops.linear(gt, wT, out_f32=True)
        return dx.view(N, H, W, Cc), None, None, None


class FpnNeck(nn.Module):
    """image_encoder.py:45-134.  Same constructor and `convs.{i}.conv.{weight,bias}` keys."""

    def __init__(self, position_encoding: nn.Module, d_model: int, backbone_channel_list: List[int],
                 kernel_size: int = 1, stride: int = 1, padding: int = 0, fpn_interp_model: str = "bilinear",
                 fuse_type: str = "sum", fpn_top_down_levels: Optional[List[int]] = None):
        super().__init__()
        if kernel_size != 1 or stride != 1 or padding != 0:
            raise NotImplementedError("the reference factory uses 1x1 lateral convs (sam2_model.py:18-24)")
        self.position_encoding = position_encoding
        self.convs = nn.ModuleList()
        self.backbone_channel_list = backbone_channel_list
        self.d_model = d_model
        for dim in backbone_channel_list:
            current = nn.Sequential()
            current.add_module("conv", nn.Conv2d(in_channels=dim, out_channels=d_model, kernel_size=1))
            self.convs.append(current)
        self.fpn_interp_model = fpn_interp_model
        assert fuse_type in ["sum", "avg"]
        self.fuse_type = fuse_type
        if fpn_top_down_levels is None:
            fpn_top_down_levels = range(len(self.convs))
        self.fpn_top_down_levels = list(fpn_top_down_levels)

    def forward(self, xs: List[torch.Tensor]):
        out = [None] * len(self.convs)
        pos = [None] * len(self.convs)
        assert len(xs) == len(self.convs)
        dt = _compute_dtype(None)
        prev = None
        n = len(self.convs) - 1
        for i in range(n, -1, -1):
            conv = self.convs[n - i].conv
            if torch.is_grad_enabled() and (conv.weight.requires_grad or conv.bias.requires_grad):
                raise LsamError("FpnNeck parameters must be frozen (wrap_image_encoder_lora does this); "
                                "liblsam has no weight-gradient GEMM for frozen tensors by design")
            x_nhwc = xs[i].permute(0, 2, 3, 1)       # the trunk hands out NCHW views of NHWC storage
            if not x_nhwc.is_contiguous():
                x_nhwc = x_nhwc.contiguous()
            lateral = _Conv1x1Fn.apply(x_nhwc, conv.weight, conv.bias, dt).permute(0, 3, 1, 2)
            if i in self.fpn_top_down_levels and prev is not None:
                top_down = F.interpolate(prev.to(dtype=torch.float32), scale_factor=2.0, mode=self.fpn_interp_model,
                                         align_corners=(None if self.fpn_interp_model == "nearest" else False),
                                         antialias=False)
                prev = lateral + top_down
                if self.fuse_type == "avg":
                    prev = prev / 2
            else:
                prev = lateral
            out[i] = prev
            pe_as = getattr(self.position_encoding, "forward_as", None)   # any user module with the reference's API works
            pos[i] = pe_as(prev) if pe_as is not None else self.position_encoding(prev).to(prev.dtype)
        return out, pos


class ImageEncoder(nn.Module):
    """image_encoder.py:14-42."""

    def __init__(self, trunk: nn.Module, neck: nn.Module, scalp: int = 0):
        super().__init__()
        self.trunk = trunk
        self.neck = neck
        self.scalp = scalp
        assert self.trunk.channel_list == self.neck.backbone_channel_list, (
            f"Channel dims of trunk and neck do not match. Trunk: {self.trunk.channel_list}, "
            f"neck: {self.neck.backbone_channel_list}")

    def forward(self, sample: torch.Tensor):
        features, pos = self.neck(self.trunk(sample))
        if self.scalp > 0:
            features, pos = features[: -self.scalp], pos[: -self.scalp]
        return {"vision_features": features[-1], "vision_pos_enc": pos, "backbone_fpn": features}


def build_image_encoder(hiera_kwargs: dict, d_model: int = 256, scalp: int = 1) -> ImageEncoder:
    """The reference factory's encoder (sam2_model.py:12-26) for any Hiera size."""
    from .hieradet import Hiera
    trunk = Hiera(**hiera_kwargs)
    neck = FpnNeck(position_encoding=PositionEmbeddingSine(num_pos_feats=256, normalize=True, temperature=10000),
                   d_model=d_model, backbone_channel_list=list(trunk.channel_list),
                   fpn_top_down_levels=[2, 3], fpn_interp_model="nearest")
    return ImageEncoder(trunk=trunk, neck=neck, scalp=scalp)
