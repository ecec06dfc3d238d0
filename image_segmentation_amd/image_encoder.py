"""Drop-in mirror of the reference's ImageEncoder / FpnNeck
(SAM2/sam2/sam2/modeling/backbones/image_encoder.py) and PositionEmbeddingSine
(SAM2/sam2/sam2/modeling/position_encoding.py:16-124, the `_pe` path only).

Every FPN level is ONE HIP kernel on the NHWC trunk output: lateral 1x1 conv + bias + nearest-x2
top-down add (`lsam_neck_level_fwd` / `lsam_lateral_conv_fwd`), with packed weights cached on the
module; the backward is the top-down adjoint plus one input-gradient GEMM for the last stage only
(the neck is frozen by wrap_image_encoder_lora, lora_qkv.py:159-160).  `forward_image` mirrors
SAM2Base.forward_image (sam2_base.py:467-479) including mask_decoder.conv_s0 / conv_s1, which are
composed with the lateral convolutions of levels 0 / 1 into single 112 -> 32 / 224 -> 64 kernels.
No PyTorch arithmetic remains in the encoder's forward or backward.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn

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


class _NeckPack:
    """Packed operands of one FpnNeck for one (dtype, device): a weight image per level for the forward kernel, fp32
    biases, and -- for the level the LoRA gradient passes through -- the transposed weight for dx = g . W.  Built once
    and rebuilt only when a conv parameter changes (data_ptr / _version) or `invalidate_packed()` is called."""

    def __init__(self, neck: "FpnNeck", dt: torch.dtype, device: torch.device):
        self.sig = neck._pack_signature()
        self.img, self.bias, self.lat_img = [], [], []
        self.split_img = [None] * len(neck.convs)     # [hi | lo] images of the top-down levels, built on first use
        for seq in neck.convs:
            w, b = seq.conv.weight.detach(), seq.conv.bias.detach()
            w = w.to(device=device, dtype=torch.float32).reshape(w.shape[0], w.shape[1]).contiguous()
            co, K = w.shape
            use_lateral = dt == torch.bfloat16 and ops.lateral_conv_supported(K, co)
            # bf16, K <= 224 (the two high-resolution levels): the LDS-resident-weight kernel of lateral.hip
            self.lat_img.append(ops.lateral_conv_pack(w) if use_lateral else None)
            self.img.append(None if use_lateral else ops.neck_conv_pack(w, dt))
            self.bias.append(b.to(device=device, dtype=torch.float32).contiguous())
        wl = neck.convs[0].conv.weight.detach()     # convs[0] serves the LAST (coarsest) level: convs[n - i], image_encoder.py:114
        self.wT_last = ops.pack_matrix(wl.to(device=device, dtype=torch.float32).reshape(wl.shape[0], wl.shape[1]).t(), dt)


class _NeckFn(torch.autograd.Function):
    """The whole FPN neck (image_encoder.py:102-134) as one autograd node: one HIP kernel per level in the forward
    (lateral 1x1 conv + bias + nearest-x2 top-down add, T NHWC out).  The neck is frozen (lora_qkv.py:159-160) and
    only the LAST stage output carries a gradient (the LoRA blocks; hieradet.py:306-334 runs the other stages under
    no_grad), so the backward is the top-down chain's adjoint into the coarsest level followed by ONE GEMM
    dx_last = g_last . W_last -- no weight gradient, no input gradient for the other levels."""

    @staticmethod
    def forward(ctx, neck, pack, dt, base, *xs):
        # xs = the stage outputs of levels base .. last (base > 0: forward_image computes the finer levels elsewhere)
        n = len(xs) - 1
        outs = [None] * (n + 1)
        prev = None
        ctx.chain = []            # (local) levels, fine -> coarse, whose gradient flows into the next coarser one
        dep = [False] * (n + 2)
        # precision policy (FpnNeck.precise_top_down): under bf16, the levels of the top-down chain -- 5 % of the neck's
        # FLOPs, and what the SAM transformer reads -- use two-term bf16 operands (fp32-accurate products, fp32 top-down
        # addends) and round ONCE, at the bf16 store; plain bf16 would round inputs, weights and outputs (0.33 % against
        # 0.17 % error on the level the masks come from: mask IoU 0.989 against 0.993 on the north-star fixture)
        precise = dt == torch.bfloat16 and neck.precise_top_down and neck.d_model == 256
        prev32 = None
        for i in range(n, -1, -1):
            x = xs[i]
            co = neck.d_model
            in_chain = (base + i) in neck.fpn_top_down_levels
            top = prev if (in_chain and prev is not None) else None
            j = n - i             # convs[n_levels - 1 - level] (image_encoder.py:114): index from the coarse end
            if precise and in_chain:
                if pack.split_img[j] is None:
                    pack.split_img[j] = ops.neck_conv_pack_split(neck.convs[j].conv.weight)
                t32 = None
                if top is not None:
                    t32 = prev32 if prev32 is not None else ops.cast(top.view(-1, co), torch.float32).view(top.shape)
                feeds_finer = i > 0 and (base + i - 1) in neck.fpn_top_down_levels
                out, prev32 = ops.neck_level_fwd_split(x, pack.split_img[j], pack.bias[j], co, t32, want32=feeds_finer)
            elif pack.lat_img[j] is not None and top is None:
                N, H, W, K = x.shape
                out = ops.lateral_conv_fwd(x.reshape(-1, K), pack.lat_img[j], pack.bias[j], co).view(N, H, W, co)
                prev32 = None
            else:
                img = pack.img[j] if pack.img[j] is not None else None
                if img is None:
                    img = pack.img[j] = ops.neck_conv_pack(neck.convs[j].conv.weight.detach().float(), dt)
                out = ops.neck_level_fwd(x, img, pack.bias[j], dt, co, top)
                prev32 = None
            if top is not None:
                ctx.chain.append(i)
            # only levels fed (through top-down adds) by the LAST stage output depend on a differentiable input
            dep[i] = i == n or (top is not None and dep[i + 1])
            outs[i] = prev = out
        ctx.pack, ctx.dt, ctx.n = pack, dt, n
        ctx.x_last_shape = tuple(xs[n].shape)
        ctx.set_materialize_grads(False)
        # the others must not make autograd differentiate the loss with respect to them (0.5 GB maps at level 0)
        ctx.mark_non_differentiable(*[o for i, o in enumerate(outs) if not dep[i]])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        n, dt = ctx.n, ctx.dt
        none = (None, None, None, None) + (None,) * (n + 1)
        if not ctx.needs_input_grad[4 + n]:
            return none
        # gradient arriving at the coarsest level's output: its own plus, through every top-down add between, the 2x2
        # sums of the finer levels' gradients (prev = lateral + up(prev), so d prev_coarse += sumpool(d prev_fine))
        g = [None if t is None else t.to(dt).contiguous() for t in gs]
        carry = None
        for i in range(0, n + 1):
            cur = g[i]
            if carry is not None:
                cur = carry if cur is None else cur + carry        # (both present only with a loss on several chained levels)
            carry = None
            if i < n and (i in ctx.chain) and cur is not None:
                carry = ops.neck_topdown_bwd(cur)
            elif i == n:
                carry = cur
        if carry is None:
            return none
        N, H, W, K = ctx.x_last_shape
        dx = ops.linear(carry.reshape(-1, carry.shape[-1]), ctx.pack.wT_last, out_f32=True)
        return (None, None, None, None) + (None,) * n + (dx.view(N, H, W, K),)


class FpnNeck(nn.Module):
    """image_encoder.py:45-134.  Same constructor and `convs.{i}.conv.{weight,bias}` keys."""

    def __init__(self, position_encoding: nn.Module, d_model: int, backbone_channel_list: List[int],
                 kernel_size: int = 1, stride: int = 1, padding: int = 0, fpn_interp_model: str = "bilinear",
                 fuse_type: str = "sum", fpn_top_down_levels: Optional[List[int]] = None):
        super().__init__()
        if kernel_size != 1 or stride != 1 or padding != 0:
            raise NotImplementedError("the reference factory uses 1x1 lateral convs (sam2_model.py:18-24)")
        if fpn_interp_model != "nearest" or fuse_type != "sum":
            raise NotImplementedError("liblsam implements the reference factory's neck: fpn_interp_model='nearest', "
                                      "fuse_type='sum' (sam2_model.py:22-24)")
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
        self._packs = {}
        self._pack_epoch = 0
        self.precise_top_down = True      # see _NeckFn.forward; False = every level in the autocast dtype
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    def invalidate_packed(self):
        """Rebuild the packed conv weights on the next forward (after editing parameters through `.data`)."""
        self._pack_epoch += 1

    def _pack_signature(self):
        return (self._pack_epoch,) + tuple((p.data_ptr(), p._version) for p in self.convs.parameters())

    def _pack_for(self, dt: torch.dtype, device: torch.device) -> _NeckPack:
        key = (dt, device)
        pack = self._packs.get(key)
        if pack is None or pack.sig != self._pack_signature():
            pack = self._packs[key] = _NeckPack(self, dt, device)
        return pack

    def _nhwc(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        out = []
        for x in xs:
            if not x.is_cuda:
                raise LsamError("FpnNeck runs on the MI355X only (no CPU or PyTorch fallback)")
            x = x.permute(0, 2, 3, 1)          # the trunk hands out NCHW views of NHWC storage
            if x.dtype != torch.float32 or not x.is_contiguous():
                x = x.to(torch.float32).contiguous()
            out.append(x)
        return out

    def forward(self, xs: List[torch.Tensor]):
        assert len(xs) == len(self.convs)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.convs.parameters()):
            raise LsamError("FpnNeck parameters must be frozen (wrap_image_encoder_lora does this); "
                            "liblsam has no weight-gradient GEMM for frozen tensors by design")
        dt = _compute_dtype(None)
        xs = self._nhwc(xs)
        pack = self._pack_for(dt, xs[0].device)
        feats = _NeckFn.apply(self, pack, dt, 0, *xs)
        out = [f.permute(0, 3, 1, 2) for f in feats]       # NCHW tensors in channels_last memory
        pe_as = getattr(self.position_encoding, "forward_as", None)   # any user module with the reference's API works
        pos = [pe_as(f) if pe_as is not None else self.position_encoding(f).to(f.dtype) for f in out]
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


class _HighResPack:
    def __init__(self, sig, imgs, biases):
        self.sig, self.imgs, self.biases = sig, imgs, biases


def forward_image(model, img_batch: torch.Tensor):
    """SAM2Base.forward_image (sam2_base.py:467-479) for any object with the reference's attributes: `image_encoder`,
    `use_high_res_features_in_sam` and `sam_mask_decoder.conv_s0 / conv_s1` (mask_decoder.py:78-81, Conv2d(256, 32, 1) and
    Conv2d(256, 64, 1)).

    The reference runs the image encoder and then overwrites backbone_fpn[0] / [1] with conv_s0 / conv_s1 of them.  A
    1x1 convolution of a 1x1 convolution with nothing in between is one 1x1 convolution, so levels 0 and 1 come straight
    from the stage-1 / stage-2 trunk outputs through the COMPOSED weights W_s . W_lateral (112 -> 32, 224 -> 64; one
    lsam_neck_level_fwd launch each): the 256-channel maps of those two levels -- 537 + 134 MB per 16-frame batch that
    the reference writes only to read them back once -- are never formed.  The composition is done once, in fp32, when
    the weights are packed.  Gradients: as in ImageEncoder.forward, only the LoRA factors (through level 2)."""
    enc = model.image_encoder
    if not getattr(model, "use_high_res_features_in_sam", False):
        return enc(img_batch)
    neck = enc.neck
    if enc.scalp != 1 or len(neck.convs) != 4 or any(l in neck.fpn_top_down_levels for l in (0, 1)):
        raise LsamError("forward_image: the composed high-resolution path expects the reference factory's neck "
                        "(4 levels, scalp=1, fpn_top_down_levels=[2, 3]; sam2_model.py:18-26)")
    dec = model.sam_mask_decoder
    convs = (dec.conv_s0, dec.conv_s1)
    if torch.is_grad_enabled() and any(p.requires_grad for c in convs for p in c.parameters()):
        raise LsamError("forward_image: conv_s0 / conv_s1 must be frozen (liblsam has no weight-gradient GEMM)")
    dt = _compute_dtype(None)
    stages = enc.trunk(img_batch)
    xs = neck._nhwc(stages)
    dev = xs[0].device
    n = len(xs) - 1
    # levels 2 (and 3 feeding it) through the neck proper
    pack = neck._pack_for(dt, dev)
    sig = neck._pack_signature() + tuple((p.data_ptr(), p._version) for c in convs for p in c.parameters())
    cache = getattr(neck, "_hr_packs", None)
    if cache is None:
        cache = neck._hr_packs = {}
    hp = cache.get((dt, dev))
    if hp is None or hp.sig != sig:
        imgs, biases = [], []
        for lvl, cs in enumerate(convs):
            lat = neck.convs[n - lvl].conv
            wl = lat.weight.detach().to(device=dev, dtype=torch.float32).reshape(lat.weight.shape[0], -1)
            ws = cs.weight.detach().to(device=dev, dtype=torch.float32).reshape(cs.weight.shape[0], -1)
            imgs.append(ops.neck_conv_pack((ws @ wl).contiguous(), dt))               # packing: W_s . W_lateral, once
            biases.append((ws @ lat.bias.detach().to(dev, torch.float32) + cs.bias.detach().to(dev, torch.float32)).contiguous())
        hp = cache[(dt, dev)] = _HighResPack(sig, imgs, biases)
    with torch.no_grad():
        hi = [ops.neck_level_fwd(xs[lvl], hp.imgs[lvl], hp.biases[lvl], dt, convs[lvl].out_channels) for lvl in (0, 1)]
    low = _NeckFn.apply(neck, pack, dt, 2, *xs[2:])    # levels 2, 3 of the same neck
    feats = [hi[0].permute(0, 3, 1, 2), hi[1].permute(0, 3, 1, 2), low[0].permute(0, 3, 1, 2)]
    pe_as = getattr(neck.position_encoding, "forward_as", None)
    pos = [pe_as(f) if pe_as is not None else neck.position_encoding(f).to(f.dtype) for f in feats]
    return {"vision_features": feats[-1], "vision_pos_enc": pos, "backbone_fpn": feats}


def build_image_encoder(hiera_kwargs: dict, d_model: int = 256, scalp: int = 1) -> ImageEncoder:
    """The reference factory's encoder (sam2_model.py:12-26) for any Hiera size."""
    from .hieradet import Hiera
    trunk = Hiera(**hiera_kwargs)
    neck = FpnNeck(position_encoding=PositionEmbeddingSine(num_pos_feats=256, normalize=True, temperature=10000),
                   d_model=d_model, backbone_channel_list=list(trunk.channel_list),
                   fpn_top_down_levels=[2, 3], fpn_interp_model="nearest")
    return ImageEncoder(trunk=trunk, neck=neck, scalp=scalp)
