"""Drop-in mirror of the reference's Hiera trunk
(SAM2/sam2/sam2/modeling/backbones/hieradet.py) whose arithmetic runs in liblsam.so.

Same class names, constructor signatures, attribute names and state_dict keys as the
reference, so Meta checkpoints, `wrap_image_encoder_lora` and saved LoRA files keep
working.  nn.Parameters are only weight *storage* here: the
forward of `Hiera` is a single `lsam_hiera_forward` call (HIP kernels for every op,
no torch math), the backward a single `lsam_hiera_backward` call that yields
gradients for the LoRA A/B factors only.

Precision follows the reference: fp32 by default (training drivers use no autocast),
bf16 MFMA under `torch.autocast("cuda", dtype=torch.bfloat16)`; set
`trunk.compute_dtype = torch.bfloat16` to force it.
"""
from __future__ import annotations

import ctypes as C
from functools import partial
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _C, ops
from ._C import check, lib, ptr, stream


class PatchEmbed(nn.Module):
    """backbones/utils.py:63-93.  Holds the 7x7/s4/p3 conv weights.  Inside `Hiera.forward` the convolution is the direct
    MFMA kernel of csrc/patchembed.hip (bias and position embedding fused); this stand-alone `forward` is the unit-operator
    form (lsam_im2col_patch + lsam_linear), kept for callers that invoke the module on its own."""

    def __init__(self, kernel_size=(7, 7), stride=(4, 4), padding=(3, 3), in_chans: int = 3,
                 embed_dim: int = 768):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        dt = _compute_dtype(None)
        e = self.proj.out_channels
        cols = ops.im2col_patch(x.float().contiguous(), dt)
        w = ops.pack_matrix(self.proj.weight.detach().reshape(e, -1), dt, 160)
        out = ops.linear(cols, w, self.proj.bias.detach(), out_f32=True)
        g = x.shape[-1] // 4
        return out.view(x.shape[0], g, g, e)


class MLP(nn.Module):
    """sam2_utils.py:112-136 (2-layer GELU MLP in Hiera)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, activation=nn.ReLU,
                 sigmoid_output: bool = False):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))
        self.sigmoid_output = sigmoid_output
        self.act = activation()

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """sam2_utils.py:131-136 as unit operators (inference): one `lsam_linear` per layer with the activation (exact-erf
        GELU in Hiera, ReLU in the SAM heads) and the optional final sigmoid in its epilogue; fp32 out.  Inside `Hiera.forward`
        / `SAM2Base._forward_sam_heads` the same layers run fused or through autograd nodes instead."""
        if not isinstance(self.act, (nn.GELU, nn.ReLU)):
            raise _C.LsamError(f"MLP.forward: activation {type(self.act).__name__} is not built (GELU and ReLU are)")
        dt = _compute_dtype(None)
        lead = x.shape[:-1]
        h = x.reshape(-1, x.shape[-1])
        h = h.to(dt).contiguous() if h.dtype != dt else h.contiguous()
        for i, layer in enumerate(self.layers):
            last = i == self.num_layers - 1
            n, k = layer.weight.shape
            n8, k8 = (n + 7) // 8 * 8, (k + 7) // 8 * 8
            wp = torch.zeros(n8, k8, dtype=torch.float32, device=h.device)       # placement only: pad to the GEMM's granule
            wp[:n, :k] = layer.weight.detach().float()
            bp = torch.zeros(n8, dtype=torch.float32, device=h.device)
            bp[:n] = layer.bias.detach().float()
            if h.shape[1] != k8:
                hp = torch.zeros(h.shape[0], k8, dtype=h.dtype, device=h.device)
                hp[:, :k] = h[:, :k]
                h = hp
            h = ops.linear(h, ops.pack_matrix(wp, dt), bp, gelu=not last and isinstance(self.act, nn.GELU),
                           relu=not last and isinstance(self.act, nn.ReLU), sigmoid=last and self.sigmoid_output, out_f32=last)
            if last:
                h = h[:, :n]
        return h.reshape(*lead, h.shape[-1])


def _compute_dtype(explicit: Optional[torch.dtype]) -> torch.dtype:
    if explicit is not None:
        return explicit
    if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        return torch.bfloat16
    return torch.float32


class MultiScaleAttention(nn.Module):
    """hieradet.py:39-81."""

    def __init__(self, dim: int, dim_out: int, num_heads: int, q_pool: nn.Module = None):
        super().__init__()
        self.dim = dim
        self.dim_out = dim_out
        self.num_heads = num_heads
        self.q_pool = q_pool
        self.qkv = nn.Linear(dim, dim_out * 3)
        self.proj = nn.Linear(dim_out, dim_out)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """hieradet.py:56-81 as unit operators (inference): x [B, H, W, dim] -> [B, H', W', dim_out]; every one of the B maps is
        ONE attention window of H x W tokens (the reference calls this on window-partitioned input), q max-pooled 2 x 2 when
        the block has a q_pool.  `MultiScaleBlock.forward` / `Hiera.forward` do not come through here (they address windows in
        place); geometries the attention kernels do not cover raise LsamError."""
        from .lora_qkv import LoRA_qkv
        dt = _compute_dtype(None)
        B, H, W, d = x.shape
        q = self.dim_out
        xt = x.reshape(-1, d)
        xt = xt.to(dt).contiguous() if xt.dtype != dt else xt.contiguous()
        if isinstance(self.qkv, LoRA_qkv):
            qkv, bias = self.qkv._forward_packed(xt), self.qkv.qkv.bias
        else:
            qkv, bias = ops.linear(xt, ops.pack_matrix(self.qkv.weight.detach().float(), dt), self.qkv.bias), self.qkv.bias
        pool = 2 if self.q_pool is not None else 1
        ws = H if H == W else 0            # one H x H window per map; a non-square map takes the global form (window 0)
        a = ops.attention_fwd(qkv.view(B, H, W, 3 * q), bias, ws, pool, self.num_heads)
        Ho, Wo = a.shape[1:3]
        out = ops.linear(a.view(-1, q), ops.pack_matrix(self.proj.weight.detach().float(), dt), self.proj.bias, out_f32=True)
        return out.view(B, Ho, Wo, q)


class MultiScaleBlock(nn.Module):
    """hieradet.py:84-166."""

    def __init__(self, dim: int, dim_out: int, num_heads: int, mlp_ratio: float = 4.0,
                 drop_path: float = 0.0, norm_layer: Union[nn.Module, str] = "LayerNorm",
                 q_stride: Tuple[int, int] = None, act_layer: nn.Module = nn.GELU, window_size: int = 0):
        super().__init__()
        if isinstance(norm_layer, str):
            norm_layer = partial(getattr(nn, norm_layer), eps=1e-6)
        if drop_path > 0.0:
            raise NotImplementedError("drop_path > 0 is not used by the reference's model factory "
                                      "(sam2_model.py:12) and is not implemented")
        self.dim = dim
        self.dim_out = dim_out
        self.norm1 = norm_layer(dim)
        self.window_size = window_size
        self.pool, self.q_stride = None, q_stride
        if self.q_stride:
            self.pool = nn.MaxPool2d(kernel_size=q_stride, stride=q_stride, ceil_mode=False)
        self.attn = MultiScaleAttention(dim, dim_out, num_heads=num_heads, q_pool=self.pool)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim_out)
        self.mlp = MLP(dim_out, int(dim_out * mlp_ratio), dim_out, num_layers=2, activation=act_layer)
        if dim != dim_out:
            self.proj = nn.Linear(dim, dim_out)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Stand-alone block forward (inference), one HIP kernel per fused op.  The trunk does
        not call this -- `Hiera.forward` runs all blocks inside one library call."""
        from .lora_qkv import LoRA_qkv
        dt = _compute_dtype(None)
        B, H, W, d = x.shape
        q = self.dim_out
        pool = 2 if self.q_stride else 1
        x2 = x.float().contiguous().view(-1, d)
        xn = ops.layernorm_fwd(x2, self.norm1.weight, self.norm1.bias, dt)
        shortcut = x2
        if d != q:
            w = ops.pack_matrix(self.proj.weight, dt)
            shortcut = ops.linear(xn, w, self.proj.bias, out_f32=True)
            if pool == 2:
                shortcut = ops.maxpool2x2(shortcut.view(B, H, W, q)).view(-1, q)
        if isinstance(self.attn.qkv, LoRA_qkv):
            qkv = self.attn.qkv._forward_packed(xn)
            qkv_bias = self.attn.qkv.qkv.bias
        else:
            qkv = ops.linear(xn, ops.pack_matrix(self.attn.qkv.weight, dt), self.attn.qkv.bias)
            qkv_bias = self.attn.qkv.bias
        a = ops.attention_fwd(qkv.view(B, H, W, 3 * q), qkv_bias, self.window_size, pool, self.attn.num_heads)
        Ho, Wo = a.shape[1:3]
        y = ops.linear(a.view(-1, q), ops.pack_matrix(self.attn.proj.weight, dt), self.attn.proj.bias,
                       res=shortcut, out_f32=True)
        yn = ops.layernorm_fwd(y, self.norm2.weight, self.norm2.bias, dt)
        h = ops.linear(yn, ops.pack_matrix(self.mlp.layers[0].weight, dt), self.mlp.layers[0].bias, gelu=True)
        out = ops.linear(h, ops.pack_matrix(self.mlp.layers[1].weight, dt), self.mlp.layers[1].bias, res=y,
                         out_f32=True)
        return out.view(B, Ho, Wo, q)


# ---------------------------------------------------------------------------------------------
# plan management
# ---------------------------------------------------------------------------------------------
class _Plan:
    """One lsam_hiera_t + its packed-weight arena for a (dtype, image size, device)."""

    def __init__(self, trunk: "Hiera", dtype: torch.dtype, img_size: int, device: torch.device):
        from .lora_qkv import LoRA_qkv
        self.dtype, self.img_size, self.device = dtype, img_size, device
        n = len(trunk.blocks)
        n_lora = 0
        for blk in reversed(trunk.blocks):
            if isinstance(blk.attn.qkv, LoRA_qkv):
                n_lora += 1
            else:
                break
        for blk in list(trunk.blocks)[: n - n_lora]:
            if isinstance(blk.attn.qkv, LoRA_qkv):
                raise _C.LsamError("LoRA_qkv is supported on a contiguous run of trailing blocks only "
                                   "(the reference wraps trunk.blocks[-2:], lora_qkv.py:162)")
        self.n_lora = n_lora
        self.rank = trunk.blocks[-1].attn.qkv.linear_a_q.weight.shape[0] if n_lora else 0
        cfg = _C.HieraConfig()
        cfg.dtype = _C.dtype_code(dtype)
        cfg.img_size = img_size
        cfg.embed_dim = trunk.blocks[0].dim
        cfg.num_heads = trunk._num_heads0
        cfg.num_stages = len(trunk._stages)
        for i, s in enumerate(trunk._stages):
            cfg.stages[i] = s
            cfg.window_spec[i] = trunk.window_spec[i]
        gab = tuple(trunk.global_att_blocks or ())
        cfg.n_global = len(gab)
        for i, b in enumerate(gab):
            cfg.global_att_blocks[i] = b
        cfg.q_pool = len(trunk.q_pool_blocks)
        cfg.lora_rank = self.rank
        cfg.n_lora_blocks = n_lora
        h = C.c_void_p()
        check(lib().lsam_hiera_create(C.byref(cfg), C.byref(h)))
        self.h = h
        # cross-check the library's geometry against the module tree
        for i, blk in enumerate(trunk.blocks):
            v = [C.c_int() for _ in range(6)]
            check(lib().lsam_hiera_block_info(h, i, *[C.byref(x) for x in v]))
            got = tuple(x.value for x in v[:5])
            want = (blk.dim, blk.dim_out, blk.attn.num_heads, blk.window_size, 1 if blk.q_stride else 0)
            if got != want:
                raise _C.LsamError(f"block {i}: library geometry {got} != module {want}")
        self.stage_info = []
        for s in range(cfg.num_stages):
            c, g = C.c_int(), C.c_int()
            check(lib().lsam_hiera_stage_info(h, s, C.byref(c), C.byref(g)))
            self.stage_info.append((c.value, g.value))
        self.arena = torch.empty(lib().lsam_hiera_arena_bytes(h), dtype=torch.uint8, device=device)
        self.packed_sig = None
        self.lora_gen = 0        # bumped by every pack_lora: which factors the arena's LoRA slices hold
        self.ws_pool = {}
        self.bucket_hook = trunk._grad_bucket_hook

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().lsam_hiera_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def signature(self, trunk: "Hiera"):
        return (trunk._pack_epoch,) + tuple((p.data_ptr(), p._version) for p in trunk._frozen_params())

    def pack(self, trunk: "Hiera"):
        from .lora_qkv import LoRA_qkv
        bp = (_C.BlockParams * len(trunk.blocks))()
        keep = []

        def dev(p):
            t = p.detach()
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                t = t.to(device=self.device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        for i, blk in enumerate(trunk.blocks):
            qkv = blk.attn.qkv.qkv if isinstance(blk.attn.qkv, LoRA_qkv) else blk.attn.qkv
            b = bp[i]
            b.norm1_w, b.norm1_b = dev(blk.norm1.weight), dev(blk.norm1.bias)
            b.qkv_w, b.qkv_b = dev(qkv.weight), dev(qkv.bias)
            b.attn_proj_w, b.attn_proj_b = dev(blk.attn.proj.weight), dev(blk.attn.proj.bias)
            b.norm2_w, b.norm2_b = dev(blk.norm2.weight), dev(blk.norm2.bias)
            b.fc1_w, b.fc1_b = dev(blk.mlp.layers[0].weight), dev(blk.mlp.layers[0].bias)
            b.fc2_w, b.fc2_b = dev(blk.mlp.layers[1].weight), dev(blk.mlp.layers[1].bias)
            if blk.dim != blk.dim_out:
                b.proj_w, b.proj_b = dev(blk.proj.weight), dev(blk.proj.bias)
        g0 = self.img_size // 4
        with torch.no_grad():
            pos = trunk._get_pos_embed((g0, g0)).to(device=self.device, dtype=torch.float32).contiguous()
        keep.append(pos)
        check(lib().lsam_hiera_pack(self.h, bp, dev(trunk.patch_embed.proj.weight),
                                    dev(trunk.patch_embed.proj.bias), pos.data_ptr(), self.arena.data_ptr(),
                                    self.arena.numel(), stream()))
        self._keep = keep  # the pack kernels are asynchronous; keep temporaries alive
        self.packed_sig = self.signature(trunk)

    def pack_lora(self, lora_params: List[torch.Tensor]):
        if not self.n_lora:
            return
        lp = (_C.LoraParams * self.n_lora)()
        for i in range(self.n_lora):
            lp[i].a_q, lp[i].b_q, lp[i].a_v, lp[i].b_v = (ptr(t) for t in lora_params[4 * i: 4 * i + 4])
        check(lib().lsam_hiera_pack_lora(self.h, lp, self.arena.data_ptr(), stream()))
        self.lora_gen += 1

    def workspace(self, batch: int, with_grad: bool) -> torch.Tensor:
        key = (batch, with_grad)
        pool = self.ws_pool.setdefault(key, [])
        if pool:
            return pool.pop()
        nbytes = lib().lsam_hiera_workspace_bytes(self.h, batch, int(with_grad))
        return torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def release(self, batch: int, with_grad: bool, ws: torch.Tensor):
        pool = self.ws_pool.setdefault((batch, with_grad), [])
        if len(pool) < 2:
            pool.append(ws)


class _TrunkFn(torch.autograd.Function):
    """Autograd boundary of the trunk: inputs are the frames and the LoRA factors; the frozen
    weights are baked into the plan.  Only the last stage output is differentiable (the first
    blocks run under no_grad in the reference, hieradet.py:307-322)."""

    @staticmethod
    def forward(ctx, plan: _Plan, img: torch.Tensor, with_grad: bool, *lora):
        N = img.shape[0]
        lora_c = [t.detach().to(torch.float32).contiguous() for t in lora]
        plan.pack_lora(lora_c)
        outs = [torch.empty(N, g, g, c, dtype=torch.float32, device=img.device) for c, g in plan.stage_info]
        ws = plan.workspace(N, with_grad)
        arr = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        check(lib().lsam_hiera_forward(plan.h, img.data_ptr(), N, arr, plan.arena.data_ptr(), ws.data_ptr(),
                                       ws.numel(), int(with_grad), stream()))
        ctx.plan, ctx.N, ctx.with_grad = plan, N, with_grad
        ctx.lora_gen = plan.lora_gen
        # no zero tensors for the stage outputs nothing differentiates through (0.8 GB of fills per step at batch 16)
        ctx.set_materialize_grads(False)
        if with_grad:
            ctx.ws = ws
            ctx.lora_shapes = [tuple(t.shape) for t in lora]
            # this step's factors in one flat copy (one small launch): the backward restores them if another forward
            # re-packs the plan's LoRA slices in between
            ctx.lora_flat = torch.cat([t.reshape(-1) for t in lora_c]) if lora_c else None
            ctx.lora_params = lora   # the Parameters themselves: handed to the gradient-bucket hook
            ctx.mark_non_differentiable(*outs[:-1])
        else:
            plan.release(N, False, ws)
            ctx.mark_non_differentiable(*outs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        plan = ctx.plan
        if not ctx.with_grad:
            return (None, None, None) + (None,) * (4 * plan.n_lora)
        g = grads[-1]
        if g is None:
            return (None, None, None) + (None,) * len(ctx.lora_shapes)
        g = g.to(torch.float32).contiguous()
        out = [torch.empty(s, dtype=torch.float32, device=g.device) for s in ctx.lora_shapes]
        arr = (C.c_void_p * len(out))(*[o.data_ptr() for o in out])
        # The backward reads LoRA-dependent packed operands (B^T, the A columns of W_qkv^T, [A_q; A_v]) from the plan's
        # arena.  Another forward between this step's forward and its backward (a teacher / eval pass with other
        # factors, LoRABank.activate) re-packed them: restore this step's factors first, from the copy the forward kept.
        if plan.lora_gen != ctx.lora_gen:
            parts, o = [], 0
            for shp in ctx.lora_shapes:
                n = 1
                for v in shp:
                    n *= v
                parts.append(ctx.lora_flat[o:o + n].view(shp))
                o += n
            plan.pack_lora(parts)
            ctx.lora_gen = plan.lora_gen
        hook = plan.bucket_hook
        if hook is None:
            check(lib().lsam_hiera_backward(plan.h, g.data_ptr(), ctx.N, arr, plan.arena.data_ptr(),
                                            ctx.ws.data_ptr(), ctx.ws.numel(), stream()))
        else:
            # one LoRA block at a time, last first: each block's four gradients are handed to the hook
            # (dp.LoRAGradReducer starts their all-reduce) while the next block's backward is enqueued
            for bi in range(plan.n_lora - 1, -1, -1):
                check(lib().lsam_hiera_backward_block(plan.h, g.data_ptr(), ctx.N, bi, arr, plan.arena.data_ptr(),
                                                      ctx.ws.data_ptr(), ctx.ws.numel(), stream()))
                hook(list(ctx.lora_params[4 * bi: 4 * bi + 4]), out[4 * bi: 4 * bi + 4])
        plan.release(ctx.N, True, ctx.ws)
        ctx.ws = None
        return (None, None, None) + tuple(out)


class Hiera(nn.Module):
    """hieradet.py:169-352 (Reference: https://arxiv.org/abs/2306.00989)."""

    def __init__(self, embed_dim: int = 96, num_heads: int = 1, drop_path_rate: float = 0.0, q_pool: int = 3,
                 q_stride: Tuple[int, int] = (2, 2), stages: Tuple[int, ...] = (2, 3, 16, 3),
                 dim_mul: float = 2.0, head_mul: float = 2.0,
                 window_pos_embed_bkg_spatial_size: Tuple[int, int] = (14, 14),
                 window_spec: Tuple[int, ...] = (8, 4, 14, 7),
                 global_att_blocks: Tuple[int, ...] = (12, 16, 20), weights_path=None,
                 return_interm_layers=True):
        super().__init__()
        assert len(stages) == len(window_spec)
        if tuple(q_stride) != (2, 2) or dim_mul != 2.0 or head_mul != 2.0 or drop_path_rate != 0.0:
            raise NotImplementedError("liblsam implements the reference factory's Hiera family: q_stride (2,2), "
                                      "dim_mul = head_mul = 2, drop_path 0 (sam2_model.py:12, hieradet.py:174-200)")
        if not return_interm_layers:
            raise NotImplementedError("return_interm_layers=False is not used by the reference factory")
        self.window_spec = window_spec
        self._stages = tuple(stages)
        self._num_heads0 = num_heads
        depth = sum(stages)
        self.q_stride = q_stride
        self.stage_ends = [sum(stages[:i]) - 1 for i in range(1, len(stages) + 1)]
        assert 0 <= q_pool <= len(self.stage_ends[:-1])
        self.q_pool_blocks = [x + 1 for x in self.stage_ends[:-1]][:q_pool]
        self.return_interm_layers = return_interm_layers
        self.patch_embed = PatchEmbed(embed_dim=embed_dim)
        self.global_att_blocks = global_att_blocks
        self.window_pos_embed_bkg_spatial_size = window_pos_embed_bkg_spatial_size
        self.pos_embed = nn.Parameter(torch.zeros(1, embed_dim, *self.window_pos_embed_bkg_spatial_size))
        self.pos_embed_window = nn.Parameter(torch.zeros(1, embed_dim, self.window_spec[0], self.window_spec[0]))
        cur_stage = 1
        self.blocks = nn.ModuleList()
        for i in range(depth):
            dim_out = embed_dim
            window_size = self.window_spec[cur_stage - 1]  # lags by a block
            if self.global_att_blocks is not None:
                window_size = 0 if i in self.global_att_blocks else window_size
            if i - 1 in self.stage_ends:
                dim_out = int(embed_dim * dim_mul)
                num_heads = int(num_heads * head_mul)
                cur_stage += 1
            self.blocks.append(MultiScaleBlock(dim=embed_dim, dim_out=dim_out, num_heads=num_heads,
                                               q_stride=self.q_stride if i in self.q_pool_blocks else None,
                                               window_size=window_size))
            embed_dim = dim_out
        self.channel_list = [self.blocks[i].dim_out for i in self.stage_ends[::-1]]
        if weights_path is not None:
            chkpt = torch.load(weights_path, map_location="cpu", weights_only=True)
            self.load_state_dict(chkpt, strict=False)
        self.compute_dtype: Optional[torch.dtype] = None
        self._plans = {}
        self._grad_bucket_hook = None
        self._pack_epoch = 0
        # loaders rewrite parameters in place, some through `.data` (which bumps no version counter): re-pack after any load
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    def invalidate_packed(self):
        """Force the packed-weight arenas to be rebuilt on the next forward.  The trunk re-packs by itself when a frozen
        parameter is replaced or modified through autograd-visible in-place ops (data_ptr / _version change) and after
        load_state_dict; call this after editing frozen parameters through `.data` (EMA code, custom loaders,
        dp.LoRAGradReducer.broadcast_params does it for you)."""
        self._pack_epoch += 1

    def set_grad_bucket_hook(self, hook):
        """hook(params, grads) is called from the backward as soon as one LoRA block's gradients are
        enqueued (last block first); None restores the single-call backward.  Used by
        dp.LoRAGradReducer.attach to overlap the gradient all-reduce with the rest of the backward."""
        self._grad_bucket_hook = hook
        for plan in self._plans.values():
            plan.bucket_hook = hook

    # -- constant per shape: computed once per plan (hieradet.py:273-281) --------------------
    def _get_pos_embed(self, hw: Tuple[int, int]) -> torch.Tensor:
        h, w = hw
        window_embed = self.pos_embed_window
        pos_embed = F.interpolate(self.pos_embed, size=(h, w), mode="bicubic")
        pos_embed = pos_embed + window_embed.tile([x // y for x, y in zip(pos_embed.shape, window_embed.shape)])
        return pos_embed.permute(0, 2, 3, 1)

    def _frozen_params(self):
        for n, p in self.named_parameters():
            if "linear_a_" in n or "linear_b_" in n:
                continue
            yield p

    def _lora_params(self, plan: _Plan) -> List[torch.Tensor]:
        out = []
        for blk in list(self.blocks)[len(self.blocks) - plan.n_lora:]:
            m = blk.attn.qkv
            out += [m.linear_a_q.weight, m.linear_b_q.weight, m.linear_a_v.weight, m.linear_b_v.weight]
        return out

    def _plan_for(self, x: torch.Tensor) -> _Plan:
        if not x.is_cuda:
            raise _C.LsamError("Hiera runs on the MI355X only: move the model and the frames to 'cuda' "
                               "(there is no CPU or PyTorch fallback)")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != x.shape[3]:
            raise _C.LsamError(f"expected frames [N,3,S,S], got {tuple(x.shape)}")
        dt = _compute_dtype(self.compute_dtype)
        wrapped = tuple(type(b.attn.qkv).__name__ for b in self.blocks)
        key = (dt, x.shape[-1], x.device, wrapped)
        plan = self._plans.get(key)
        if plan is None:
            # plans of a superseded module tree (the model was wrapped after a first forward) are dropped with their arenas
            for k in [k for k in self._plans if k[3] != wrapped]:
                del self._plans[k]
            plan = _Plan(self, dt, x.shape[-1], x.device)
            self._plans[key] = plan
        if plan.packed_sig != plan.signature(self):
            plan.pack(self)
        return plan

    def _run(self, x: torch.Tensor, allow_grad: bool) -> List[torch.Tensor]:
        plan = self._plan_for(x)
        lora = self._lora_params(plan)
        if torch.is_grad_enabled() and allow_grad:
            for p in self._frozen_params():
                if p.requires_grad:
                    raise _C.LsamError(
                        "a non-LoRA trunk parameter requires grad: liblsam produces gradients for the LoRA "
                        "A/B factors only (freeze the encoder first, as wrap_image_encoder_lora does)")
        with_grad = allow_grad and torch.is_grad_enabled() and any(p.requires_grad for p in lora)
        img = x.detach().to(torch.float32).contiguous()
        outs = _TrunkFn.apply(plan, img, with_grad, *lora)
        return [o.permute(0, 3, 1, 2) for o in outs]

    def get_embedding(self, x: torch.Tensor):
        """hieradet.py:284-302: every block under no_grad."""
        return self._run(x, allow_grad=False)

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        """hieradet.py:306-334: blocks[:-2] under no_grad, the LoRA blocks with autograd."""
        return self._run(x, allow_grad=True)

    def get_num_layers(self) -> int:
        return len(self.blocks)
