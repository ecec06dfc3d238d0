"""CPU oracle for the LoRA-SAM2 Hiera image-encoder hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch (fp32, CPU) functional
restatement of the reference algorithm.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product path (``image_segmentation_amd``) never does.

Parity status: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned by fixtures generated from the
reference itself, imported in the build container by
``oracle/make_goldens.py`` (fixtures in ``tests/golden/``; checked by
``tests/test_oracle_vs_golden.py``).

Every function cites the reference file:line it restates (paths relative to
the reference checkout).  Parameters travel as a flat ``dict[str, Tensor]``
using the reference's state_dict key names relative to the trunk / neck.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# geometry  (SAM2/sam2/sam2/modeling/backbones/hieradet.py:203-266)
# --------------------------------------------------------------------------
def block_specs(
    embed_dim: int = 96,
    num_heads: int = 1,
    q_pool: int = 3,
    q_stride: Tuple[int, int] = (2, 2),
    stages: Sequence[int] = (2, 3, 16, 3),
    dim_mul: float = 2.0,
    head_mul: float = 2.0,
    window_spec: Sequence[int] = (8, 4, 14, 7),
    global_att_blocks: Optional[Sequence[int]] = (12, 16, 20),
):
    """Per-block geometry exactly as the Hiera constructor derives it.

    Note the window size "lags" a block: the first block of a stage still uses
    the previous stage's window (hieradet.py:237-248)."""
    depth = sum(stages)
    stage_ends = [sum(stages[:i]) - 1 for i in range(1, len(stages) + 1)]
    q_pool_blocks = [x + 1 for x in stage_ends[:-1]][:q_pool]
    cur_stage = 1
    blocks = []
    for i in range(depth):
        dim_out = embed_dim
        window_size = window_spec[cur_stage - 1]
        if global_att_blocks is not None:
            window_size = 0 if i in global_att_blocks else window_size
        if i - 1 in stage_ends:
            dim_out = int(embed_dim * dim_mul)
            num_heads = int(num_heads * head_mul)
            cur_stage += 1
        blocks.append(
            dict(
                dim=embed_dim,
                dim_out=dim_out,
                num_heads=num_heads,
                window_size=window_size,
                q_stride=tuple(q_stride) if i in q_pool_blocks else None,
            )
        )
        embed_dim = dim_out
    channel_list = [blocks[i]["dim_out"] for i in stage_ends[::-1]]
    return dict(blocks=blocks, stage_ends=stage_ends, q_pool_blocks=q_pool_blocks,
                channel_list=channel_list, window_spec=tuple(window_spec))


# --------------------------------------------------------------------------
# windows  (SAM2/sam2/sam2/modeling/backbones/utils.py:16-60)
# --------------------------------------------------------------------------
def window_partition(x: torch.Tensor, ws: int):
    B, H, W, C = x.shape
    pad_h = (ws - H % ws) % ws
    pad_w = (ws - W % ws) % ws
    if pad_h > 0 or pad_w > 0:
        x = F.pad(x, (0, 0, 0, pad_w, 0, pad_h))  # zero padding, after LN1
    Hp, Wp = H + pad_h, W + pad_w
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    windows = x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)
    return windows, (Hp, Wp)


def window_unpartition(windows: torch.Tensor, ws: int, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = windows.shape[0] // (Hp * Wp // ws // ws)
    x = windows.reshape(B, Hp // ws, Wp // ws, ws, ws, -1)
    x = x.permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    if Hp > H or Wp > W:
        x = x[:, :H, :W, :]
    return x


def max_pool_nhwc(x: torch.Tensor, stride: Tuple[int, int]) -> torch.Tensor:
    """do_pool (hieradet.py:25-36) with nn.MaxPool2d(k=s=q_stride, ceil_mode=False)
    (hieradet.py:110-112)."""
    x = x.permute(0, 3, 1, 2)
    x = F.max_pool2d(x, kernel_size=stride, stride=stride, ceil_mode=False)
    return x.permute(0, 2, 3, 1)


# --------------------------------------------------------------------------
# qkv (+LoRA)  (hieradet.py:59 ; lora_qkv.py:102-109)
# --------------------------------------------------------------------------
def _qkv_keys(p: Params, pre: str):
    """After wrap_image_encoder_lora the base Linear moves from
    ``attn.qkv.{weight,bias}`` to ``attn.qkv.qkv.{weight,bias}``
    (lora_qkv.py:94,177).  Accept both spellings."""
    if pre + "attn.qkv.qkv.weight" in p:
        return pre + "attn.qkv.qkv.weight", pre + "attn.qkv.qkv.bias"
    return pre + "attn.qkv.weight", pre + "attn.qkv.bias"


def qkv_lora(x: torch.Tensor, p: Params, pre: str) -> torch.Tensor:
    """qkv = W x + b;  qkv[..., :d] += B_q(A_q x);  qkv[..., -d:] += B_v(A_v x)
    with d = in_features; LoRA linears have no bias and no alpha/r scaling
    (lora_qkv.py:99-107)."""
    wk, bk = _qkv_keys(p, pre)
    qkv = F.linear(x, p[wk], p[bk])
    la = pre + "attn.qkv.linear_a_q.weight"
    if la in p:
        d = p[wk].shape[1]
        q_ba = F.linear(F.linear(x, p[pre + "attn.qkv.linear_a_q.weight"]),
                        p[pre + "attn.qkv.linear_b_q.weight"])
        v_ba = F.linear(F.linear(x, p[pre + "attn.qkv.linear_a_v.weight"]),
                        p[pre + "attn.qkv.linear_b_v.weight"])
        n = qkv.shape[-1]
        # out-of-place restatement of the two in-place slice adds
        add = torch.zeros_like(qkv)
        add[..., :d] = add[..., :d] + q_ba
        add[..., n - d:] = add[..., n - d:] + v_ba
        qkv = qkv + add
    return qkv


# --------------------------------------------------------------------------
# attention  (hieradet.py:56-81)
# --------------------------------------------------------------------------
def attention(x: torch.Tensor, p: Params, pre: str, num_heads: int,
              q_stride: Optional[Tuple[int, int]]) -> torch.Tensor:
    B, H, W, _ = x.shape
    qkv = qkv_lora(x, p, pre).reshape(B, H * W, 3, num_heads, -1)
    q, k, v = torch.unbind(qkv, 2)
    if q_stride:
        q = max_pool_nhwc(q.reshape(B, H, W, -1), q_stride)
        H, W = q.shape[1:3]
        q = q.reshape(B, H * W, num_heads, -1)
    q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    hd = q.shape[-1]
    # F.scaled_dot_product_attention defaults: scale 1/sqrt(hd), no mask, p=0
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    o = torch.softmax(s, dim=-1) @ v
    o = o.transpose(1, 2).reshape(B, H, W, -1)
    return F.linear(o, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"])


# --------------------------------------------------------------------------
# block  (hieradet.py:134-166; LN eps 1e-6 :100; exact-erf GELU :94; MLP
#         sam2_utils.py:131-136)
# --------------------------------------------------------------------------
def block_forward(x: torch.Tensor, p: Params, pre: str, spec: dict) -> torch.Tensor:
    dim, dim_out = spec["dim"], spec["dim_out"]
    ws, qs = spec["window_size"], spec["q_stride"]
    shortcut = x
    x = F.layer_norm(x, (dim,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps=1e-6)
    if dim != dim_out:
        shortcut = F.linear(x, p[pre + "proj.weight"], p[pre + "proj.bias"])
        if qs:
            shortcut = max_pool_nhwc(shortcut, qs)
    window_size = ws
    if ws > 0:
        H, W = x.shape[1], x.shape[2]
        x, pad_hw = window_partition(x, ws)
    x = attention(x, p, pre, spec["num_heads"], qs)
    if qs:
        window_size = ws // qs[0]
        H, W = shortcut.shape[1:3]
        pad_h = (window_size - H % window_size) % window_size
        pad_w = (window_size - W % window_size) % window_size
        pad_hw = (H + pad_h, W + pad_w)
    if ws > 0:
        x = window_unpartition(x, window_size, pad_hw, (H, W))
    x = shortcut + x
    y = F.layer_norm(x, (dim_out,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps=1e-6)
    y = F.linear(y, p[pre + "mlp.layers.0.weight"], p[pre + "mlp.layers.0.bias"])
    y = F.gelu(y)  # nn.GELU() default = exact erf
    y = F.linear(y, p[pre + "mlp.layers.1.weight"], p[pre + "mlp.layers.1.bias"])
    return x + y


# --------------------------------------------------------------------------
# trunk  (hieradet.py:273-334; PatchEmbed backbones/utils.py:85-93)
# --------------------------------------------------------------------------
def pos_embed(p: Params, hw: Tuple[int, int]) -> torch.Tensor:
    h, w = hw
    window_embed = p["pos_embed_window"]
    pe = F.interpolate(p["pos_embed"], size=(h, w), mode="bicubic")
    pe = pe + window_embed.tile([x // y for x, y in zip(pe.shape, window_embed.shape)])
    return pe.permute(0, 2, 3, 1)


def patch_embed(p: Params, img: torch.Tensor) -> torch.Tensor:
    x = F.conv2d(img, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"],
                 stride=(4, 4), padding=(3, 3))
    return x.permute(0, 2, 3, 1)


def trunk_forward(p: Params, img: torch.Tensor, spec: dict, n_grad_blocks: int = 2,
                  collect_blocks: Optional[list] = None) -> List[torch.Tensor]:
    """Hiera.forward (hieradet.py:306-334): patch/pos embed and blocks[:-2] run
    under no_grad, blocks[-2:] with autograd.  n_grad_blocks=0 gives
    Hiera.get_embedding (hieradet.py:284-302).  Returns NCHW stage outputs."""
    blocks = spec["blocks"]
    stage_ends = spec["stage_ends"]
    n = len(blocks)
    outs = []
    with torch.no_grad():
        x = patch_embed(p, img)
        x = x + pos_embed(p, x.shape[1:3])
    for i in range(n):
        pre = f"blocks.{i}."
        if i < n - n_grad_blocks:
            with torch.no_grad():
                x = block_forward(x, p, pre, blocks[i])
        else:
            x = block_forward(x, p, pre, blocks[i])
        if collect_blocks is not None:
            collect_blocks.append(x)
        if i in stage_ends:
            outs.append(x.permute(0, 3, 1, 2))
    return outs


# --------------------------------------------------------------------------
# neck + encoder  (image_encoder.py:29-42,102-134; position_encoding.py:89-124)
# --------------------------------------------------------------------------
def sine_pe(B: int, H: int, W: int, num_pos_feats: int = 256, temperature: float = 10000.0,
            scale: float = 2 * math.pi) -> torch.Tensor:
    npf = num_pos_feats // 2
    y_embed = torch.arange(1, H + 1, dtype=torch.float32).view(1, -1, 1).repeat(B, 1, W)
    x_embed = torch.arange(1, W + 1, dtype=torch.float32).view(1, 1, -1).repeat(B, H, 1)
    eps = 1e-6
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / npf)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def neck_forward(p: Params, xs: List[torch.Tensor], fpn_top_down_levels=(2, 3),
                 d_model: int = 256, pre: str = "neck."):
    """FpnNeck.forward with fpn_interp_model='nearest', fuse_type='sum'
    (sam2_model.py:18-24).  convs are indexed convs[n-i] (image_encoder.py:114)."""
    n = len(xs) - 1
    out = [None] * len(xs)
    pos = [None] * len(xs)
    prev = None
    for i in range(n, -1, -1):
        w = p[f"{pre}convs.{n - i}.conv.weight"]
        b = p[f"{pre}convs.{n - i}.conv.bias"]
        lateral = F.conv2d(xs[i], w, b)
        if i in fpn_top_down_levels and prev is not None:
            top_down = F.interpolate(prev.to(torch.float32), scale_factor=2.0, mode="nearest")
            prev = lateral + top_down
        else:
            prev = lateral
        out[i] = prev
        pos[i] = sine_pe(prev.shape[0], prev.shape[2], prev.shape[3], d_model).to(prev.dtype)
    return out, pos


def image_encoder_forward(p: Params, img: torch.Tensor, spec: dict, scalp: int = 1,
                          n_grad_blocks: int = 2, trunk_pre: str = "trunk."):
    """ImageEncoder.forward (image_encoder.py:29-42).  ``p`` uses keys
    ``trunk.*`` and ``neck.*``."""
    tp = {k[len(trunk_pre):]: v for k, v in p.items() if k.startswith(trunk_pre)}
    feats, pos = neck_forward(p, trunk_forward(tp, img, spec, n_grad_blocks))
    if scalp > 0:
        feats, pos = feats[:-scalp], pos[:-scalp]
    return {"vision_features": feats[-1], "vision_pos_enc": pos, "backbone_fpn": feats}


def fpn_loss(out: dict) -> torch.Tensor:
    """The fixed scalar loss used for gradient goldens and the bench:
    sum over FPN levels of mean(level**2)."""
    return sum((f.float() ** 2).mean() for f in out["backbone_fpn"])
