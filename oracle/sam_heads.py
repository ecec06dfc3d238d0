"""CPU restatement of the reference's SAM prompt encoder + mask decoder for ONE click on an initial
frame -- TEST INFRASTRUCTURE ONLY (see oracle/hiera_oracle.py's header): it exists so that the
north-star's "predicted masks reach IoU >= 0.99 against the reference masks" can be checked on the
GPU box, where the reference cannot travel.  It turns image-encoder outputs (the product) into masks
exactly as the reference does; it is not part of the product path and nothing under
image_segmentation_amd/ imports it.

What it follows (all under /root/reference/SAM2/sam2/sam2/modeling/):
  SAM2Base.forward_image            sam2_base.py:467-479   conv_s0 / conv_s1 on FPN levels 0 and 1
  _prepare_memory_conditioned_features  sam2_base.py:669-674   first frame: + no_mem_embed
  SAM2Base._forward_sam_heads       sam2_base.py:257-413   prompt encoder -> mask decoder -> object gating,
                                                           bilinear x4 upsample, best-IoU mask
  PromptEncoder                     sam/prompt_encoder.py:87-202 (_embed_points, get_dense_pe, no_mask_embed)
  PositionEmbeddingRandom           position_encoding.py:133-176
  MaskDecoder.predict_masks/forward sam/mask_decoder.py:110-250 (multimask_output=True, high-res features)
  TwoWayTransformer / Attention     sam/transformer.py:17-265
  MLP / LayerNorm2d                 sam2_utils.py:110-153
Model options are the reference factory's (sam2_model.py:62-88): hidden 256, image 1024, stride 16,
use_high_res_features_in_sam, pred_obj_scores(+_mlp), iou_prediction_use_sigmoid, multimask output.

Pinned by tests/golden/masks_bplus_r4.npz (oracle/make_mask_goldens.py runs the reference's own
SAM2Base on the same synthetic weights) in tests/test_masks_cpu.py.
"""
from __future__ import annotations

import math
import os
import sys
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_segmentation_amd.synth import _gen  # noqa: E402  (seeded generator only)

D = 256            # hidden_dim = neck d_model
IMAGE = 1024
EMB = IMAGE // 16  # sam_image_embedding_size
NO_OBJ_SCORE = -1024.0   # sam2_base.py:18
Params = Dict[str, torch.Tensor]


# ---------------------------------------------------------------------------------------------
# parameters: names/shapes as the reference constructors create them; values from the seeded generator
# ---------------------------------------------------------------------------------------------
def head_param_shapes() -> List[Tuple[str, Tuple[int, ...]]]:
    out: List[Tuple[str, Tuple[int, ...]]] = [("no_mem_embed", (1, 1, D))]
    pe = "sam_prompt_encoder."
    out += [(pe + "pe_layer.positional_encoding_gaussian_matrix", (2, D // 2))]
    out += [(pe + f"point_embeddings.{i}.weight", (1, D)) for i in range(4)]
    out += [(pe + "not_a_point_embed.weight", (1, D)), (pe + "no_mask_embed.weight", (1, D))]
    md = "sam_mask_decoder."

    def attn(pre, internal):
        return [(pre + "q_proj.weight", (internal, D)), (pre + "q_proj.bias", (internal,)),
                (pre + "k_proj.weight", (internal, D)), (pre + "k_proj.bias", (internal,)),
                (pre + "v_proj.weight", (internal, D)), (pre + "v_proj.bias", (internal,)),
                (pre + "out_proj.weight", (D, internal)), (pre + "out_proj.bias", (D,))]

    def mlp(pre, dims):
        r = []
        for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
            r += [(pre + f"layers.{i}.weight", (b, a)), (pre + f"layers.{i}.bias", (b,))]
        return r

    for l in range(2):
        t = md + f"transformer.layers.{l}."
        out += attn(t + "self_attn.", D)
        out += attn(t + "cross_attn_token_to_image.", D // 2)
        out += attn(t + "cross_attn_image_to_token.", D // 2)
        out += mlp(t + "mlp.", [D, 2048, D])
        for n in ("norm1", "norm2", "norm3", "norm4"):
            out += [(t + n + ".weight", (D,)), (t + n + ".bias", (D,))]
    out += attn(md + "transformer.final_attn_token_to_image.", D // 2)
    out += [(md + "transformer.norm_final_attn.weight", (D,)), (md + "transformer.norm_final_attn.bias", (D,))]
    out += [(md + "iou_token.weight", (1, D)), (md + "mask_tokens.weight", (4, D)), (md + "obj_score_token.weight", (1, D))]
    out += [(md + "output_upscaling.0.weight", (D, D // 4, 2, 2)), (md + "output_upscaling.0.bias", (D // 4,)),
            (md + "output_upscaling.1.weight", (D // 4,)), (md + "output_upscaling.1.bias", (D // 4,)),
            (md + "output_upscaling.3.weight", (D // 4, D // 8, 2, 2)), (md + "output_upscaling.3.bias", (D // 8,))]
    out += [(md + "conv_s0.weight", (D // 8, D, 1, 1)), (md + "conv_s0.bias", (D // 8,)),
            (md + "conv_s1.weight", (D // 4, D, 1, 1)), (md + "conv_s1.bias", (D // 4,))]
    for i in range(4):
        out += mlp(md + f"output_hypernetworks_mlps.{i}.", [D, D, D, D // 8])
    out += mlp(md + "iou_prediction_head.", [D, D, D, 4])
    out += mlp(md + "pred_obj_score_head.", [D, D, D, 1])
    return out


def synth_head_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    """Seeded synthetic value for a head tensor (no checkpoint exists offline).  Random heads must be
    put in the regime trained heads work in, or the masks say nothing about the encoder:
      * the object-score bias is positive, so that the click yields an object (otherwise the reference
        overwrites every mask with NO_OBJ_SCORE, sam2_base.py:359-368);
      * the IoU head's output biases separate the three candidates, so that "best mask" is not a coin
        flip between two near-equal estimates;
      * q/k projections are scaled by 0.1: the synthetic FPN features have std ~5, which with
        unit-variance projections drives every attention softmax to one-hot and makes the token stream
        chaotic (a 0.4 % feature perturbation then moves 15-25 % of the mask); the high-resolution skip
        convolutions are scaled by 0.1 so that the mask follows the 64x64 embedding, as a trained decoder's does.
    With these scales a 0.4-0.8 % i.i.d. perturbation of all FPN levels keeps IoU at 0.995-0.998
    (oracle/make_mask_goldens.py prints the study), so IoU >= 0.99 is a meaningful bar for the bf16 path."""
    g = _gen("heads." + name, seed)
    leaf = name.split(".")[-1]
    if name.endswith("positional_encoding_gaussian_matrix"):
        return torch.randn(shape, generator=g)                       # scale 1.0, position_encoding.py:140-145
    if name == "sam_mask_decoder.pred_obj_score_head.layers.2.bias":
        return torch.full(shape, 4.0)
    if name == "sam_mask_decoder.iou_prediction_head.layers.2.bias":
        return torch.tensor([0.0, 3.0, -3.0, -3.0])
    if ".q_proj." in name or ".k_proj." in name or ".conv_s0." in name or ".conv_s1." in name:
        if leaf == "bias":
            return 0.01 * torch.randn(shape, generator=g)
        fan = 1
        for d_ in shape[1:]:
            fan *= d_
        return 0.1 * torch.randn(shape, generator=g) / math.sqrt(fan)
    if leaf == "weight" and len(shape) == 1:                          # LayerNorm / LayerNorm2d gains
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf == "bias":
        return 0.1 * torch.randn(shape, generator=g)
    if "embed" in name or "token" in name:                            # learned tokens / embeddings
        return 0.5 * torch.randn(shape, generator=g)
    if name.endswith(("output_upscaling.0.weight", "output_upscaling.3.weight")):   # ConvTranspose2d [in, out, 2, 2]
        return torch.randn(shape, generator=g) / math.sqrt(shape[0])
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) / math.sqrt(fan_in)


def make_head_params(seed: int = 0) -> Params:
    return {n: synth_head_tensor(n, s, seed) for n, s in head_param_shapes()}


def decoder_lora_params(rank: int = 4, seed: int = 0) -> Params:
    """What the reference's `wrap_decoder_lora` (lora_qkv.py:112-143) ADDS to the mask decoder, with seeded values: per
    transformer layer the four factors A_q [r, D], B_q [D, r], A_v [r, D], B_v [D, r] (non-zero B, so that every gradient
    is non-trivial) and the FRESH q / k / v / out projections `Attention_LoRA.__init__` creates in place of the wrapped
    ones (lora_qkv.py:26-29); the latter take the place of the `self_attn.*_proj.*` entries of make_head_params."""
    out: Params = {}
    for l in range(2):
        pre = f"sam_mask_decoder.transformer.layers.{l}.self_attn."
        for n, shape in (("A_q.weight", (rank, D)), ("B_q.weight", (D, rank)), ("A_v.weight", (rank, D)), ("B_v.weight", (D, rank))):
            g = _gen("decoder_lora." + pre + n, seed)
            out[pre + n] = torch.randn(shape, generator=g) * (1.0 / math.sqrt(D) if n[0] == "A" else 0.05)
        for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
            for leaf, shape in (("weight", (D, D)), ("bias", (D,))):
                out[pre + f"{proj}.{leaf}"] = synth_head_tensor("fresh." + pre + f"{proj}.{leaf}", shape, seed)
    return out


def functional_weights(B: int, seed: int = 4242):
    """Fixed random weights (wl, wi, wo) of the scalar functional the decoder-LoRA gradient goldens differentiate:
    L = sum(low_res_multimasks * wl) + sum(ious * wi) + sum(object_score_logits * wo)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 3, 256, 256, generator=g) / 256.0, torch.randn(B, 3, generator=g), torch.randn(B, 1, generator=g))


def synthetic_fpn(B: int, seed: int = 3):
    """Encoder-like pyramid: std ~5 smooth-ish features (the regime synth_head_tensor's scales were chosen for)."""
    g = torch.Generator().manual_seed(seed)
    f2 = 5 * torch.randn(B, 256, 64, 64, generator=g)
    f2 = F.avg_pool2d(F.pad(f2, (2, 2, 2, 2), mode="replicate"), 5, 1) * 2.5       # spatially correlated
    f1 = F.interpolate(f2, scale_factor=2, mode="nearest") + torch.randn(B, 256, 128, 128, generator=g)
    f0 = F.interpolate(f1, scale_factor=2, mode="nearest") + torch.randn(B, 256, 256, 256, generator=g)
    return f0, f1, f2


# ---------------------------------------------------------------------------------------------
# prompt encoder  (sam/prompt_encoder.py, position_encoding.py:133-176)
# ---------------------------------------------------------------------------------------------
def _pe_encoding(G: torch.Tensor, coords01: torch.Tensor) -> torch.Tensor:
    c = 2 * coords01 - 1
    c = c @ G
    c = 2 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(hp: Params) -> torch.Tensor:
    """PromptEncoder.get_dense_pe: [1, D, EMB, EMB]."""
    G = hp["sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    grid = torch.ones((EMB, EMB), dtype=torch.float32)
    y = (grid.cumsum(dim=0) - 0.5) / EMB
    x = (grid.cumsum(dim=1) - 0.5) / EMB
    return _pe_encoding(G, torch.stack([x, y], dim=-1)).permute(2, 0, 1).unsqueeze(0)


def embed_points(hp: Params, coords: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """_embed_points with pad=True (no box given): coords [B,P,2] in pixels (x, y), labels [B,P]."""
    pe = "sam_prompt_encoder."
    pts = coords + 0.5
    pts = torch.cat([pts, torch.zeros((pts.shape[0], 1, 2))], dim=1)
    lab = torch.cat([labels, -torch.ones((labels.shape[0], 1), dtype=labels.dtype)], dim=1)
    c = pts.clone()
    c[:, :, 0] = c[:, :, 0] / IMAGE
    c[:, :, 1] = c[:, :, 1] / IMAGE
    e = _pe_encoding(hp[pe + "pe_layer.positional_encoding_gaussian_matrix"], c.to(torch.float))
    e = torch.where((lab == -1).unsqueeze(-1), torch.zeros_like(e) + hp[pe + "not_a_point_embed.weight"], e)
    for i in range(4):
        e = torch.where((lab == i).unsqueeze(-1), e + hp[pe + f"point_embeddings.{i}.weight"], e)
    return e


# ---------------------------------------------------------------------------------------------
# two-way transformer  (sam/transformer.py)
# ---------------------------------------------------------------------------------------------
def _mlp(hp: Params, pre: str, x: torch.Tensor, n: int, sigmoid: bool = False) -> torch.Tensor:
    for i in range(n):
        x = F.linear(x, hp[pre + f"layers.{i}.weight"], hp[pre + f"layers.{i}.bias"])
        if i < n - 1:
            x = F.relu(x)
    return torch.sigmoid(x) if sigmoid else x


def _attention(hp: Params, pre: str, q, k, v, nh: int = 8) -> torch.Tensor:
    """sam/transformer.py:233-265; with `A_q / B_q / A_v / B_v` entries under `pre` it is Attention_LoRA.forward
    (lora_qkv.py:49-67): q = q_proj(q) + B_q(A_q(q)), v = v_proj(v) + B_v(A_v(v)), no scaling, no bias on the factors."""
    q_in, v_in = q, v
    q = F.linear(q, hp[pre + "q_proj.weight"], hp[pre + "q_proj.bias"])
    k = F.linear(k, hp[pre + "k_proj.weight"], hp[pre + "k_proj.bias"])
    v = F.linear(v, hp[pre + "v_proj.weight"], hp[pre + "v_proj.bias"])
    if pre + "A_q.weight" in hp:
        q = q + F.linear(F.linear(q_in, hp[pre + "A_q.weight"]), hp[pre + "B_q.weight"])
        v = v + F.linear(F.linear(v_in, hp[pre + "A_v.weight"]), hp[pre + "B_v.weight"])

    def heads(x):
        b, n, c = x.shape
        return x.reshape(b, n, nh, c // nh).transpose(1, 2)

    o = F.scaled_dot_product_attention(heads(q), heads(k), heads(v))
    b, _, n, c = o.shape
    o = o.transpose(1, 2).reshape(b, n, nh * c)
    return F.linear(o, hp[pre + "out_proj.weight"], hp[pre + "out_proj.bias"])


def _ln(hp: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), hp[pre + ".weight"], hp[pre + ".bias"], 1e-5)


def two_way_transformer(hp: Params, image_embedding, image_pe, point_embedding):
    t = "sam_mask_decoder.transformer."
    keys = image_embedding.flatten(2).permute(0, 2, 1)
    key_pe = image_pe.flatten(2).permute(0, 2, 1)
    queries, query_pe = point_embedding, point_embedding
    for l in range(2):
        pre = t + f"layers.{l}."
        if l == 0:   # skip_first_layer_pe
            queries = _attention(hp, pre + "self_attn.", queries, queries, queries)
        else:
            q = queries + query_pe
            queries = queries + _attention(hp, pre + "self_attn.", q, q, queries)
        queries = _ln(hp, pre + "norm1", queries)
        q, k = queries + query_pe, keys + key_pe
        queries = queries + _attention(hp, pre + "cross_attn_token_to_image.", q, k, keys)
        queries = _ln(hp, pre + "norm2", queries)
        queries = queries + _mlp(hp, pre + "mlp.", queries, 2)
        queries = _ln(hp, pre + "norm3", queries)
        q, k = queries + query_pe, keys + key_pe
        keys = keys + _attention(hp, pre + "cross_attn_image_to_token.", k, q, queries)
        keys = _ln(hp, pre + "norm4", keys)
    q, k = queries + point_embedding, keys + key_pe
    queries = queries + _attention(hp, t + "final_attn_token_to_image.", q, k, keys)
    return _ln(hp, t + "norm_final_attn", queries), keys


def _layernorm2d(x, w, b, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


# ---------------------------------------------------------------------------------------------
# mask decoder + _forward_sam_heads
# ---------------------------------------------------------------------------------------------
def high_res_features(hp: Params, fpn0: torch.Tensor, fpn1: torch.Tensor):
    md = "sam_mask_decoder."
    return (F.conv2d(fpn0, hp[md + "conv_s0.weight"], hp[md + "conv_s0.bias"]),
            F.conv2d(fpn1, hp[md + "conv_s1.weight"], hp[md + "conv_s1.bias"]))


def predict_masks(hp: Params, pix, image_pe, sparse, dense, feat_s0, feat_s1):
    md = "sam_mask_decoder."
    out_tokens = torch.cat([hp[md + "obj_score_token.weight"], hp[md + "iou_token.weight"], hp[md + "mask_tokens.weight"]], 0)
    tokens = torch.cat((out_tokens.unsqueeze(0).expand(sparse.size(0), -1, -1), sparse), dim=1)
    src = pix + dense
    pos_src = torch.repeat_interleave(image_pe, tokens.shape[0], dim=0)
    b, c, h, w = src.shape
    hs, src = two_way_transformer(hp, src, pos_src, tokens)
    iou_token_out = hs[:, 1, :]
    mask_tokens_out = hs[:, 2:6, :]
    src = src.transpose(1, 2).view(b, c, h, w)
    up = F.conv_transpose2d(src, hp[md + "output_upscaling.0.weight"], hp[md + "output_upscaling.0.bias"], stride=2)
    up = F.gelu(_layernorm2d(up + feat_s1, hp[md + "output_upscaling.1.weight"], hp[md + "output_upscaling.1.bias"]))
    up = F.gelu(F.conv_transpose2d(up, hp[md + "output_upscaling.3.weight"], hp[md + "output_upscaling.3.bias"], stride=2) + feat_s0)
    hyper = torch.stack([_mlp(hp, md + f"output_hypernetworks_mlps.{i}.", mask_tokens_out[:, i, :], 3) for i in range(4)], 1)
    b, c, h, w = up.shape
    masks = (hyper @ up.view(b, c, h * w)).view(b, -1, h, w)
    iou_pred = _mlp(hp, md + "iou_prediction_head.", iou_token_out, 3, sigmoid=True)
    obj = _mlp(hp, md + "pred_obj_score_head.", hs[:, 0, :], 3)
    return masks, iou_pred, obj


@torch.no_grad()
def forward_sam_heads(hp: Params, backbone_fpn, coords: torch.Tensor, labels: torch.Tensor) -> dict:
    """backbone_fpn: the image encoder's three levels ([B,256,256,256], [B,256,128,128], [B,256,64,64]),
    any float dtype / device; the heads run in fp32 on the CPU.  One positive click per frame."""
    f0, f1, f2 = (t.detach().float().cpu() for t in backbone_fpn)
    feat_s0, feat_s1 = high_res_features(hp, f0, f1)
    pix = f2 + hp["no_mem_embed"].view(1, D, 1, 1)
    sparse = embed_points(hp, coords.float(), labels)
    dense = hp["sam_prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(f2.shape[0], -1, EMB, EMB)
    masks, iou, obj = predict_masks(hp, pix, dense_pe(hp), sparse, dense, feat_s0, feat_s1)
    masks, iou = masks[:, 1:], iou[:, 1:]                     # multimask_output=True
    low = torch.where((obj > 0)[:, None, None], masks, torch.tensor(NO_OBJ_SCORE)).float()
    high = F.interpolate(low, size=(IMAGE, IMAGE), mode="bilinear", align_corners=False)
    best = torch.argmax(iou, dim=-1)
    bi = torch.arange(low.shape[0])
    return {"low_res_multimasks": low, "high_res_multimasks": high, "ious": iou, "object_score_logits": obj,
            "best": best, "low_res_masks": low[bi, best].unsqueeze(1), "high_res_masks": high[bi, best].unsqueeze(1)}


def mask_iou(a: torch.Tensor, b: torch.Tensor) -> float:
    """IoU of two boolean masks; IoU(empty, empty) = 1."""
    a, b = a.bool(), b.bool()
    u = (a | b).sum().item()
    return 1.0 if u == 0 else (a & b).sum().item() / u
