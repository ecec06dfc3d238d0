"""Drop-in mirror of the reference's SAM prompt encoder + mask decoder (SAM2/sam2/sam2/modeling/sam/{prompt_encoder,
mask_decoder,transformer}.py), of `Attention_LoRA` / `wrap_decoder_lora` (lora_qkv.py:11-67,112-143) and of the part of
`SAM2Base` that turns image-encoder outputs into masks on a prompted frame (`forward_image`, `_forward_sam_heads`,
sam2_base.py:257-413,467-479) -- SURVEY.md section 8f row 4, forward / inference only.

Same class names, constructor arguments, attribute names and state_dict keys as the reference for everything the
reference's factory (sam2_model.py:62-88) uses on this path, so a Meta checkpoint's `sam_prompt_encoder.*`,
`sam_mask_decoder.*` and `no_mem_embed` entries load.  The modules own no arithmetic: every Linear, 1x1 convolution,
ConvTranspose2d(2, 2) and hyper-network product is an `lsam_linear` call (bias / ReLU / sigmoid / residual in the
epilogue), every LayerNorm / LayerNorm2d an `lsam_layernorm_fwd` or the fused `lsam_upscale2x2_fwd`, attention is
`lsam_mha_fwd`; activations are token-major NHWC rows, the token and image streams fp32 with T = fp32 / bf16 operands.

The BACKWARD of the heads is built too (heads_autograd.py: every forward call is an autograd node whose backward is
liblsam work): gradients reach the image embedding, the decoder-side LoRA factors and the fresh projections
`Attention_LoRA` creates (everything the reference leaves trainable after `wrap_decoder_lora`); the mask losses are in
loss_fns.py.

NOT built (out of scope): mask prompts (`mask_downscaling`), box prompts, the memory
encoder / memory attention / object pointers.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from . import heads_autograd as HA
from ._C import LsamError
from .hieradet import MLP, _compute_dtype
from .image_encoder import forward_image as _forward_image

NO_OBJ_SCORE = -1024.0   # sam2_base.py:19


# ---------------------------------------------------------------------------------------------
# parameter containers with the reference's names
# ---------------------------------------------------------------------------------------------
class LayerNorm2d(nn.Module):
    """sam2_utils.py:139-153 (channel LayerNorm of an NCHW map, eps 1e-6): a row LayerNorm in NHWC."""

    def __init__(self, num_channels: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps


def _unit_linear(x2, lin: nn.Linear, dt, res=None, out_f32=False):
    """x2 T [M, K] -> [M, N] through `lin` as a stand-alone `lsam_linear` (weights packed on the spot, N / K padded to 8)."""
    n, k = lin.weight.shape
    n8, k8 = _pad8(n), _pad8(k)
    wp = torch.zeros(n8, k8, dtype=torch.float32, device=x2.device)
    wp[:n, :k] = lin.weight.detach().float()
    bp = None
    if lin.bias is not None:
        bp = torch.zeros(n8, dtype=torch.float32, device=x2.device)
        bp[:n] = lin.bias.detach().float()
    if x2.shape[1] != k8:
        xp = torch.zeros(x2.shape[0], k8, dtype=x2.dtype, device=x2.device)
        xp[:, :min(k, x2.shape[1])] = x2[:, :k]
        x2 = xp
    out = ops.linear(x2.contiguous(), ops.pack_matrix(wp, dt), bp, res=res, out_f32=out_f32)
    return out, n


@torch.no_grad()
def _unit_attention(att, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """sam/transformer.py:233-265 (and lora_qkv.py:49-67 when `att` carries LoRA factors) as unit operators, inference
    only: q [B, nq, D], k / v [B, nk, Dkv] -> fp32 [B, nq, D].  `SAM2Base._forward_sam_heads` does not come through here
    (it keeps packed weights and autograd nodes); this is the drop-in for callers that invoke the module directly."""
    if not q.is_cuda:
        raise LsamError("Attention.forward runs on the MI355X only (no CPU or PyTorch fallback)")
    dt = _compute_dtype(None)
    B, nq, _ = q.shape
    T = lambda t: (t.to(dt) if t.dtype != dt else t).reshape(-1, t.shape[-1]).contiguous()
    q2, k2, v2 = T(q), T(k), T(v)
    tq = tv = None
    if hasattr(att, "A_q"):
        uq, _ = _unit_linear(q2, att.A_q, dt)
        tq, _ = _unit_linear(uq, att.B_q, dt, out_f32=True)
        uv, _ = _unit_linear(v2, att.A_v, dt)
        tv, _ = _unit_linear(uv, att.B_v, dt, out_f32=True)
    c = att.internal_dim
    qh, _ = _unit_linear(q2, att.q_proj, dt, res=tq)
    kh, _ = _unit_linear(k2, att.k_proj, dt)
    vh, _ = _unit_linear(v2, att.v_proj, dt, res=tv)
    o = ops.mha(qh.view(B, -1, c), kh.view(B, -1, c), vh.view(B, -1, c), att.num_heads)
    out, n = _unit_linear(o.view(-1, c), att.out_proj, dt, out_f32=True)
    return out[:, :n].reshape(B, nq, n)


class _HeadsContainer(nn.Module):
    """Parameter container of the SAM heads with the reference's names.  Its arithmetic is composed in
    `SAM2Base._forward_sam_heads` / `forward_first_frame` (packed weights, autograd nodes); calling the module itself
    says so instead of failing inside torch."""

    def forward(self, *args, **kwargs):
        raise LsamError(f"{type(self).__name__} is a parameter container here: run it through "
                        "image_segmentation_amd.sam_heads.SAM2Base._forward_sam_heads (or forward_first_frame)")


class Attention(nn.Module):
    """sam/transformer.py:199-265."""
    forward = _unit_attention

    def __init__(self, embedding_dim: int, num_heads: int, downsample_rate: int = 1, dropout: float = 0.0,
                 kv_in_dim: int = None):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.kv_in_dim = kv_in_dim if kv_in_dim is not None else embedding_dim
        self.internal_dim = embedding_dim // downsample_rate
        self.num_heads = num_heads
        assert self.internal_dim % num_heads == 0, "num_heads must divide embedding_dim."
        self.q_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.k_proj = nn.Linear(self.kv_in_dim, self.internal_dim)
        self.v_proj = nn.Linear(self.kv_in_dim, self.internal_dim)
        self.out_proj = nn.Linear(self.internal_dim, embedding_dim)
        self.dropout_p = dropout


class Attention_LoRA(nn.Module):
    """lora_qkv.py:11-67: q = q_proj(q) + B_q(A_q(q)), v = v_proj(v) + B_v(A_v(v)).

    As in the reference, the constructor creates FRESH q / k / v / out projections (lora_qkv.py:26-29): the wrapped
    attention's weights are discarded, and since `wrap_decoder_lora` runs after the freeze loop the new projections are
    trainable (train_lora_sam.py:116-133 hands them to the optimizer with the factors).  Their weight / bias gradients
    are computed by `heads_autograd.LinearFn`; tests/golden/decoder_lora_r4.npz pins logits and all 24 gradients to the
    reference's own run.  `fresh_projections=False` is this package's extension: keep (and keep frozen) the wrapped
    module's trained projections -- what LoRA means, and what `LoRA_qkv` does on the encoder side."""

    def __init__(self, self_attn: Attention, A_q, B_q, A_v, B_v, fresh_projections: bool = True):
        super().__init__()
        self.embedding_dim = self_attn.embedding_dim
        self.kv_in_dim = self_attn.kv_in_dim
        self.internal_dim = self_attn.internal_dim
        self.num_heads = self_attn.num_heads
        if fresh_projections:
            self.q_proj = nn.Linear(self.embedding_dim, self.internal_dim)
            self.k_proj = nn.Linear(self.kv_in_dim, self.internal_dim)
            self.v_proj = nn.Linear(self.kv_in_dim, self.internal_dim)
            self.out_proj = nn.Linear(self.internal_dim, self.embedding_dim)
            self.to(self_attn.q_proj.weight.device)
        else:
            self.q_proj, self.k_proj, self.v_proj, self.out_proj = (self_attn.q_proj, self_attn.k_proj, self_attn.v_proj,
                                                                    self_attn.out_proj)
        self.A_q, self.B_q, self.A_v, self.B_v = A_q, B_q, A_v, B_v
        self.dropout_p = self_attn.dropout_p

    forward = _unit_attention


class TwoWayAttentionBlock(_HeadsContainer):
    """sam/transformer.py:111-196."""

    def __init__(self, embedding_dim: int, num_heads: int, mlp_dim: int = 2048, activation=nn.ReLU,
                 attention_downsample_rate: int = 2, skip_first_layer_pe: bool = False):
        super().__init__()
        if activation is not nn.ReLU:
            raise NotImplementedError("the reference factory's two-way transformer uses ReLU (sam2_base.py:214-232)")
        self.self_attn = Attention(embedding_dim, num_heads)
        self.norm1 = nn.LayerNorm(embedding_dim)
        self.cross_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm2 = nn.LayerNorm(embedding_dim)
        self.mlp = MLP(embedding_dim, mlp_dim, embedding_dim, num_layers=2, activation=activation)
        self.norm3 = nn.LayerNorm(embedding_dim)
        self.norm4 = nn.LayerNorm(embedding_dim)
        self.cross_attn_image_to_token = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.skip_first_layer_pe = skip_first_layer_pe


class TwoWayTransformer(_HeadsContainer):
    """sam/transformer.py:17-108."""

    def __init__(self, depth: int, embedding_dim: int, num_heads: int, mlp_dim: int, activation=nn.ReLU,
                 attention_downsample_rate: int = 2):
        super().__init__()
        self.depth, self.embedding_dim, self.num_heads, self.mlp_dim = depth, embedding_dim, num_heads, mlp_dim
        self.layers = nn.ModuleList(
            TwoWayAttentionBlock(embedding_dim, num_heads, mlp_dim, activation, attention_downsample_rate,
                                 skip_first_layer_pe=(i == 0)) for i in range(depth))
        self.final_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm_final_attn = nn.LayerNorm(embedding_dim)


class PositionEmbeddingRandom(nn.Module):
    """position_encoding.py:133-176."""

    def __init__(self, num_pos_feats: int = 64, scale: Optional[float] = None):
        super().__init__()
        if scale is None or scale <= 0.0:
            scale = 1.0
        self.register_buffer("positional_encoding_gaussian_matrix", scale * torch.randn((2, num_pos_feats)))


class PromptEncoder(_HeadsContainer):
    """sam/prompt_encoder.py:17-202, point prompts and the no-mask dense embedding (mask_downscaling is not built)."""

    def __init__(self, embed_dim: int, image_embedding_size: Tuple[int, int], input_image_size: Tuple[int, int],
                 mask_in_chans: int = 16, activation=nn.GELU):
        super().__init__()
        self.embed_dim = embed_dim
        self.input_image_size = input_image_size
        self.image_embedding_size = image_embedding_size
        self.pe_layer = PositionEmbeddingRandom(embed_dim // 2)
        self.num_point_embeddings = 4
        self.point_embeddings = nn.ModuleList(nn.Embedding(1, embed_dim) for _ in range(self.num_point_embeddings))
        self.not_a_point_embed = nn.Embedding(1, embed_dim)
        self.no_mask_embed = nn.Embedding(1, embed_dim)


class MaskDecoder(_HeadsContainer):
    """sam/mask_decoder.py:15-110 (constructor); predict_masks (:161-250) is `SAM2Base._predict_masks` below."""

    def __init__(self, *, transformer_dim: int, transformer: nn.Module, num_multimask_outputs: int = 3, activation=nn.GELU,
                 iou_head_depth: int = 3, iou_head_hidden_dim: int = 256, use_high_res_features: bool = False,
                 iou_prediction_use_sigmoid=False, dynamic_multimask_via_stability=False,
                 dynamic_multimask_stability_delta=0.05, dynamic_multimask_stability_thresh=0.98,
                 pred_obj_scores: bool = False, pred_obj_scores_mlp: bool = False,
                 use_multimask_token_for_obj_ptr: bool = False):
        super().__init__()
        if activation is not nn.GELU:
            raise NotImplementedError("the reference factory's mask decoder uses GELU")
        self.transformer_dim = transformer_dim
        self.transformer = transformer
        self.num_multimask_outputs = num_multimask_outputs
        self.iou_token = nn.Embedding(1, transformer_dim)
        self.num_mask_tokens = num_multimask_outputs + 1
        self.mask_tokens = nn.Embedding(self.num_mask_tokens, transformer_dim)
        self.pred_obj_scores = pred_obj_scores
        if pred_obj_scores:
            self.obj_score_token = nn.Embedding(1, transformer_dim)
        self.output_upscaling = nn.Sequential(
            nn.ConvTranspose2d(transformer_dim, transformer_dim // 4, kernel_size=2, stride=2),
            LayerNorm2d(transformer_dim // 4), activation(),
            nn.ConvTranspose2d(transformer_dim // 4, transformer_dim // 8, kernel_size=2, stride=2), activation())
        self.use_high_res_features = use_high_res_features
        if use_high_res_features:
            self.conv_s0 = nn.Conv2d(transformer_dim, transformer_dim // 8, kernel_size=1, stride=1)
            self.conv_s1 = nn.Conv2d(transformer_dim, transformer_dim // 4, kernel_size=1, stride=1)
        self.output_hypernetworks_mlps = nn.ModuleList(
            MLP(transformer_dim, transformer_dim, transformer_dim // 8, 3) for _ in range(self.num_mask_tokens))
        self.iou_prediction_head = MLP(transformer_dim, iou_head_hidden_dim, self.num_mask_tokens, iou_head_depth,
                                       sigmoid_output=iou_prediction_use_sigmoid)
        if pred_obj_scores:
            self.pred_obj_score_head = nn.Linear(transformer_dim, 1)
            if pred_obj_scores_mlp:
                self.pred_obj_score_head = MLP(transformer_dim, transformer_dim, 1, 3)
        if dynamic_multimask_via_stability:
            raise NotImplementedError("dynamic_multimask_via_stability is an inference-demo option, off in the reference factory")


def wrap_decoder_lora(sam_model, rank: int, fresh_projections: bool = True):
    """lora_qkv.py:112-143: freeze the mask decoder, wrap every transformer layer's self-attention with Attention_LoRA
    (A ~ kaiming_uniform(a=sqrt(5)), B = 0).  With the default (the reference's behaviour) each wrapped layer gets fresh,
    trainable q / k / v / out projections; `fresh_projections=False` keeps the wrapped ones, frozen."""
    assert rank > 0
    for param in sam_model.sam_mask_decoder.parameters():
        param.requires_grad = False
    for blk in sam_model.sam_mask_decoder.transformer.layers:
        sa = blk.self_attn
        dev = sa.q_proj.weight.device
        A_q = nn.Linear(sa.embedding_dim, rank, bias=False).to(dev)
        B_q = nn.Linear(rank, sa.internal_dim, bias=False).to(dev)
        A_v = nn.Linear(sa.kv_in_dim, rank, bias=False).to(dev)
        B_v = nn.Linear(rank, sa.embedding_dim, bias=False).to(dev)
        nn.init.kaiming_uniform_(A_q.weight, a=math.sqrt(5))
        nn.init.kaiming_uniform_(A_v.weight, a=math.sqrt(5))
        nn.init.zeros_(B_q.weight)
        nn.init.zeros_(B_v.weight)
        blk.self_attn = Attention_LoRA(sa, A_q, B_q, A_v, B_v, fresh_projections=fresh_projections)


# ---------------------------------------------------------------------------------------------
# packed operands (built once, rebuilt when a parameter changes)
# ---------------------------------------------------------------------------------------------
def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


class _PackCache:
    def __init__(self):
        self.d = {}

    def get(self, key, params, dt, build):
        sig = tuple((p.data_ptr(), p._version) for p in params)
        e = self.d.get((key, dt))
        if e is None or e[0] != sig:
            e = (sig, build())
            self.d[(key, dt)] = e
        return e[1]

    def linear(self, mod: nn.Module, dt, weight2d=None, bias=None, name=None):
        """(packed T weight [pad8(N), pad8(K)], fp32 bias [pad8(N)] or None, its packed transpose [pad8(K), pad8(N)] for the
        input gradient, N) of a Linear-like module."""
        params = [p for p in (mod.weight, getattr(mod, "bias", None)) if p is not None]

        def build():
            w = mod.weight if weight2d is None else weight2d()
            b = getattr(mod, "bias", None) if bias is None else bias()
            w2 = w.detach().float().reshape(w.shape[0], -1)
            n, k = w2.shape
            wp = torch.zeros(_pad8(n), _pad8(k), dtype=torch.float32, device=w2.device)
            wp[:n, :k] = w2
            bp = None
            if b is not None:
                bp = torch.zeros(_pad8(n), dtype=torch.float32, device=w2.device)
                bp[:n] = b.detach().float().reshape(-1)
            return ops.pack_matrix(wp, dt), bp, ops.pack_matrix(wp.t(), dt), n

        return self.get((id(mod), name), params, dt, build)


# ---------------------------------------------------------------------------------------------
class SAM2Base(nn.Module):
    """The prompted-frame part of sam2_base.py's SAM2Base with the reference factory's options (sam2_model.py:62-88):
    use_high_res_features_in_sam, pred_obj_scores (+ _mlp), iou_prediction_use_sigmoid, multimask output on the first
    click.  `image_encoder` may be None when only the heads are used."""

    def __init__(self, image_encoder: Optional[nn.Module] = None, image_size: int = 1024, backbone_stride: int = 16,
                 hidden_dim: int = 256, use_high_res_features_in_sam: bool = True, pred_obj_scores: bool = True,
                 pred_obj_scores_mlp: bool = True, iou_prediction_use_sigmoid: bool = True):
        super().__init__()
        self.image_encoder = image_encoder
        self.image_size = image_size
        self.backbone_stride = backbone_stride
        self.hidden_dim = hidden_dim
        self.use_high_res_features_in_sam = use_high_res_features_in_sam
        self.pred_obj_scores = pred_obj_scores
        self.sam_image_embedding_size = image_size // backbone_stride
        self.no_mem_embed = nn.Parameter(torch.zeros(1, 1, hidden_dim))
        e = self.sam_image_embedding_size
        self.sam_prompt_encoder = PromptEncoder(embed_dim=hidden_dim, image_embedding_size=(e, e),
                                                input_image_size=(image_size, image_size), mask_in_chans=16)
        self.sam_mask_decoder = MaskDecoder(
            num_multimask_outputs=3,
            transformer=TwoWayTransformer(depth=2, embedding_dim=hidden_dim, mlp_dim=2048, num_heads=8),
            transformer_dim=hidden_dim, iou_head_depth=3, iou_head_hidden_dim=256,
            use_high_res_features=use_high_res_features_in_sam, iou_prediction_use_sigmoid=iou_prediction_use_sigmoid,
            pred_obj_scores=pred_obj_scores, pred_obj_scores_mlp=pred_obj_scores_mlp)
        self.compute_dtype: Optional[torch.dtype] = torch.float32   # the heads are ~1 % of the FLOPs: exact-fp32 MFMA by default
        self._packs = _PackCache()
        self._dense_pe = {}
        # packed weights are keyed by (data_ptr, _version); checkpoint loaders that write through `.data` do not bump the
        # version, so a load drops every packed image (as Hiera.invalidate_packed does for the trunk)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    def invalidate_packed(self):
        """Drop the packed head weights and the cached dense positional encoding (call after editing parameters through
        `.data`; `load_state_dict` does it by itself).  The attached image encoder has its own `invalidate_packed`."""
        self._packs = _PackCache()
        self._dense_pe = {}
        for m in (getattr(self.image_encoder, "trunk", None), getattr(self.image_encoder, "neck", None)):
            if m is not None and hasattr(m, "invalidate_packed"):
                m.invalidate_packed()

    # -- sam2_base.py:467-479 ------------------------------------------------------------------------------------
    def forward_image(self, img_batch: torch.Tensor):
        if self.image_encoder is None:
            raise LsamError("SAM2Base.forward_image: no image encoder attached")
        return _forward_image(self, img_batch)

    # -- helpers -----------------------------------------------------------------------------------------------------
    def _dt(self) -> torch.dtype:
        return self.compute_dtype if self.compute_dtype is not None else _compute_dtype(None)

    def _lin(self, x, mod, dt, weight2d=None, bias=None, name=None, keep_pad=False, **kw):
        """x T [M, K] -> [M, N] through mod's packed weight (N, K zero-padded to multiples of 8; `keep_pad` returns the
        padded columns, all zero, for a following product over them).  `few_rows`: x is the token stream (a handful of rows
        per frame), see LSAM_EPI_FEW_ROWS; a property of the call site, never of the batch size."""
        pack = self._packs.linear(mod, dt, weight2d, bias, name)
        wp = mod.weight if (weight2d is None and mod.weight.requires_grad) else None      # a LoRA factor or a fresh projection
        mb = getattr(mod, "bias", None)
        bp = mb if (bias is None and mb is not None and mb.requires_grad) else None
        out = HA.LinearFn.apply(x, kw.get("res"), wp, pack, kw.get("relu", False), kw.get("sigmoid", False), kw.get("out_f32", False),
                                kw.get("few_rows", False), bp)
        n = pack[3]
        return out if keep_pad or out.shape[1] == n else out[:, :n]

    def _T(self, x, dt):
        return x if x.dtype == dt else HA.AddRowsFn.apply(x, None, dt)

    def _add(self, a, b, dt):
        return HA.AddRowsFn.apply(a, b, dt)

    def _mlp(self, mlp: MLP, x, dt):
        """sam2_utils.MLP: Linear + ReLU ... Linear (+ sigmoid); x T [M, in] -> fp32 [M, out]."""
        n = len(mlp.layers)
        for i, layer in enumerate(mlp.layers):
            last = i == n - 1
            x = self._lin(x, layer, dt, relu=not last, sigmoid=last and mlp.sigmoid_output, out_f32=last, few_rows=True)
            if not last and x.stride(0) != x.shape[1]:
                x = x.contiguous()
        return x

    def _attention(self, att, q, k, v, B: int, dt, residual=None, q_tokens: bool = True, k_tokens: bool = True):
        """sam/transformer.py:233-265 (+ lora_qkv.py:50-53 for Attention_LoRA): q T [B*nq, D], k / v T [B*nk, D] ->
        fp32 [B*nq, D] = (residual +) out_proj(softmax(q k^T / sqrt(hd)) v).  `q_tokens` / `k_tokens`: that side is the token
        stream (LSAM_EPI_FEW_ROWS GEMMs) rather than the 4096-row image stream -- a property of the CALL SITE, so the kernel
        choice never depends on how many prompt points there are."""
        lora = isinstance(att, Attention_LoRA)
        tq = tv = None
        fq, fk = q_tokens, k_tokens
        if lora:   # the rank-r updates as fp32 residuals of the projections
            tq = self._lin(self._lin(q, att.A_q, dt, keep_pad=True, few_rows=fq), att.B_q, dt, out_f32=True, few_rows=fq)
            tv = self._lin(self._lin(v, att.A_v, dt, keep_pad=True, few_rows=fk), att.B_v, dt, out_f32=True, few_rows=fk)
        qh = self._lin(q, att.q_proj, dt, res=tq, few_rows=fq)
        kh = self._lin(k, att.k_proj, dt, few_rows=fk)
        vh = self._lin(v, att.v_proj, dt, res=tv, few_rows=fk)
        c = att.internal_dim
        o = HA.MhaFn.apply(qh.reshape(B, -1, c), kh.reshape(B, -1, c), vh.reshape(B, -1, c), att.num_heads)
        return self._lin(o.view(-1, c), att.out_proj, dt, res=residual, out_f32=True, few_rows=fq)

    def _ln(self, norm: nn.LayerNorm, x):
        return HA.LayerNormFn.apply(x, norm.weight.detach(), norm.bias.detach(), norm.eps)

    def get_dense_pe(self, device) -> torch.Tensor:
        """PromptEncoder.get_dense_pe (prompt_encoder.py:63-72) as fp32 rows [E*E, D]; constant per model, cached."""
        pe = self.sam_prompt_encoder
        g = pe.pe_layer.positional_encoding_gaussian_matrix
        key = (str(device), g.data_ptr(), g._version)
        if key not in self._dense_pe:
            e = pe.image_embedding_size[0]
            ar = (torch.arange(e, device=device, dtype=torch.float32) + 0.5) / e
            coords = torch.stack([ar.repeat(e), ar.repeat_interleave(e)], dim=1)        # (x, y) per row-major position
            self._dense_pe = {key: ops.pe_encode(g.to(device), coords)}
        return self._dense_pe[key]

    def _embed_points(self, coords: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """PromptEncoder._embed_points with pad=True (prompt_encoder.py:75-111): fp32 [B, P+1, D]."""
        pe = self.sam_prompt_encoder
        B, P = labels.shape
        dev = coords.device
        pts = torch.cat([coords.float() + 0.5, torch.zeros(B, 1, 2, device=dev)], dim=1)          # index plumbing
        lab = torch.cat([labels, -torch.ones(B, 1, dtype=labels.dtype, device=dev)], dim=1)
        c01 = (pts / float(pe.input_image_size[0])).reshape(-1, 2).contiguous()
        enc = ops.pe_encode(pe.pe_layer.positional_encoding_gaussian_matrix, c01)                   # [B*(P+1), D]
        # per-label embedding table: row 0 = not_a_point (replaces the encoding), rows 1..4 = point_embeddings (added)
        table = torch.cat([pe.not_a_point_embed.weight] + [m.weight for m in pe.point_embeddings], 0).detach().float()
        idx = (lab.reshape(-1) + 1).long()
        enc[idx == 0] = 0.0                          # a padding point's encoding is replaced, not added to
        return ops.add_rows(enc, table[idx].contiguous(), torch.float32).view(B, P + 1, -1)

    # -- sam/transformer.py:60-108 + mask_decoder.py:161-250 --------------------------------------------------------------
    def _predict_masks(self, pix: torch.Tensor, sparse: torch.Tensor, feat_s0: torch.Tensor, feat_s1: torch.Tensor):
        """pix: image embedding rows [B*E*E, D] (fp32 or T), sparse fp32 [B, P+1, D], feat_s0 / feat_s1 NHWC
        [B, 4E, 4E, D/8] / [B, 2E, 2E, D/4].  Returns (mask logits fp32 [B, (4E)^2, 8], iou fp32 [B, 4], obj fp32 [B, 1])."""
        dt = self._dt()
        dec, tr = self.sam_mask_decoder, self.sam_mask_decoder.transformer
        D, E = self.hidden_dim, self.sam_image_embedding_size
        B = sparse.shape[0]
        dev = sparse.device
        # mask_decoder.py:178-196: the object-score token exists only with pred_obj_scores, and shifts the others by one
        s_ = 1 if dec.pred_obj_scores else 0
        out_tokens = torch.cat(([dec.obj_score_token.weight] if s_ else []) + [dec.iou_token.weight, dec.mask_tokens.weight], 0).detach().float()
        tokens = torch.cat([out_tokens.unsqueeze(0).expand(B, -1, -1), sparse], dim=1).contiguous()   # concatenation only
        Nt = tokens.shape[1]
        point_embedding = tokens.view(B * Nt, D)
        queries = point_embedding
        key_pe = self.get_dense_pe(dev)                                                                 # [E*E, D]
        keys = self._add(pix, self.sam_prompt_encoder.no_mask_embed.weight.detach().float(), torch.float32)   # src = emb + dense

        for l, blk in enumerate(tr.layers):
            if blk.skip_first_layer_pe:
                qT = self._T(queries, dt)
                queries = self._attention(blk.self_attn, qT, qT, qT, B, dt)
            else:
                q = self._add(queries, point_embedding, dt)
                queries = self._attention(blk.self_attn, q, q, self._T(queries, dt), B, dt, residual=queries)
            queries = self._ln(blk.norm1, queries)
            q = self._add(queries, point_embedding, dt)
            k = self._add(keys, key_pe, dt)
            queries = self._attention(blk.cross_attn_token_to_image, q, k, self._T(keys, dt), B, dt, residual=queries, k_tokens=False)
            queries = self._ln(blk.norm2, queries)
            h = self._lin(self._T(queries, dt), blk.mlp.layers[0], dt, relu=True, few_rows=True)
            queries = self._lin(h, blk.mlp.layers[1], dt, res=queries, out_f32=True, few_rows=True)
            queries = self._ln(blk.norm3, queries)
            q = self._add(queries, point_embedding, dt)
            keys = self._attention(blk.cross_attn_image_to_token, k, q, self._T(queries, dt), B, dt, residual=keys, q_tokens=False)
            keys = self._ln(blk.norm4, keys)
        q = self._add(queries, point_embedding, dt)
        k = self._add(keys, key_pe, dt)
        queries = self._attention(tr.final_attn_token_to_image, q, k, self._T(keys, dt), B, dt, residual=queries, k_tokens=False)
        hs = self._ln(tr.norm_final_attn, queries).view(B, Nt, D)

        # ---- upscaling: two ConvTranspose2d(2, 2) as GEMMs over the (dy, dx, co) columns -----------------------------
        up0, ln1, up3 = dec.output_upscaling[0], dec.output_upscaling[1], dec.output_upscaling[3]
        c1, c2 = up0.out_channels, up3.out_channels
        g1 = self._lin(self._T(keys, dt), up0, dt, weight2d=lambda: up0.weight.permute(2, 3, 1, 0).reshape(4 * c1, -1),
                       bias=lambda: up0.bias.repeat(4), name="ct", out_f32=True)
        skip1 = (feat_s1.to(dt) if feat_s1.dtype != dt else feat_s1).detach().contiguous()   # LoRA-free stages: no gradient path
        skip0 = (feat_s0.to(dt) if feat_s0.dtype != dt else feat_s0).detach().contiguous()
        u1 = HA.Upscale2x2Fn.apply(g1.contiguous(), skip1, ln1.weight.detach(), ln1.bias.detach(), ln1.eps, B, E, E, c1, dt)
        g2 = self._lin(u1.view(-1, c1), up3, dt, weight2d=lambda: up3.weight.permute(2, 3, 1, 0).reshape(4 * c2, -1),
                       bias=lambda: up3.bias.repeat(4), name="ct", out_f32=True)
        u2 = HA.Upscale2x2Fn.apply(g2.contiguous(), skip0, None, None, 0.0, B, 2 * E, 2 * E, c2, dt)
        # ---- hyper-networks, IoU and object-score heads ----------------------------------------------------------------------
        nm = dec.num_mask_tokens
        hsT = self._T(hs.reshape(B * Nt, D), dt).view(B, Nt, D)
        hyper = torch.stack([self._mlp(dec.output_hypernetworks_mlps[i], hsT[:, s_ + 1 + i].contiguous(), dt) for i in range(nm)]
                            + [torch.zeros(B, c2, dtype=torch.float32, device=dev)] * (8 - nm), dim=1)   # placement only: [B, 8, c2]
        hyperT = self._T(hyper.reshape(B * 8, c2), dt).view(B, 8, c2)
        HW = 16 * E * E
        logits = HA.MaskProductFn.apply(u2.view(B, HW, c2), hyperT)   # masks[b] = hyper_in[b] @ upscaled[b]  (mask_decoder.py:238-239)
        iou = self._mlp(dec.iou_prediction_head, hsT[:, s_].contiguous(), dt)
        if self.pred_obj_scores:
            head = dec.pred_obj_score_head
            obj = self._mlp(head, hsT[:, 0].contiguous(), dt) if isinstance(head, MLP) else \
                self._lin(hsT[:, 0].contiguous(), head, dt, out_f32=True, few_rows=True)
        else:
            obj = torch.full((B, 1), 10.0, dtype=torch.float32, device=dev)
        return logits, iou.contiguous(), obj.contiguous()

    # -- sam2_base.py:257-413 ----------------------------------------------------------------------------------------------
    def _forward_sam_heads(self, backbone_features: torch.Tensor, point_inputs=None, mask_inputs=None,
                           high_res_features: Optional[List[torch.Tensor]] = None, multimask_output: bool = False):
        """Same arguments and return tuple as the reference (obj_ptr is None: object pointers belong to the memory path).
        backbone_features [B, D, E, E]; point_inputs {"point_coords": [B, P, 2] pixels (x, y), "point_labels": [B, P]};
        high_res_features [feat_s0 [B, D/8, 4E, 4E], feat_s1 [B, D/4, 2E, 2E]] (forward_image's levels 0 and 1)."""
        if mask_inputs is not None:
            raise NotImplementedError("mask prompts (mask_downscaling) are not built")
        if not backbone_features.is_cuda:
            raise LsamError("the SAM heads run on the MI355X only (no CPU or PyTorch fallback)")
        if high_res_features is None or not self.use_high_res_features_in_sam:
            raise NotImplementedError("the reference factory always passes high-resolution features")
        B = backbone_features.shape[0]
        D, E = self.hidden_dim, self.sam_image_embedding_size
        assert tuple(backbone_features.shape[1:]) == (D, E, E)
        dev = backbone_features.device
        if point_inputs is not None:
            coords, labels = point_inputs["point_coords"].to(dev), point_inputs["point_labels"].to(dev)
        else:   # no prompt: one padding point with label -1 (sam2_base.py:301-304)
            coords = torch.zeros(B, 1, 2, device=dev)
            labels = -torch.ones(B, 1, dtype=torch.int32, device=dev)
        sparse = self._embed_points(coords, labels)
        nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()      # no copy for the encoder's channels_last outputs
        pix = nhwc(backbone_features).view(B * E * E, D)
        logits, iou, obj = self._predict_masks(pix, sparse, nhwc(high_res_features[0]), nhwc(high_res_features[1]))
        first, nm = (1, 3) if multimask_output else (0, 1)
        low = HA.MaskGateFn.apply(logits, obj.detach() if self.pred_obj_scores else None, first, nm, NO_OBJ_SCORE).view(B, nm, 4 * E, 4 * E)
        high = HA.BilinearFn.apply(low, self.image_size // (4 * E))
        ious = iou[:, first:first + nm]
        if multimask_output:
            best = torch.argmax(ious, dim=-1)                    # index selection
            bi = torch.arange(B, device=dev)
            low_best, high_best = low[bi, best].unsqueeze(1), high[bi, best].unsqueeze(1)
        else:
            low_best, high_best = low, high
        return low, high, ious, low_best, high_best, None, obj

    def forward_first_frame(self, backbone_out: dict, point_coords: torch.Tensor, point_labels: torch.Tensor) -> dict:
        """A click on an initial frame: `_prepare_memory_conditioned_features` with no memory (`+ no_mem_embed`,
        sam2_base.py:669-674) and `_forward_sam_heads(multimask_output=True)` (the first click, sam2_base.py:797-812)."""
        f0, f1, f2 = backbone_out["backbone_fpn"]
        B, D, E, _ = f2.shape
        rows = f2.permute(0, 2, 3, 1).contiguous().view(B * E * E, D)
        pix = self._add(rows, self.no_mem_embed.detach().float().view(1, D), torch.float32)
        pix = pix.view(B, E, E, D).permute(0, 3, 1, 2)
        low, high, ious, low_best, high_best, _, obj = self._forward_sam_heads(
            pix, {"point_coords": point_coords, "point_labels": point_labels}, None, [f0, f1], multimask_output=True)
        return {"low_res_multimasks": low, "high_res_multimasks": high, "ious": ious, "object_score_logits": obj,
                "best": torch.argmax(ious, dim=-1), "low_res_masks": low_best, "high_res_masks": high_best}
