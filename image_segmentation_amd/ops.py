"""Tensor-level wrappers over the C ABI (one call = one HIP kernel launch or two).

These exist for the parity tests and for the stand-alone drop-in modules
(`LoRA_qkv`, `MultiScaleBlock` used outside a `Hiera`).  The trunk hot path does not
go through here: `Hiera.forward` makes a single `lsam_hiera_forward` call.
Everything requires CUDA(HIP) tensors; nothing here computes in PyTorch."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _C
from ._C import check, dtype_code, lib, ptr, stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _C.LsamError("liblsam operators need tensors on the MI355X (cuda) device; "
                               "there is no CPU fallback")


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """A vector / matrix argument the kernels read as contiguous fp32 (biases, LayerNorm parameters and statistics,
    residuals): module Parameters after `model.half()` / `.to(bfloat16)`, or non-contiguous views, are converted
    instead of being read as garbage.  No copy when the tensor already qualifies."""
    if t is None:
        return None
    t = t.detach()
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    return t


def _f32rows(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """fp32 2-D tensor with contiguous rows (a leading dimension is passed separately)."""
    if t is None:
        return None
    t = t.detach()
    if t.dtype != torch.float32 or t.stride(-1) != 1:
        t = t.to(torch.float32).contiguous()
    return t


def _rows2d(t: torch.Tensor) -> Tuple[int, int, int]:
    """(rows, cols, ld) of a 2-D tensor whose rows are contiguous."""
    assert t.dim() == 2 and t.stride(1) == 1, "expect row-major 2-D tensor"
    return t.shape[0], t.shape[1], t.stride(0)


def pack_matrix(src: torch.Tensor, dtype: torch.dtype, ld: Optional[int] = None) -> torch.Tensor:
    """fp32 2-D (any strides, e.g. a transposed view) -> packed T [rows, ld], zero padded."""
    _need_gpu(src)
    assert src.dim() == 2 and src.dtype == torch.float32
    rows, cols = src.shape
    ld = cols if ld is None else ld
    dst = torch.empty(rows, ld, dtype=dtype, device=src.device)
    check(lib().lsam_pack_matrix(dtype_code(dtype), ptr(src), src.stride(0), src.stride(1), rows, cols,
                                 ptr(dst), ld, stream()))
    return dst


def linear(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None,
           res: Optional[torch.Tensor] = None, res_row_mod: int = 0, gelu: bool = False,
           gelu_bwd_aux: Optional[torch.Tensor] = None, out_f32: bool = False, save_pre: bool = False,
           out: Optional[torch.Tensor] = None, K: Optional[int] = None, N: Optional[int] = None, relu: bool = False,
           sigmoid: bool = False, few_rows: bool = False):
    """out = epilogue(A[:, :K] @ W[:N, :K].T + bias) + res   (see lsam_linear)."""
    _need_gpu(A, W, bias, res)
    assert A.dtype == W.dtype
    bias, res = _f32c(bias), _f32rows(res)
    M, Ka, lda = _rows2d(A)
    Nw, Kw, ldw = _rows2d(W)
    K = min(Ka, Kw) if K is None else K
    N = Nw if N is None else N
    odt = torch.float32 if out_f32 else A.dtype
    if out is None:
        out = torch.empty(M, N, dtype=odt, device=A.device)
    assert out.dtype == odt
    _, _, ldo = _rows2d(out)
    out2 = torch.empty(M, N, dtype=A.dtype, device=A.device) if save_pre else None
    flags = (_C.EPI_GELU if gelu else 0) | (_C.EPI_GELU_BWD if gelu_bwd_aux is not None else 0) | \
        (_C.EPI_OUT_F32 if out_f32 else 0) | (_C.EPI_SAVE_PRE if save_pre else 0) | (_C.EPI_RELU if relu else 0) | \
        (_C.EPI_SIGMOID if sigmoid else 0) | (_C.EPI_FEW_ROWS if few_rows else 0)
    ldres = res.stride(0) if res is not None else 0
    ldaux = gelu_bwd_aux.stride(0) if gelu_bwd_aux is not None else 0
    check(lib().lsam_linear(dtype_code(A.dtype), ptr(A), lda, ptr(W), ldw, ptr(bias), ptr(res), ldres,
                            res_row_mod, ptr(gelu_bwd_aux), ldaux, ptr(out), ldo, ptr(out2),
                            N if save_pre else 0, flags, M, N, K, stream()))
    return (out, out2) if save_pre else out


def linear_maxpool2x2(A: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, n_img: int, H: int, Wd: int):
    """fp32 [n_img, H/2, Wd/2, N] = MaxPool2d(2,2)(A @ W.T + bias) over the token grid (see lsam_linear_maxpool2x2)."""
    _need_gpu(A, W, bias)
    assert A.dtype == torch.bfloat16 and W.dtype == torch.bfloat16
    M, K, lda = _rows2d(A)
    N, Kw, ldw = _rows2d(W)
    assert M == n_img * H * Wd and K == Kw
    b = bias.detach().float().contiguous()
    out = torch.empty(n_img, H // 2, Wd // 2, N, dtype=torch.float32, device=A.device)
    check(lib().lsam_linear_maxpool2x2(ptr(A), lda, ptr(W), ldw, ptr(b), ptr(out), N, n_img, H, Wd, N, K, stream()))
    return out


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, dtype: torch.dtype,
                  ld: Optional[int] = None, want_stats: bool = False, eps: float = 1e-6):
    _need_gpu(x, gamma, beta)
    x, gamma, beta = _f32rows(x), _f32c(gamma), _f32c(beta)
    M, Cc, ldx = _rows2d(x)
    ld = Cc if ld is None else ld
    y = torch.zeros(M, ld, dtype=dtype, device=x.device) if ld != Cc else \
        torch.empty(M, ld, dtype=dtype, device=x.device)
    stats = torch.empty(M, 2, dtype=torch.float32, device=x.device) if want_stats else None
    check(lib().lsam_layernorm_fwd(dtype_code(dtype), ptr(x), ldx, ptr(gamma), ptr(beta), ptr(y), ld,
                                   ptr(stats), M, Cc, eps, stream()))
    return (y, stats) if want_stats else y


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor,
                  dres: Optional[torch.Tensor], dtype: torch.dtype):
    _need_gpu(dy, x, stats, gamma, dres)
    dy, x, stats, gamma, dres = _f32rows(dy), _f32rows(x), _f32c(stats), _f32c(gamma), _f32rows(dres)
    M, Cc, lddy = _rows2d(dy)
    dx = torch.empty(M, Cc, dtype=torch.float32, device=x.device)
    dxt = torch.empty(M, Cc, dtype=dtype, device=x.device)
    check(lib().lsam_layernorm_bwd(dtype_code(dtype), ptr(dy), lddy, ptr(x), x.stride(0), ptr(stats),
                                   ptr(gamma), ptr(dres), dres.stride(0) if dres is not None else 0,
                                   ptr(dx), Cc, ptr(dxt), Cc, M, Cc, stream()))
    return dx, dxt


def attention_fwd(qkv: torch.Tensor, bias: torch.Tensor, ws: int, pool: int, nh: int,
                  want_lse: bool = False, q_pooled: Optional[torch.Tensor] = None):
    """qkv T [N,H,W,3*nh*hd] (contiguous) -> out T [N,H/pool,W/pool,nh*hd] (+ lse fp32 [N,H',W',nh]).
    q_pooled (pool == 2 only): fp32 [N,H/2,W/2,nh*hd], the pooled queries; qkv's q columns are then not read."""
    _need_gpu(qkv, bias)
    bias = _f32c(bias)
    assert qkv.dim() == 4 and qkv.is_contiguous()
    N, H, W, C3 = qkv.shape
    dout = C3 // 3
    hd = dout // nh
    Ho, Wo = H // pool, W // pool
    out = torch.empty(N, Ho, Wo, dout, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(N, Ho, Wo, nh, dtype=torch.float32, device=qkv.device) if want_lse else None
    bias_t = bias.to(qkv.dtype) if qkv.dtype != torch.float32 else None   # dtype cast only (plumbing)
    if q_pooled is not None:
        _need_gpu(q_pooled)
        assert pool == 2 and q_pooled.dtype == torch.float32 and q_pooled.is_contiguous()
        assert tuple(q_pooled.shape) == (N, Ho, Wo, dout)
        check(lib().lsam_attention_fwd_pooled_q(dtype_code(qkv.dtype), ptr(qkv), C3, ptr(bias), ptr(bias_t), ptr(q_pooled),
                                                dout, ptr(out), dout, ptr(lse), N, H, W, ws, nh, hd, stream()))
        return (out, lse) if want_lse else out
    check(lib().lsam_attention_fwd(dtype_code(qkv.dtype), ptr(qkv), C3, ptr(bias), ptr(bias_t), ptr(out), dout,
                                   ptr(lse), N, H, W, ws, pool, nh, hd, stream()))
    return (out, lse) if want_lse else out


def attention_bwd(qkv: torch.Tensor, bias: torch.Tensor, dout_t: torch.Tensor, lse: torch.Tensor,
                  ws: int, nh: int, ld_out: Optional[int] = None, pool: int = 1) -> torch.Tensor:
    """dqkv [N,H,W,ld] from dout [N,H/pool,W/pool,d] (see lsam_attention_bwd / lsam_attention_bwd_qpool)."""
    _need_gpu(qkv, bias, dout_t, lse)
    bias, lse = _f32c(bias), _f32c(lse)
    N, H, W, C3 = qkv.shape
    d = C3 // 3
    ld = C3 if ld_out is None else ld_out
    dqkv = torch.zeros(N, H, W, ld, dtype=qkv.dtype, device=qkv.device)
    if pool == 1:
        check(lib().lsam_attention_bwd(dtype_code(qkv.dtype), ptr(qkv), C3, ptr(bias), ptr(dout_t), d,
                                       ptr(lse), ptr(dqkv), ld, N, H, W, ws, nh, d // nh, stream()))
    else:
        check(lib().lsam_attention_bwd_qpool(dtype_code(qkv.dtype), ptr(qkv), C3, ptr(bias), ptr(dout_t), d,
                                             ptr(lse), ptr(dqkv), ld, N, H, W, ws, pool, nh, d // nh, stream()))
    return dqkv


def maxpool2x2(x: torch.Tensor) -> torch.Tensor:
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    N, H, W, Cc = x.shape
    out = torch.empty(N, H // 2, W // 2, Cc, dtype=torch.float32, device=x.device)
    check(lib().lsam_maxpool2x2(ptr(x), ptr(out), N, H, W, Cc, stream()))
    return out


def spatial_mean(x: torch.Tensor) -> torch.Tensor:
    """fp32 NHWC [N,H,W,C] -> [N,C] mean over the grid (see lsam_spatial_mean)."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    N, H, W, Cc = x.shape
    out = torch.empty(N, Cc, dtype=torch.float32, device=x.device)
    check(lib().lsam_spatial_mean(ptr(x), ptr(out), N, H * W, Cc, stream()))
    return out


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    _need_gpu(x)
    rows, cols, lds = _rows2d(x)
    out = torch.empty(rows, cols, dtype=dtype, device=x.device)
    check(lib().lsam_cast(dtype_code(dtype), ptr(x), lds, ptr(out), cols, rows, cols, stream()))
    return out


def im2col_patch(img: torch.Tensor, dtype: torch.dtype, ldc: int = 160) -> torch.Tensor:
    _need_gpu(img)
    assert img.dtype == torch.float32 and img.is_contiguous()
    N, _, S, _ = img.shape
    cols = torch.empty(N * (S // 4) ** 2, ldc, dtype=dtype, device=img.device)
    check(lib().lsam_im2col_patch(dtype_code(dtype), ptr(img), ptr(cols), ldc, N, S, stream()))
    return cols


def outer_reduce(X: torch.Tensor, Y: torch.Tensor, R: int, transpose_out: bool) -> torch.Tensor:
    """G[c][k] = sum_m X[m][c] Y[m][k]; returns [C,R] or, with transpose_out, [R,C]."""
    _need_gpu(X, Y)
    M, Cc, ldx = _rows2d(X)
    _, _, ldy = _rows2d(Y)
    ws = torch.empty(lib().lsam_outer_reduce_ws(M, Cc), dtype=torch.uint8, device=X.device)
    if transpose_out:
        out = torch.empty(R, Cc, dtype=torch.float32, device=X.device)
        cs, ks = 1, Cc
    else:
        out = torch.empty(Cc, R, dtype=torch.float32, device=X.device)
        cs, ks = R, 1
    check(lib().lsam_outer_reduce(dtype_code(X.dtype), ptr(X), ldx, ptr(Y), ldy, M, Cc, R, ptr(out), cs, ks,
                                  ptr(ws), stream()))
    return out


def window_block_pack(qkv_w, qkv_b, proj_w, proj_b, nh: int):
    """fp32 module parameters -> (weight image uint8, bias table fp32) of the fused window-block kernel."""
    _need_gpu(qkv_w, qkv_b, proj_w, proj_b)
    d = proj_w.shape[0]
    hd = d // nh
    nbytes, nfl = lib().lsam_window_block_image_bytes(nh), lib().lsam_window_block_bias_floats(nh)
    if nbytes == 0:
        raise _C.LsamError(f"fused window block: {nh} heads not supported")
    img = torch.empty(nbytes, dtype=torch.uint8, device=qkv_w.device)
    bias = torch.empty(nfl, dtype=torch.float32, device=qkv_w.device)
    args = [t.detach().float().contiguous() for t in (qkv_w, qkv_b, proj_w, proj_b)]
    check(lib().lsam_window_block_pack(*[ptr(t) for t in args], nh, hd, ptr(img), ptr(bias), stream()))
    return img, bias


def window_block_fwd(x, norm1_w, norm1_b, img, bias, nh: int, eps: float = 1e-6, out=None):
    """y = x + proj(attn(qkv(LN1(x)))) for x fp32 [N,H,W,d] with 8x8 windows (see lsam_window_block_fwd)."""
    _need_gpu(x, norm1_w, norm1_b, img, bias)
    norm1_w, norm1_b = _f32c(norm1_w), _f32c(norm1_b)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    N, H, W, d = x.shape
    y = torch.empty_like(x) if out is None else out
    check(lib().lsam_window_block_fwd(ptr(x), ptr(y), ptr(norm1_w), ptr(norm1_b), ptr(img), ptr(bias), N, H, W, nh,
                                      d // nh, eps, stream()))
    return y


def mlp_block_pack(fc1_w, fc1_b, fc2_w, fc2_b):
    _need_gpu(fc1_w, fc1_b, fc2_w, fc2_b)
    d = fc2_w.shape[0]
    nbytes, nfl = lib().lsam_mlp_block_image_bytes(d), lib().lsam_mlp_block_bias_floats(d)
    if nbytes == 0:
        raise _C.LsamError(f"fused MLP block: width {d} not supported")
    img = torch.empty(nbytes, dtype=torch.uint8, device=fc1_w.device)
    bias = torch.empty(nfl, dtype=torch.float32, device=fc1_w.device)
    args = [t.detach().float().contiguous() for t in (fc1_w, fc1_b, fc2_w, fc2_b)]
    check(lib().lsam_mlp_block_pack(*[ptr(t) for t in args], d, ptr(img), ptr(bias), stream()))
    return img, bias


def mlp_block_fwd(x, norm2_w, norm2_b, img, bias, eps: float = 1e-6, out=None):
    """y = x + fc2(GELU(fc1(LN2(x)))) for x fp32 [M, d] (see lsam_mlp_block_fwd)."""
    _need_gpu(x, norm2_w, norm2_b, img, bias)
    norm2_w, norm2_b = _f32c(norm2_w), _f32c(norm2_b)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    M, d = x.shape
    y = torch.empty_like(x) if out is None else out
    check(lib().lsam_mlp_block_fwd(ptr(x), ptr(y), ptr(norm2_w), ptr(norm2_b), ptr(img), ptr(bias), M, d, eps,
                                   stream()))
    return y


def mlp_block_fwd_next_norm(x, norm2_w, norm2_b, img, bias, next_norm_w, next_norm_b, eps: float = 1e-6):
    """(y, LayerNorm_next(y) as bf16): lsam_mlp_block_fwd plus the following block's norm1 from the same registers."""
    _need_gpu(x, norm2_w, norm2_b, img, bias, next_norm_w, next_norm_b)
    norm2_w, norm2_b, next_norm_w, next_norm_b = _f32c(norm2_w), _f32c(norm2_b), _f32c(next_norm_w), _f32c(next_norm_b)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    M, d = x.shape
    y = torch.empty_like(x)
    xn = torch.empty(M, d, dtype=torch.bfloat16, device=x.device)
    check(lib().lsam_mlp_block_fwd_next_norm(ptr(x), ptr(y), ptr(norm2_w), ptr(norm2_b), ptr(img), ptr(bias), ptr(xn), d,
                                             ptr(next_norm_w), ptr(next_norm_b), M, d, eps, stream()))
    return y, xn


def window_block_attention_fwd(x, norm1_w, norm1_b, img, bias, nh: int, eps: float = 1e-6):
    """a = attn(qkv(LN1(x))) as bf16 [N,H,W,d]: the attention-only form of window_block_fwd (see lsam_window_block_attention_fwd)."""
    _need_gpu(x, norm1_w, norm1_b, img, bias)
    norm1_w, norm1_b = _f32c(norm1_w), _f32c(norm1_b)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    N, H, W, d = x.shape
    a = torch.empty(N, H, W, d, dtype=torch.bfloat16, device=x.device)
    check(lib().lsam_window_block_attention_fwd(ptr(x), ptr(a), ptr(norm1_w), ptr(norm1_b), ptr(img), ptr(bias), N, H, W, nh,
                                                d // nh, eps, stream()))
    return a


def mlp_block_proj_pack(proj_w, proj_b):
    """attn.proj as one chunk of the MLP block kernel's weight stream (see lsam_mlp_block_fwd_proj): (image, bias)."""
    _need_gpu(proj_w, proj_b)
    d = proj_w.shape[0]
    nbytes = lib().lsam_mlp_block_proj_image_bytes(d)
    if nbytes == 0:
        raise _C.LsamError(f"fused MLP block with the projection in front: width {d} not supported")
    img = torch.empty(nbytes, dtype=torch.uint8, device=proj_w.device)
    pb = torch.empty(16 * 14, dtype=torch.float32, device=proj_w.device)
    check(lib().lsam_mlp_block_proj_pack(ptr(_f32c(proj_w)), ptr(_f32c(proj_b)), d, ptr(img), ptr(pb), stream()))
    return img, pb


def mlp_block_fwd_proj(x, a, norm2_w, norm2_b, img, bias, pimg, pbias, next_norm=None, eps: float = 1e-6):
    """y1 = x + proj(a); y = y1 + fc2(GELU(fc1(LN2(y1)))) (and LayerNorm_next(y) as bf16 with next_norm = (w, b)):
    x fp32 [M, d], a bf16 [M, d] (see lsam_mlp_block_fwd_proj)."""
    _need_gpu(x, a, norm2_w, norm2_b, img, bias, pimg, pbias)
    norm2_w, norm2_b = _f32c(norm2_w), _f32c(norm2_b)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    assert a.dtype == torch.bfloat16 and a.is_contiguous() and a.shape == x.shape
    M, d = x.shape
    y = torch.empty_like(x)
    xn, nw, nb = None, None, None
    if next_norm is not None:
        nw, nb = _f32c(next_norm[0]), _f32c(next_norm[1])
        xn = torch.empty(M, d, dtype=torch.bfloat16, device=x.device)
    check(lib().lsam_mlp_block_fwd_proj(ptr(x), ptr(a), d, ptr(y), ptr(norm2_w), ptr(norm2_b), ptr(img), ptr(bias), ptr(pimg),
                                        ptr(pbias), ptr(xn) if xn is not None else None, d, ptr(nw) if nw is not None else None,
                                        ptr(nb) if nb is not None else None, M, d, eps, stream()))
    return (y, xn) if next_norm is not None else y


def lateral_conv_supported(in_channels: int, out_channels: int) -> bool:
    return lib().lsam_lateral_conv_image_bytes(in_channels, out_channels) > 0


def lateral_conv_pack(weight: torch.Tensor) -> torch.Tensor:
    """Conv2d(K, 256, 1) weight -> LDS-resident bf16 weight image of lsam_lateral_conv_fwd (K <= 224)."""
    _need_gpu(weight)
    co, K = weight.shape[0], weight.shape[1]
    nbytes = lib().lsam_lateral_conv_image_bytes(K, co)
    if nbytes == 0:
        raise _C.LsamError(f"lateral conv: {K} -> {co} channels not supported")
    img = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    w = _f32c(weight.reshape(co, K))
    check(lib().lsam_lateral_conv_pack(ptr(w), K, co, ptr(img), stream()))
    return img


def lateral_conv_fwd(x2d: torch.Tensor, img: torch.Tensor, bias: torch.Tensor, co: int) -> torch.Tensor:
    _need_gpu(x2d, img, bias)
    assert x2d.dtype == torch.float32 and x2d.is_contiguous() and x2d.dim() == 2
    M, K = x2d.shape
    b = _f32c(bias)
    out = torch.empty(M, co, dtype=torch.bfloat16, device=x2d.device)
    check(lib().lsam_lateral_conv_fwd(ptr(x2d), ptr(img), ptr(b), ptr(out), M, K, co, stream()))
    return out


def lateral_conv(x2d: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor):
    """bf16 [M, 256] = x2d (fp32 [M, K]) @ weight.T + bias in one kernel (see lsam_lateral_conv_fwd)."""
    return lateral_conv_fwd(x2d, lateral_conv_pack(weight), bias, weight.shape[0])


def stream_linear_pack(weight: torch.Tensor) -> torch.Tensor:
    """Linear(K <= 224, N) weight -> the LDS-resident bf16 weight image of lsam_stream_linear_fwd."""
    _need_gpu(weight)
    n, k = weight.shape
    nbytes = lib().lsam_stream_linear_image_bytes(k, n)
    if nbytes == 0:
        raise _C.LsamError(f"stream linear: {k} -> {n} features not supported")
    img = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    w = _f32c(weight)
    check(lib().lsam_stream_linear_pack(ptr(w), k, n, ptr(img), stream()))
    return img


def stream_linear_fwd(A: torch.Tensor, img: torch.Tensor, bias: torch.Tensor, k: int, n: int, out: Optional[torch.Tensor] = None):
    """bf16 [M, n] = A[:, :k] (bf16) @ W.T + bias with the weights resident in LDS (see lsam_stream_linear_fwd)."""
    _need_gpu(A, img, bias)
    assert A.dtype == torch.bfloat16
    M, _, lda = _rows2d(A)
    b = _f32c(bias)
    if out is None:
        out = torch.empty(M, n, dtype=torch.bfloat16, device=A.device)
    _, _, ldo = _rows2d(out)
    check(lib().lsam_stream_linear_fwd(ptr(A), lda, ptr(img), ptr(b), ptr(out), ldo, M, k, n, stream()))
    return out


def patch_embed_pack(proj_w):
    """PatchEmbed proj.weight [E,3,7,7] fp32 -> fragment-order bf16 weight image of the direct convolution."""
    _need_gpu(proj_w)
    e0 = proj_w.shape[0]
    nbytes = lib().lsam_patch_embed_image_bytes(e0)
    if nbytes == 0:
        raise _C.LsamError(f"direct patch embed: embed_dim {e0} not supported")
    img = torch.empty(nbytes, dtype=torch.uint8, device=proj_w.device)
    w = proj_w.detach().float().contiguous()
    check(lib().lsam_patch_embed_pack(ptr(w), e0, ptr(img), stream()))
    return img


def patch_embed_fwd(frames, img, proj_b, pos):
    """frames fp32 [N,3,S,S] -> fp32 [N, S/4, S/4, E] = conv7x7/4 (bf16 MFMA) + bias + pos (see lsam_patch_embed_fwd)."""
    _need_gpu(frames, img, proj_b, pos)
    assert frames.dtype == torch.float32 and frames.is_contiguous() and frames.shape[1] == 3
    N, _, S, S2 = frames.shape
    assert S == S2 and S % 4 == 0
    e0 = proj_b.numel()
    G = S // 4
    pos = pos.detach().float().contiguous()
    assert pos.numel() == G * G * e0
    b = proj_b.detach().float().contiguous()
    out = torch.empty(N, G, G, e0, dtype=torch.float32, device=frames.device)
    check(lib().lsam_patch_embed_fwd(ptr(frames), ptr(img), ptr(b), ptr(pos), ptr(out), N, S, e0, stream()))
    return out


def tail_block_pack(proj_w, proj_b, norm2_w, norm2_b, fc1_w, fc1_b, fc2_w, fc2_b):
    """fp32 module parameters -> (streamed weight image uint8, packed fp32 vectors) of the fused block tail."""
    params = (proj_w, proj_b, norm2_w, norm2_b, fc1_w, fc1_b, fc2_w, fc2_b)
    _need_gpu(*params)
    d = proj_w.shape[0]
    nbytes, nfl = lib().lsam_tail_block_image_bytes(d), lib().lsam_tail_block_vec_floats(d)
    if nbytes == 0:
        raise _C.LsamError(f"fused block tail: width {d} not supported")
    img = torch.empty(nbytes, dtype=torch.uint8, device=proj_w.device)
    vec = torch.empty(nfl, dtype=torch.float32, device=proj_w.device)
    args = [t.detach().float().contiguous() for t in params]
    check(lib().lsam_tail_block_pack(*[ptr(t) for t in args], d, ptr(img), ptr(vec), stream()))
    return img, vec


def tail_block_fwd(a, res, img, vec, next_norm=None, eps: float = 1e-6, out=None):
    """out = y + fc2(GELU(fc1(LN2(y)))), y = res + proj(a); with next_norm = (weight, bias) also returns
    LayerNorm(out) in bf16 (see lsam_tail_block_fwd)."""
    _need_gpu(a, res, img, vec)
    assert a.dtype == torch.bfloat16 and res.dtype == torch.float32 and a.dim() == 2 and res.is_contiguous()
    assert a.stride(1) == 1
    M, d = res.shape
    y = torch.empty_like(res) if out is None else out
    xn = None
    gw = gb = None
    if next_norm is not None:
        gw, gb = (t.detach().float().contiguous() for t in next_norm)
        xn = torch.empty(M, d, dtype=torch.bfloat16, device=res.device)
    check(lib().lsam_tail_block_fwd(ptr(a), a.stride(0), ptr(res), ptr(y), ptr(img), ptr(vec),
                                    ptr(xn) if xn is not None else None, d, ptr(gw) if xn is not None else None,
                                    ptr(gb) if xn is not None else None, M, d, eps, stream()))
    return (y, xn) if next_norm is not None else y


def neck_conv_pack(weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Conv2d(K, N, 1) weight (fp32 [N, K] or [N, K, 1, 1]) -> fragment-ordered weight image of lsam_neck_level_fwd."""
    _need_gpu(weight)
    co, K = weight.shape[0], weight.shape[1]
    nbytes = lib().lsam_neck_conv_image_bytes(dtype_code(dtype), K, co)
    if nbytes == 0:
        raise _C.LsamError(f"neck level: {K} -> {co} channels not supported")
    img = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    w = _f32c(weight.reshape(co, K))
    check(lib().lsam_neck_conv_pack(dtype_code(dtype), ptr(w), K, co, ptr(img), stream()))
    return img


def neck_level_fwd(x_nhwc: torch.Tensor, img: torch.Tensor, bias: torch.Tensor, dtype: torch.dtype, co: int,
                   top: Optional[torch.Tensor] = None) -> torch.Tensor:
    """T [N,H,W,co] = conv1x1(x) + bias (+ nearest-x2 upsample of `top` [N,H/2,W/2,co]) (see lsam_neck_level_fwd)."""
    _need_gpu(x_nhwc, img, bias, top)
    assert x_nhwc.dtype == torch.float32 and x_nhwc.is_contiguous() and x_nhwc.dim() == 4
    N, H, W, K = x_nhwc.shape
    if top is not None:
        assert top.dtype == dtype and top.is_contiguous() and tuple(top.shape) == (N, H // 2, W // 2, co)
    out = torch.empty(N, H, W, co, dtype=dtype, device=x_nhwc.device)
    check(lib().lsam_neck_level_fwd(dtype_code(dtype), ptr(x_nhwc), ptr(img), ptr(bias), ptr(top), ptr(out), N, H, W, K, co,
                                    stream()))
    return out


def neck_conv_pack_split(weight: torch.Tensor) -> torch.Tensor:
    """[hi | lo] bf16 fragment image of a [co, K] (or [co, K, 1, 1]) convolution weight for neck_level_fwd_split."""
    _need_gpu(weight)
    w = weight.detach().to(torch.float32).reshape(weight.shape[0], -1).contiguous()
    co, K = w.shape
    nbytes = lib().lsam_neck_conv_split_image_bytes(K, co)
    if nbytes == 0:
        raise _C.LsamError(f"neck_conv_pack_split: {K} -> {co} channels not supported")
    img = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w.device)
    check(lib().lsam_neck_conv_pack_split(ptr(w), K, co, ptr(img), stream()))
    return img


def neck_level_fwd_split(x_nhwc: torch.Tensor, img: torch.Tensor, bias: torch.Tensor, co: int,
                         top32: Optional[torch.Tensor] = None, want32: bool = False):
    """(bf16 out, fp32 out or None) = lateral conv (+ nearest-x2 top-down add of the fp32 `top32`) with two-term bf16
    operands: fp32-accurate product, one rounding at the bf16 store (see lsam_neck_level_fwd_split)."""
    _need_gpu(x_nhwc, img, bias, top32)
    assert x_nhwc.dtype == torch.float32 and x_nhwc.is_contiguous() and x_nhwc.dim() == 4
    N, H, W, K = x_nhwc.shape
    if top32 is not None:
        assert top32.dtype == torch.float32 and top32.is_contiguous() and tuple(top32.shape) == (N, H // 2, W // 2, co)
    out = torch.empty(N, H, W, co, dtype=torch.bfloat16, device=x_nhwc.device)
    out32 = torch.empty(N, H, W, co, dtype=torch.float32, device=x_nhwc.device) if want32 else None
    check(lib().lsam_neck_level_fwd_split(ptr(x_nhwc), ptr(img), ptr(_f32c(bias)), ptr(top32), ptr(out), ptr(out32), N, H, W, K, co,
                                          stream()))
    return out, out32


def neck_topdown_bwd(g_fine: torch.Tensor, g_coarse: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Adjoint of the nearest-x2 upsample: [N,H/2,W/2,C] = 2x2 sums of g_fine [N,H,W,C] (+ g_coarse)."""
    _need_gpu(g_fine, g_coarse)
    assert g_fine.is_contiguous() and g_fine.dim() == 4
    N, H, W, Cc = g_fine.shape
    out = torch.empty(N, H // 2, W // 2, Cc, dtype=g_fine.dtype, device=g_fine.device)
    if g_coarse is not None:
        assert g_coarse.dtype == g_fine.dtype and g_coarse.is_contiguous() and g_coarse.shape == out.shape
    check(lib().lsam_neck_topdown_bwd(dtype_code(g_fine.dtype), ptr(g_fine), ptr(g_coarse), ptr(out), N, H // 2, W // 2, Cc,
                                      stream()))
    return out


def kv_window_attention_pack(qkv_w: torch.Tensor, qkv_b: torch.Tensor, nh: int) -> torch.Tensor:
    """Weight image of kv_window_attention_fwd from qkv.weight [3*nh*hd, dim] and qkv.bias."""
    _need_gpu(qkv_w, qkv_b)
    w, b = _f32c(qkv_w), _f32c(qkv_b)
    dim, hd = w.shape[1], w.shape[0] // (3 * nh)
    nbytes = lib().lsam_kv_window_attention_image_bytes(dim, nh, hd)
    if nbytes == 0:
        raise _C.LsamError(f"kv_window_attention: {nh} heads of {hd} over {dim} channels not supported")
    img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    check(lib().lsam_kv_window_attention_pack(ptr(w), ptr(b), dim, nh, hd, ptr(img), stream()))
    return img


def kv_window_attention_fwd(xn: torch.Tensor, q_pooled: torch.Tensor, img: torch.Tensor, nh: int) -> torch.Tensor:
    """bf16 [N, H/2, W/2, nh*hd] = 8x8-window attention of the pooled queries (fp32 [N, H/2, W/2, nh*hd]) against
    k | v = xn . W_kv^T + b_kv projected in the kernel; xn bf16 [N, H, W, dim] (see lsam_kv_window_attention_fwd)."""
    _need_gpu(xn, q_pooled, img)
    assert xn.dtype == torch.bfloat16 and xn.is_contiguous() and xn.dim() == 4
    assert q_pooled.dtype == torch.float32 and q_pooled.is_contiguous() and q_pooled.dim() == 4
    N, H, W, dim = xn.shape
    q = q_pooled.shape[-1]
    out = torch.empty(N, H // 2, W // 2, q, dtype=torch.bfloat16, device=xn.device)
    check(lib().lsam_kv_window_attention_fwd(ptr(xn), dim, ptr(q_pooled), q, ptr(img), ptr(out), q, N, H, W, dim, nh, q // nh,
                                             stream()))
    return out


def qkv_window4_attention_pack(qkv_w: torch.Tensor, qkv_b: torch.Tensor, nh: int, pooled_q: bool = False) -> torch.Tensor:
    """Weight image of qkv_window4_attention_fwd from qkv.weight [3*nh*hd, dim] and qkv.bias (k | v only with pooled_q)."""
    _need_gpu(qkv_w, qkv_b)
    w, b = _f32c(qkv_w), _f32c(qkv_b)
    dim, hd = w.shape[1], w.shape[0] // (3 * nh)
    nbytes = lib().lsam_qkv_window4_attention_image_bytes(dim, nh, hd, int(pooled_q))
    if nbytes == 0:
        raise _C.LsamError(f"qkv_window4_attention: {nh} heads of {hd} over {dim} channels not supported")
    img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    check(lib().lsam_qkv_window4_attention_pack(ptr(w), ptr(b), dim, nh, hd, int(pooled_q), ptr(img), stream()))
    return img


def qkv_window4_attention_fwd(xn: torch.Tensor, img: torch.Tensor, nh: int, hd: int, q_pooled: Optional[torch.Tensor] = None) -> torch.Tensor:
    """4x4-window attention with the projection inside the kernel (see lsam_qkv_window4_attention_fwd): xn bf16 [N, H, W, dim];
    without q_pooled bf16 [N, H, W, nh*hd] from qkv = xn . W^T + b; with q_pooled (fp32 [N, H/2, W/2, nh*hd]) bf16
    [N, H/2, W/2, nh*hd] from k | v = xn . W_kv^T + b_kv and the window's four pooled queries."""
    _need_gpu(xn, img)
    assert xn.dtype == torch.bfloat16 and xn.is_contiguous() and xn.dim() == 4
    N, H, W, dim = xn.shape
    q = nh * hd
    if q_pooled is None:
        out = torch.empty(N, H, W, q, dtype=torch.bfloat16, device=xn.device)
        qp, ldq = None, 0
    else:
        _need_gpu(q_pooled)
        assert q_pooled.dtype == torch.float32 and q_pooled.is_contiguous() and tuple(q_pooled.shape) == (N, H // 2, W // 2, q)
        out = torch.empty(N, H // 2, W // 2, q, dtype=torch.bfloat16, device=xn.device)
        qp, ldq = ptr(q_pooled), q
    check(lib().lsam_qkv_window4_attention_fwd(ptr(xn), dim, qp, ldq, ptr(img), ptr(out), q, N, H, W, dim, nh, hd, stream()))
    return out


# ---- SAM heads pieces (csrc/heads.hip) ---------------------------------------------------------------------------
def add_rows(a: torch.Tensor, b: Optional[torch.Tensor], out_dtype: torch.dtype) -> torch.Tensor:
    """out[m] = a[m] + b[m % rows(b)] as out_dtype; a fp32 / bf16 [M, C] (row-major), b fp32 [R, C] or None (a cast)."""
    _need_gpu(a, b)
    M, Cc, lda = _rows2d(a)
    b = _f32rows(b)
    out = torch.empty(M, Cc, dtype=out_dtype, device=a.device)
    check(lib().lsam_add_rows(dtype_code(a.dtype), ptr(a), lda, ptr(b), b.stride(0) if b is not None else 0,
                              b.shape[0] if b is not None else 1, dtype_code(out_dtype), ptr(out), Cc, M, Cc, stream()))
    return out


def mha(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, nh: int) -> torch.Tensor:
    """softmax(q k^T / sqrt(hd)) v for q [B, nq, C], k / v [B, nk, C] (contiguous, same dtype), C = nh * hd."""
    _need_gpu(q, k, v)
    assert q.dim() == 3 and k.dim() == 3 and v.dim() == 3 and q.dtype == k.dtype == v.dtype
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    B, nq, Cc = q.shape
    nk = k.shape[1]
    out = torch.empty_like(q)
    check(lib().lsam_mha_fwd(dtype_code(q.dtype), ptr(q), Cc, nq * Cc, ptr(k), Cc, nk * Cc, ptr(v), Cc, nk * Cc, ptr(out), Cc,
                             nq * Cc, B, nq, nk, nh, Cc // nh, stream()))
    return out


def pe_encode(gauss: torch.Tensor, coords01: torch.Tensor) -> torch.Tensor:
    """[M, 2F] = [sin | cos](2 pi (2 c - 1) @ gauss) for gauss fp32 [2, F], coords01 fp32 [M, 2]."""
    _need_gpu(gauss, coords01)
    g, c = _f32c(gauss), _f32c(coords01)
    M, F_ = c.shape[0], g.shape[1]
    out = torch.empty(M, 2 * F_, dtype=torch.float32, device=c.device)
    check(lib().lsam_pe_encode(ptr(g), ptr(c), ptr(out), M, F_, stream()))
    return out


def upscale2x2(g: torch.Tensor, skip: Optional[torch.Tensor], ln: Optional[Tuple[torch.Tensor, torch.Tensor]], eps: float,
               B: int, H: int, W: int, Cc: int, dtype: torch.dtype) -> torch.Tensor:
    """T [B, 2H, 2W, C] = GELU(LN2d?(pixel_shuffle(g) + skip)) for g fp32 [B*H*W, 4C] (see lsam_upscale2x2_fwd)."""
    _need_gpu(g, skip)
    assert g.dtype == torch.float32 and g.is_contiguous() and tuple(g.shape) == (B * H * W, 4 * Cc)
    if skip is not None:
        assert skip.dtype == dtype and skip.is_contiguous() and skip.numel() == B * 4 * H * W * Cc
    w, b = (_f32c(ln[0]), _f32c(ln[1])) if ln is not None else (None, None)
    out = torch.empty(B, 2 * H, 2 * W, Cc, dtype=dtype, device=g.device)
    check(lib().lsam_upscale2x2_fwd(dtype_code(dtype), ptr(g), ptr(skip), ptr(w), ptr(b), eps, ptr(out), B, H, W, Cc, stream()))
    return out


def mask_gate(logits: torch.Tensor, obj: Optional[torch.Tensor], first: int, n_masks: int, no_obj: float) -> torch.Tensor:
    """fp32 [B, n_masks, HW] from logits fp32 [B, HW, L]: columns first.., replaced by no_obj where obj[b] <= 0."""
    _need_gpu(logits, obj)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.dim() == 3
    B, HW, L = logits.shape
    o = _f32c(obj.reshape(-1)) if obj is not None else None
    out = torch.empty(B, n_masks, HW, dtype=torch.float32, device=logits.device)
    check(lib().lsam_mask_gate(ptr(logits), L, ptr(o), ptr(out), B, HW, first, n_masks, no_obj, stream()))
    return out


def bilinear_upsample(x: torch.Tensor, scale: int) -> torch.Tensor:
    """F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=False) for fp32 [B, C, H, W]."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    B, Cc, H, W = x.shape
    out = torch.empty(B, Cc, H * scale, W * scale, dtype=torch.float32, device=x.device)
    check(lib().lsam_bilinear_upsample(ptr(x), ptr(out), B * Cc, H, W, scale, stream()))
    return out


# ---- input gradients of the SAM heads, mask-loss sums (csrc/heads_bwd.hip) -------------------------------------------
def act_bwd(dy: torch.Tensor, y: torch.Tensor, kind: int) -> torch.Tensor:
    """fp32 dx = dy * act'(.) from the activation's output y (kind 0 ReLU, 1 sigmoid); same shape, contiguous."""
    _need_gpu(dy, y)
    dy = _f32c(dy)
    y = y.detach().contiguous()
    assert dy.shape == y.shape and dy.numel() % 4 == 0
    dx = torch.empty_like(dy)
    check(lib().lsam_act_bwd(kind, ptr(dy), dtype_code(y.dtype), ptr(y), ptr(dx), dy.numel(), stream()))
    return dx


def mha_bwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, nh: int):
    """(dq, dk, dv) fp32 of ops.mha for contiguous [B, n, C] operands."""
    _need_gpu(q, k, v, out, dout)
    q, k, v, out = q.detach().contiguous(), k.detach().contiguous(), v.detach().contiguous(), out.detach().contiguous()
    dout = _f32c(dout)
    B, nq, Cc = q.shape
    nk = k.shape[1]
    if nq > 16 and nk > 16:
        # the kernels keep one side (<= 16 rows) in LDS.  Attention rows are independent, so more than 16 tokens on BOTH sides
        # (the two-way transformer's self-attention with more than 9 accumulated clicks) is exact as a sum over query groups
        # of 16: each group gives its own dq rows and a dk / dv contribution.
        dq = torch.empty(B, nq, Cc, dtype=torch.float32, device=q.device)
        dk = dv = None
        for o in range(0, nq, 16):
            e = min(nq, o + 16)
            dqc, dkc, dvc = mha_bwd(q[:, o:e], k, v, out[:, o:e], dout[:, o:e], nh)
            dq[:, o:e] = dqc
            dk, dv = (dkc, dvc) if dk is None else (dk.add_(dkc), dv.add_(dvc))
        return dq, dk, dv
    dq = torch.empty(B, nq, Cc, dtype=torch.float32, device=q.device)
    dk = torch.empty(B, nk, Cc, dtype=torch.float32, device=q.device)
    dv = torch.empty(B, nk, Cc, dtype=torch.float32, device=q.device)
    check(lib().lsam_mha_bwd(dtype_code(q.dtype), ptr(q), ptr(k), ptr(v), ptr(out), ptr(dout), ptr(dq), ptr(dk), ptr(dv), B, nq, nk,
                             nh, Cc // nh, stream()))
    return dq, dk, dv


def upscale2x2_bwd(dout: torch.Tensor, g: torch.Tensor, skip: Optional[torch.Tensor],
                   ln: Optional[Tuple[torch.Tensor, torch.Tensor]], eps: float, B: int, H: int, W: int, Cc: int,
                   dtype: torch.dtype) -> torch.Tensor:
    """fp32 d_g [B*H*W, 4C] of ops.upscale2x2 (dout [B, 2H, 2W, C])."""
    _need_gpu(dout, g, skip)
    dout = _f32c(dout)
    assert g.dtype == torch.float32 and g.is_contiguous() and dout.numel() == B * 4 * H * W * Cc
    if skip is not None:
        assert skip.dtype == dtype and skip.is_contiguous()
    w, b = (_f32c(ln[0]), _f32c(ln[1])) if ln is not None else (None, None)
    dg = torch.empty_like(g)
    check(lib().lsam_upscale2x2_bwd(dtype_code(dtype), ptr(dout), ptr(g), ptr(skip), ptr(w), ptr(b), eps, ptr(dg), B, H, W, Cc,
                                    stream()))
    return dg


def mask_gate_bwd(d_masks: torch.Tensor, obj: Optional[torch.Tensor], first: int, L: int) -> torch.Tensor:
    """fp32 d_logits [B, HW, L] from d_masks fp32 [B, n_masks, HW]."""
    _need_gpu(d_masks, obj)
    d_masks = _f32c(d_masks)
    B, nm, HW = d_masks.shape
    o = _f32c(obj.reshape(-1)) if obj is not None else None
    out = torch.empty(B, HW, L, dtype=torch.float32, device=d_masks.device)
    check(lib().lsam_mask_gate_bwd(ptr(d_masks), ptr(o), ptr(out), L, B, HW, first, nm, stream()))
    return out


def bilinear_upsample_bwd(dout: torch.Tensor, scale: int) -> torch.Tensor:
    _need_gpu(dout)
    dout = _f32c(dout)
    B, Cc, Ho, Wo = dout.shape
    din = torch.empty(B, Cc, Ho // scale, Wo // scale, dtype=torch.float32, device=dout.device)
    check(lib().lsam_bilinear_upsample_bwd(ptr(dout), ptr(din), B * Cc, Ho // scale, Wo // scale, scale, stream()))
    return din


def mask_loss_sums(logits: torch.Tensor, targets: torch.Tensor, alpha: float, gamma: float) -> torch.Tensor:
    """fp32 [N, M, 6] per-candidate sums of the mask losses (see lsam_mask_loss_sums); logits [N, M, H, W], targets [N, H, W]."""
    _need_gpu(logits, targets)
    x, t = _f32c(logits), _f32c(targets)
    N, M = x.shape[:2]
    HW = x[0, 0].numel()
    assert t.numel() == N * HW
    sums = torch.empty(N, M, 6, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib().lsam_mask_loss_ws_floats(N, M), dtype=torch.float32, device=x.device)
    check(lib().lsam_mask_loss_sums(ptr(x), ptr(t), ptr(sums), ptr(ws), N, M, HW, alpha, gamma, stream()))
    return sums


def mask_loss_bwd(logits: torch.Tensor, targets: torch.Tensor, coef: torch.Tensor, alpha: float, gamma: float) -> torch.Tensor:
    _need_gpu(logits, targets, coef)
    x, t, c = _f32c(logits), _f32c(targets), _f32c(coef)
    N, M = x.shape[:2]
    HW = x[0, 0].numel()
    dx = torch.empty_like(x)
    check(lib().lsam_mask_loss_bwd(ptr(x), ptr(t), ptr(c), ptr(dx), N, M, HW, alpha, gamma, stream()))
    return dx


def batched_outer(G: torch.Tensor, U: torch.Tensor) -> torch.Tensor:
    """fp32 [B, L, C] = sum_m G[b, m, l] U[b, m, c] for G fp32 [B, M, L] (L <= 16), U T [B, M, C] (C <= 64)."""
    _need_gpu(G, U)
    G = _f32c(G)
    U = U.detach().contiguous()
    B, M, L = G.shape
    Cc = U.shape[2]
    out = torch.empty(B, L, Cc, dtype=torch.float32, device=G.device)
    ws = torch.empty(lib().lsam_batched_outer_ws_floats(B, L, Cc), dtype=torch.float32, device=G.device)
    check(lib().lsam_batched_outer(dtype_code(U.dtype), ptr(G), ptr(U), ptr(out), ptr(ws), B, M, L, Cc, stream()))
    return out
