// Small kernels of the SAM prompt encoder / mask decoder (the step after the image encoder in train_lora_sam.py's model:
// SAM2Base._forward_sam_heads, sam2_base.py:257-413).  The heads are ~1 % of a frame's FLOPs: every Linear, 1x1 /
// transposed convolution and hyper-network product runs through lsam_linear, every LayerNorm through lsam_layernorm_fwd;
// what is left are the pieces below, all HBM-/latency-bound and written for coalesced 16-byte accesses:
//   add_rows      out = a + b[m % rows]      (`queries + query_pe`, `keys + key_pe`, `+ no_mem_embed`; sam/transformer.py:88-101,
//                                             sam2_base.py:669-674), fp32 or bf16 in and out
//   mha           F.scaled_dot_product_attention of sam/transformer.py:233-265 (Attention.forward) for token <-> image
//                 attention: 8 heads of 16 / 32 channels, 8..9 tokens against 4096 image positions and back
//   pe_encode     PositionEmbeddingRandom._pe_encoding (position_encoding.py:147-155): sin / cos of 2 pi (2 c - 1) G
//   upscale2x2    the tail of ConvTranspose2d(k=2, s=2) taken as a GEMM: pixel shuffle + skip add (+ LayerNorm2d) + GELU
//                 (mask_decoder.py:216-225: `act1(ln1(dc1(src) + feat_s1))`, `act2(dc2(upscaled) + feat_s0)`)
//   mask_gate     object-score gating of the mask logits (sam2_base.py:359-368) + NCHW emission of the candidates
//   bilinear_up4  F.interpolate(low_res, size=(1024, 1024), mode="bilinear", align_corners=False) (sam2_base.py:372-377)
#include "kernels.hpp"

namespace lsam {

// ---------------------------------------------------------------------------------------------
template <typename TA, typename TO>
__global__ void add_rows_kernel(const TA* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int brows,
                                TO* __restrict__ out, int ldo, int M, int c4) {
    const long total = (long)M * c4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / c4), c = (int)(i % c4) * 4;
        f32x4 v = load4(a + (size_t)m * lda + c);
        if (b) v += load4(b + (size_t)(m % brows) * ldb + c);
        store4(out + (size_t)m * ldo + c, v);
    }
}

int launch_add_rows(int a_dtype, const void* a, int lda, const float* b, int ldb, int brows, int out_dtype, void* out, int ldo,
                    int M, int C, hipStream_t s) {
    LSAM_REQUIRE(a && out && M > 0 && C > 0, LSAM_ERR_BAD_ARG, "add_rows: null/empty");
    LSAM_REQUIRE(C % 4 == 0 && lda % 4 == 0 && ldo % 4 == 0 && (!b || (ldb % 4 == 0 && brows > 0)), LSAM_ERR_BAD_ARG,
                 "add_rows: C and leading dimensions must be multiples of 4");
    LSAM_REQUIRE((a_dtype == LSAM_F32 || a_dtype == LSAM_BF16) && (out_dtype == LSAM_F32 || out_dtype == LSAM_BF16),
                 LSAM_ERR_BAD_ARG, "add_rows: dtype");
    const long total = (long)M * (C / 4);
    const dim3 grid((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), blk(256);
#define ADD_CASE(TA, TO)                                                                                              \
    hipLaunchKernelGGL((add_rows_kernel<TA, TO>), grid, blk, 0, s, static_cast<const TA*>(a), lda, b, ldb, brows,     \
                       static_cast<TO*>(out), ldo, M, C / 4)
    if (a_dtype == LSAM_F32 && out_dtype == LSAM_F32) ADD_CASE(float, float);
    else if (a_dtype == LSAM_F32) ADD_CASE(float, bf16);
    else if (out_dtype == LSAM_F32) ADD_CASE(bf16, float);
    else ADD_CASE(bf16, bf16);
#undef ADD_CASE
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// ---------------------------------------------------------------------------------------------
// Multi-head attention on row-major [B][n][nh * HD] operands (separate q, k, v: cross attention).
// Few keys (<= 64: image -> token attention, token self-attention): one thread per (batch, head, query).
// Many keys (token -> image attention over 4096 positions): one wave per (batch, head, query), the lanes stride over the
// keys with a private running maximum and are merged at the end.
// ---------------------------------------------------------------------------------------------
struct MhaArgs {
    const void *q, *k, *v;
    void* out;
    int ldq, ldk, ldv, ldo;
    long qbs, kbs, vbs, obs;   // batch strides in elements
    int B, nq, nk, nh;
    float scale;
};

template <typename T, int HD>
__global__ __launch_bounds__(256) void mha_fewkeys_kernel(const MhaArgs p) {
    const long idx = blockIdx.x * 256L + threadIdx.x;
    const long total = (long)p.B * p.nh * p.nq;
    if (idx >= total) return;
    const int h = (int)(idx % p.nh);
    const int i = (int)((idx / p.nh) % p.nq);
    const int b = (int)(idx / ((long)p.nh * p.nq));
    const T* q = static_cast<const T*>(p.q) + b * p.qbs + (size_t)i * p.ldq + h * HD;
    const T* kb = static_cast<const T*>(p.k) + b * p.kbs + h * HD;
    const T* vb = static_cast<const T*>(p.v) + b * p.vbs + h * HD;
    float qv[HD];
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        const f32x4 t = load4(q + d);
#pragma unroll
        for (int r = 0; r < 4; ++r) qv[d + r] = t[r] * p.scale;
    }
    float m = -3.0e38f;
    for (int j = 0; j < p.nk; ++j) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const f32x4 t = load4(kb + (size_t)j * p.ldk + d);
#pragma unroll
            for (int r = 0; r < 4; ++r) s = fmaf(qv[d + r], t[r], s);
        }
        m = fmaxf(m, s);
    }
    float l = 0.f, o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    for (int j = 0; j < p.nk; ++j) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const f32x4 t = load4(kb + (size_t)j * p.ldk + d);
#pragma unroll
            for (int r = 0; r < 4; ++r) s = fmaf(qv[d + r], t[r], s);
        }
        const float e = __expf(s - m);
        l += e;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const f32x4 t = load4(vb + (size_t)j * p.ldv + d);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[d + r] = fmaf(e, t[r], o[d + r]);
        }
    }
    const float inv = 1.0f / l;
    T* out = static_cast<T*>(p.out) + b * p.obs + (size_t)i * p.ldo + h * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) store4(out + d, f32x4{o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv});
}

template <typename T, int HD>
__global__ __launch_bounds__(256) void mha_manykeys_kernel(const MhaArgs p) {
    const int lane = threadIdx.x & 63;
    const long idx = blockIdx.x * 4L + (threadIdx.x >> 6);
    const long total = (long)p.B * p.nh * p.nq;
    if (idx >= total) return;   // wave-uniform
    const int h = (int)(idx % p.nh);
    const int i = (int)((idx / p.nh) % p.nq);
    const int b = (int)(idx / ((long)p.nh * p.nq));
    const T* q = static_cast<const T*>(p.q) + b * p.qbs + (size_t)i * p.ldq + h * HD;
    const T* kb = static_cast<const T*>(p.k) + b * p.kbs + h * HD;
    const T* vb = static_cast<const T*>(p.v) + b * p.vbs + h * HD;
    float qv[HD];
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        const f32x4 t = load4(q + d);
#pragma unroll
        for (int r = 0; r < 4; ++r) qv[d + r] = t[r] * p.scale;
    }
    float m = -3.0e38f, l = 0.f, o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    for (int j = lane; j < p.nk; j += 64) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const f32x4 t = load4(kb + (size_t)j * p.ldk + d);
#pragma unroll
            for (int r = 0; r < 4; ++r) s = fmaf(qv[d + r], t[r], s);
        }
        if (s > m) {
            const float a = __expf(m - s);
            l *= a;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] *= a;
            m = s;
        }
        const float e = __expf(s - m);
        l += e;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const f32x4 t = load4(vb + (size_t)j * p.ldv + d);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[d + r] = fmaf(e, t[r], o[d + r]);
        }
    }
    float mm = m;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mm = fmaxf(mm, __shfl_xor(mm, off, 64));
    const float a = l > 0.f ? __expf(m - mm) : 0.f;   // lanes that saw no key contribute nothing
    l *= a;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        float x = o[d] * a;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        o[d] = x;
    }
    if (lane == 0) {
        const float inv = 1.0f / l;
        T* out = static_cast<T*>(p.out) + b * p.obs + (size_t)i * p.ldo + h * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) store4(out + d, f32x4{o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv});
    }
}

template <typename T> int launch_mha(const MhaArgs& a, int hd, hipStream_t s) {
    LSAM_REQUIRE(a.q && a.k && a.v && a.out, LSAM_ERR_BAD_ARG, "mha: null pointer");
    LSAM_REQUIRE(a.B > 0 && a.nq > 0 && a.nk > 0 && a.nh > 0, LSAM_ERR_BAD_ARG, "mha: empty shape");
    LSAM_REQUIRE(hd == 16 || hd == 32 || hd == 64, LSAM_ERR_UNSUPPORTED, "mha: head_dim %d (16, 32 or 64)", hd);
    LSAM_REQUIRE(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldo % 4 == 0 && a.qbs % 4 == 0 && a.kbs % 4 == 0 &&
                     a.vbs % 4 == 0 && a.obs % 4 == 0 && aligned16(a.q) && aligned16(a.k) && aligned16(a.v) && aligned16(a.out),
                 LSAM_ERR_BAD_ARG, "mha: strides must be multiples of 4 elements and pointers 16-byte aligned");
    const long total = (long)a.B * a.nh * a.nq;
    LSAM_REQUIRE(total < (1L << 31) - 256, LSAM_ERR_UNSUPPORTED, "mha: too many (batch, head, query) triples");
    ProfScope ps(s, 4.0 * total * a.nk * hd, PROF_ATTN_WIN, (int)(a.B * a.nq), a.nk, a.nh * hd);
#define MHA_CASE(HD)                                                                                                  \
    if (a.nk <= 64) hipLaunchKernelGGL((mha_fewkeys_kernel<T, HD>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a); \
    else hipLaunchKernelGGL((mha_manykeys_kernel<T, HD>), dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, a)
    if (hd == 16) { MHA_CASE(16); }
    else if (hd == 32) { MHA_CASE(32); }
    else { MHA_CASE(64); }
#undef MHA_CASE
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
template int launch_mha<float>(const MhaArgs&, int, hipStream_t);
template int launch_mha<bf16>(const MhaArgs&, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
__global__ void pe_encode_kernel(const float* __restrict__ gauss, const float* __restrict__ coords01, float* __restrict__ out,
                                 int M, int F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * F) return;
    const int m = i / F, f = i % F;
    const float cx = 2.0f * coords01[2 * m] - 1.0f, cy = 2.0f * coords01[2 * m + 1] - 1.0f;
    const float t = 6.283185307179586f * (cx * gauss[f] + cy * gauss[F + f]);
    out[(size_t)m * 2 * F + f] = sinf(t);
    out[(size_t)m * 2 * F + F + f] = cosf(t);
}
int launch_pe_encode(const float* gauss, const float* coords01, float* out, int M, int F, hipStream_t s) {
    LSAM_REQUIRE(gauss && coords01 && out && M > 0 && F > 0, LSAM_ERR_BAD_ARG, "pe_encode: null/empty");
    hipLaunchKernelGGL(pe_encode_kernel, dim3(cdiv(M * F, 256)), dim3(256), 0, s, gauss, coords01, out, M, F);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// ---------------------------------------------------------------------------------------------
// g fp32 [B*H*W][4*C] = the GEMM form of ConvTranspose2d(k=2, s=2) (columns (dy, dx, co), bias included);
// out[b][2y+dy][2x+dx][:] = GELU( LN2d?( g[b,y,x][(dy,dx),:] + skip[b][2y+dy][2x+dx][:] ) ), T NHWC.
// A group of C/4 lanes owns one output pixel (4 channels per lane), so LayerNorm2d is a group reduction.
// ---------------------------------------------------------------------------------------------
template <typename T, int C, bool LN>
__global__ __launch_bounds__(256) void upscale2x2_kernel(const float* __restrict__ g, const T* __restrict__ skip,
                                                         const float* __restrict__ lnw, const float* __restrict__ lnb, float eps,
                                                         T* __restrict__ out, int B, int H, int W) {
    constexpr int LPP = C / 4;                       // lanes per output pixel
    const long pix = (blockIdx.x * 256L + threadIdx.x) / LPP;
    const int li = threadIdx.x % LPP;
    const long npix = (long)B * 4 * H * W;
    const bool ok = pix < npix;
    const long pc = ok ? pix : npix - 1;
    const int X = (int)(pc % (2 * W)), Y = (int)((pc / (2 * W)) % (2 * H)), b = (int)(pc / ((long)4 * H * W));
    const size_t tok = ((size_t)b * H + (Y >> 1)) * W + (X >> 1);
    const int sub = (Y & 1) * 2 + (X & 1);
    f32x4 v = load4(g + tok * (4 * C) + sub * C + 4 * li);
    if (skip) v += load4(skip + (size_t)pc * C + 4 * li);
    if constexpr (LN) {
        float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
        for (int off = LPP / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) q += (v[r] - mean) * (v[r] - mean);
#pragma unroll
        for (int off = LPP / 2; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
        const float rstd = 1.0f / sqrtf(q / (float)C + eps);
        const f32x4 w = load4(lnw + 4 * li), bb = load4(lnb + 4 * li);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (v[r] - mean) * rstd * w[r] + bb[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = gelu_t<false>(v[r]);
    if (ok) store4(out + (size_t)pix * C + 4 * li, v);
}

template <typename T>
int launch_upscale2x2(const float* g, const void* skip, const float* lnw, const float* lnb, float eps, void* out, int B, int H,
                      int W, int C, hipStream_t s) {
    LSAM_REQUIRE(g && out && B > 0 && H > 0 && W > 0, LSAM_ERR_BAD_ARG, "upscale2x2: null/empty");
    LSAM_REQUIRE(C == 32 || C == 64, LSAM_ERR_UNSUPPORTED, "upscale2x2: %d channels (32 or 64)", C);
    LSAM_REQUIRE((lnw == nullptr) == (lnb == nullptr), LSAM_ERR_BAD_ARG, "upscale2x2: LayerNorm needs weight and bias");
    LSAM_REQUIRE(aligned16(g) && (reinterpret_cast<uintptr_t>(out) & 7) == 0 && (!skip || (reinterpret_cast<uintptr_t>(skip) & 7) == 0),
                 LSAM_ERR_BAD_ARG, "upscale2x2: alignment");
    const long lanes = (long)B * 4 * H * W * (C / 4);
    const dim3 grid((unsigned)((lanes + 255) / 256)), blk(256);
    const T* sk = static_cast<const T*>(skip);
    T* o = static_cast<T*>(out);
    if (C == 64 && lnw) hipLaunchKernelGGL((upscale2x2_kernel<T, 64, true>), grid, blk, 0, s, g, sk, lnw, lnb, eps, o, B, H, W);
    else if (C == 64) hipLaunchKernelGGL((upscale2x2_kernel<T, 64, false>), grid, blk, 0, s, g, sk, lnw, lnb, eps, o, B, H, W);
    else if (lnw) hipLaunchKernelGGL((upscale2x2_kernel<T, 32, true>), grid, blk, 0, s, g, sk, lnw, lnb, eps, o, B, H, W);
    else hipLaunchKernelGGL((upscale2x2_kernel<T, 32, false>), grid, blk, 0, s, g, sk, lnw, lnb, eps, o, B, H, W);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
template int launch_upscale2x2<float>(const float*, const void*, const float*, const float*, float, void*, int, int, int, int, hipStream_t);
template int launch_upscale2x2<bf16>(const float*, const void*, const float*, const float*, float, void*, int, int, int, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// logits fp32 [B][HW][ldl] (column j = mask token j) -> masks fp32 [B][nm][HW] (NCHW) of tokens first .. first + nm - 1,
// overwritten with no_obj where the frame's object score is <= 0 (sam2_base.py:359-368).
// ---------------------------------------------------------------------------------------------
__global__ void mask_gate_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ obj, float* __restrict__ out,
                                 int B, int HW, int first, int nm, float no_obj) {
    const long total = (long)B * nm * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW), j = (int)((i / HW) % nm), b = (int)(i / ((long)HW * nm));
        const float v = logits[((size_t)b * HW + p) * ldl + first + j];
        out[i] = (!obj || obj[b] > 0.f) ? v : no_obj;
    }
}
int launch_mask_gate(const float* logits, int ldl, const float* obj, float* out, int B, int HW, int first, int nm, float no_obj,
                     hipStream_t s) {
    LSAM_REQUIRE(logits && out && B > 0 && HW > 0 && nm > 0 && first >= 0 && first + nm <= ldl, LSAM_ERR_BAD_ARG,
                 "mask_gate: null/empty or columns out of range");
    const long total = (long)B * nm * HW;
    hipLaunchKernelGGL(mask_gate_kernel, dim3((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0, s,
                       logits, ldl, obj, out, B, HW, first, nm, no_obj);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// ---------------------------------------------------------------------------------------------
// Bilinear upsample, align_corners = False (PyTorch's upsample_bilinear2d: source coordinate
// max(0, (dst + 0.5) / scale - 0.5), neighbour index clamped to the last row / column).
// ---------------------------------------------------------------------------------------------
__global__ void bilinear_up_kernel(const float* __restrict__ in, float* __restrict__ out, int BC, int H, int W, int S) {
    const int Ho = H * S, Wo = W * S;
    const long total = (long)BC * Ho * (Wo / 4);
    const float inv = 1.0f / (float)S;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x4 = (int)(i % (Wo / 4)) * 4, y = (int)((i / (Wo / 4)) % Ho), c = (int)(i / ((long)(Wo / 4) * Ho));
        const float sy = fmaxf(0.f, ((float)y + 0.5f) * inv - 0.5f);
        const int y0 = (int)sy, y1 = min(y0 + 1, H - 1);
        const float wy = sy - (float)y0;
        const float* r0 = in + ((size_t)c * H + y0) * W;
        const float* r1 = in + ((size_t)c * H + y1) * W;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sx = fmaxf(0.f, ((float)(x4 + r) + 0.5f) * inv - 0.5f);
            const int x0 = (int)sx, x1 = min(x0 + 1, W - 1);
            const float wx = sx - (float)x0;
            const float top = r0[x0] + wx * (r0[x1] - r0[x0]);   // PyTorch: w0 * a + w1 * b with w0 = 1 - w1; same value to rounding
            const float bot = r1[x0] + wx * (r1[x1] - r1[x0]);
            o[r] = top + wy * (bot - top);
        }
        store4(out + ((size_t)c * Ho + y) * Wo + x4, o);
    }
}
int launch_bilinear_up(const float* in, float* out, int BC, int H, int W, int S, hipStream_t s) {
    LSAM_REQUIRE(in && out && BC > 0 && H > 0 && W > 0 && S >= 1, LSAM_ERR_BAD_ARG, "bilinear_up: null/empty");
    LSAM_REQUIRE((W * S) % 4 == 0 && aligned16(out), LSAM_ERR_BAD_ARG, "bilinear_up: output rows must be multiples of 4 floats");
    const long total = (long)BC * H * S * (W * S / 4);
    hipLaunchKernelGGL(bilinear_up_kernel, dim3((unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384)), dim3(256), 0, s,
                       in, out, BC, H, W, S);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

}  // namespace lsam

using namespace lsam;

extern "C" {

int lsam_add_rows(int a_dtype, const void* a, int lda, const float* b, int ldb, int b_rows, int out_dtype, void* out, int ldo, int M,
                  int C, lsam_stream_t stream) {
    return launch_add_rows(a_dtype, a, lda, b, ldb, b_rows, out_dtype, out, ldo, M, C, static_cast<hipStream_t>(stream));
}

int lsam_mha_fwd(int dtype, const void* q, int ldq, long q_batch_stride, const void* k, int ldk, long k_batch_stride, const void* v,
                 int ldv, long v_batch_stride, void* out, int ldo, long out_batch_stride, int B, int nq, int nk, int nh, int hd,
                 lsam_stream_t stream) {
    LSAM_REQUIRE(hd > 0, LSAM_ERR_BAD_ARG, "mha: head_dim");
    const MhaArgs a{q, k, v, out, ldq, ldk, ldv, ldo, q_batch_stride, k_batch_stride, v_batch_stride, out_batch_stride,
                    B, nq, nk, nh, 1.0f / sqrtf((float)hd)};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == LSAM_F32) return launch_mha<float>(a, hd, s);
    if (dtype == LSAM_BF16) return launch_mha<bf16>(a, hd, s);
    set_error("unknown dtype %d", dtype);
    return LSAM_ERR_BAD_ARG;
}

int lsam_pe_encode(const float* gauss, const float* coords01, float* out, int M, int F, lsam_stream_t stream) {
    return launch_pe_encode(gauss, coords01, out, M, F, static_cast<hipStream_t>(stream));
}

int lsam_upscale2x2_fwd(int dtype, const float* g, const void* skip, const float* ln_w, const float* ln_b, float eps, void* out, int B,
                        int H, int W, int C, lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == LSAM_F32) return launch_upscale2x2<float>(g, skip, ln_w, ln_b, eps, out, B, H, W, C, s);
    if (dtype == LSAM_BF16) return launch_upscale2x2<bf16>(g, skip, ln_w, ln_b, eps, out, B, H, W, C, s);
    set_error("unknown dtype %d", dtype);
    return LSAM_ERR_BAD_ARG;
}

int lsam_mask_gate(const float* logits, int ldl, const float* obj_score, float* out, int B, int HW, int first, int n_masks,
                   float no_obj_score, lsam_stream_t stream) {
    return launch_mask_gate(logits, ldl, obj_score, out, B, HW, first, n_masks, no_obj_score, static_cast<hipStream_t>(stream));
}

int lsam_bilinear_upsample(const float* in, float* out, int BC, int H, int W, int scale, lsam_stream_t stream) {
    return launch_bilinear_up(in, out, BC, H, W, scale, static_cast<hipStream_t>(stream));
}

}  // extern "C"
