// Input-gradient kernels of the SAM heads (SAM2Base._forward_sam_heads, sam2_base.py:257-413) and the mask losses
// (training/loss_fns.py:19-129), for the pieces that are not GEMMs or LayerNorms (those reuse lsam_linear with packed
// transposed weights and lsam_layernorm_bwd).  The heads' weights are frozen (train_lora_sam.py:116-117): as in the trunk
// there are no weight-gradient GEMMs; the decoder-side LoRA factors (lora_qkv.py:11-67) get theirs from lsam_outer_reduce.
// Everything here is small next to the encoder (the heads are ~1 % of the step's FLOPs): the kernels are written for
// determinism (fixed summation orders, no atomics) and clarity, fp32 arithmetic throughout.
#include "kernels.hpp"

namespace lsam {

// ---------------------------------------------------------------------------------------------
// dx = dy * act'(.) from the activation's OUTPUT y: kind 0 ReLU (y > 0), 1 sigmoid (y (1 - y)).
// ---------------------------------------------------------------------------------------------
template <typename TY>
__global__ void act_bwd_kernel(int kind, const float* __restrict__ dy, const TY* __restrict__ y, float* __restrict__ dx, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 d = load4(dy + 4 * i), v = load4(y + 4 * i);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = kind == 0 ? (v[r] > 0.f ? d[r] : 0.f) : d[r] * v[r] * (1.0f - v[r]);
        store4(dx + 4 * i, o);
    }
}
template <typename TY> int launch_act_bwd(int kind, const float* dy, const void* y, float* dx, long n, hipStream_t s) {
    LSAM_REQUIRE(dy && y && dx && n > 0 && n % 4 == 0 && (kind == 0 || kind == 1), LSAM_ERR_BAD_ARG, "act_bwd: bad argument");
    const long n4 = n / 4;
    hipLaunchKernelGGL(act_bwd_kernel<TY>, dim3((unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096)), dim3(256), 0, s, kind,
                       dy, static_cast<const TY*>(y), dx, n4);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
template int launch_act_bwd<float>(int, const float*, const void*, float*, long, hipStream_t);
template int launch_act_bwd<bf16>(int, const float*, const void*, float*, long, hipStream_t);

// ---------------------------------------------------------------------------------------------
// Multi-head attention backward on contiguous [B][n][nh * HD] operands (sam/transformer.py:233-265).  One workgroup per
// (batch, head).  The softmax statistics are recomputed; delta_q = dO_q . O_q comes from the saved forward output.
//   few queries (nq <= 16; token -> image attention and the 8 x 8 self-attention): a thread owns one KEY of a chunk of
//     256 -- it writes that key's dK and dV rows itself -- and dQ[q][d] is summed over the chunk's keys from LDS by
//     (q, d)-threads, chunk after chunk, in a fixed order.
//   few keys (nk <= 16; image -> token attention): the mirror image -- a thread owns one QUERY of a chunk, writes its dQ
//     row, and dK / dV [k][d] are summed over the chunk's queries by (k, d)-threads.
// ---------------------------------------------------------------------------------------------
struct MhaBwdArgs {
    const void *q, *k, *v, *o;   // T
    const float* dout;
    float *dq, *dk, *dv;
    int B, nq, nk, nh;
    float scale;
};
constexpr int MB_FEW = 16, MB_CH = 256, MB_CHK = 128;   // (static LDS stays under 64 KiB)

template <typename T, int HD>
__global__ __launch_bounds__(256) void mha_bwd_fewq_kernel(const MhaBwdArgs p) {
    __shared__ float Qs[MB_FEW][HD], dOs[MB_FEW][HD], lse[MB_FEW], delta[MB_FEW];
    __shared__ float Ks[MB_CH][HD + 1], dSs[MB_FEW][MB_CH];
    __shared__ float red_m[4][MB_FEW], red_l[4][MB_FEW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.nh, h = blockIdx.x % p.nh, C = p.nh * HD, nq = p.nq, nk = p.nk;
    const T* Q = static_cast<const T*>(p.q) + (size_t)b * nq * C + h * HD;
    const T* K = static_cast<const T*>(p.k) + (size_t)b * nk * C + h * HD;
    const T* V = static_cast<const T*>(p.v) + (size_t)b * nk * C + h * HD;
    const T* O = static_cast<const T*>(p.o) + (size_t)b * nq * C + h * HD;
    const float* dO = p.dout + (size_t)b * nq * C + h * HD;
    for (int i = tid; i < nq * HD; i += 256) {
        const int q = i / HD, d = i % HD;
        Qs[q][d] = to_f(Q[(size_t)q * C + d]);
        dOs[q][d] = dO[(size_t)q * C + d];
    }
    __syncthreads();
    if (tid < nq) {
        float s = 0.f;
        for (int d = 0; d < HD; ++d) s += dOs[tid][d] * to_f(O[(size_t)tid * C + d]);
        delta[tid] = s;
    }
    // ---- log-sum-exp per query: every thread over its keys, merged across the workgroup -------------------------------
    float m[MB_FEW], l[MB_FEW];
#pragma unroll
    for (int q = 0; q < MB_FEW; ++q) { m[q] = -1.0e30f; l[q] = 0.f; }
    for (int key = tid; key < nk; key += 256) {
        float kr[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) kr[d] = to_f(K[(size_t)key * C + d]);
#pragma unroll
        for (int q = 0; q < MB_FEW; ++q)
            if (q < nq) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) s += Qs[q][d] * kr[d];
                s *= p.scale;
                const float mn = fmaxf(m[q], s);
                l[q] = l[q] * __expf(m[q] - mn) + __expf(s - mn);
                m[q] = mn;
            }
    }
#pragma unroll
    for (int q = 0; q < MB_FEW; ++q) {
        float mm = m[q], ll = l[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float mo = __shfl_xor(mm, off, 64), lo = __shfl_xor(ll, off, 64);
            const float mn = fmaxf(mm, mo);
            ll = ll * __expf(mm - mn) + lo * __expf(mo - mn);
            mm = mn;
        }
        if (lane == 0) { red_m[wave][q] = mm; red_l[wave][q] = ll; }
    }
    __syncthreads();
    if (tid < nq) {
        float mm = red_m[0][tid], ll = red_l[0][tid];
        for (int w = 1; w < 4; ++w) {
            const float mn = fmaxf(mm, red_m[w][tid]);
            ll = ll * __expf(mm - mn) + red_l[w][tid] * __expf(red_m[w][tid] - mn);
            mm = mn;
        }
        lse[tid] = mm + __logf(ll);
    }
    __syncthreads();
    // ---- chunks of 256 keys ------------------------------------------------------------------------------------------------
    constexpr int NP = (MB_FEW * HD + 255) / 256;   // (q, d) pairs per thread
    float dqa[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) dqa[i] = 0.f;
    for (int c0 = 0; c0 < nk; c0 += MB_CH) {
        const int key = c0 + tid;
        const bool ok = key < nk;
        float kr[HD], vr[HD], dkr[HD], dvr[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            kr[d] = ok ? to_f(K[(size_t)key * C + d]) : 0.f;
            vr[d] = ok ? to_f(V[(size_t)key * C + d]) : 0.f;
            dkr[d] = 0.f;
            dvr[d] = 0.f;
            Ks[tid][d] = kr[d];
        }
#pragma unroll
        for (int q = 0; q < MB_FEW; ++q) {
            float ds = 0.f;
            if (q < nq && ok) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) { s += Qs[q][d] * kr[d]; dp += dOs[q][d] * vr[d]; }
                const float pr = __expf(s * p.scale - lse[q]);
                ds = pr * (dp - delta[q]) * p.scale;
#pragma unroll
                for (int d = 0; d < HD; ++d) { dvr[d] += pr * dOs[q][d]; dkr[d] += ds * Qs[q][d]; }
            }
            dSs[q][tid] = ds;
        }
        if (ok) {
            float* dk = p.dk + ((size_t)b * nk + key) * C + h * HD;
            float* dv = p.dv + ((size_t)b * nk + key) * C + h * HD;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                store4(dk + d, f32x4{dkr[d], dkr[d + 1], dkr[d + 2], dkr[d + 3]});
                store4(dv + d, f32x4{dvr[d], dvr[d + 1], dvr[d + 2], dvr[d + 3]});
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int pr = tid + 256 * i, q = pr / HD, d = pr % HD;
            if (q < nq) {
                float s = 0.f;
                for (int t = 0; t < MB_CH; ++t) s += dSs[q][t] * Ks[t][d];
                dqa[i] += s;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int pr = tid + 256 * i, q = pr / HD, d = pr % HD;
        if (q < nq) p.dq[((size_t)b * nq + q) * C + h * HD + d] = dqa[i];
    }
}

template <typename T, int HD>
__global__ __launch_bounds__(MB_CHK) void mha_bwd_fewk_kernel(const MhaBwdArgs p) {
    __shared__ float Ksh[MB_FEW][HD], Vsh[MB_FEW][HD];
    __shared__ float Qs[MB_CHK][HD + 1], dOs[MB_CHK][HD + 1], Ps[MB_FEW][MB_CHK], dSs[MB_FEW][MB_CHK];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / p.nh, h = blockIdx.x % p.nh, C = p.nh * HD, nq = p.nq, nk = p.nk;
    const T* Q = static_cast<const T*>(p.q) + (size_t)b * nq * C + h * HD;
    const T* K = static_cast<const T*>(p.k) + (size_t)b * nk * C + h * HD;
    const T* V = static_cast<const T*>(p.v) + (size_t)b * nk * C + h * HD;
    const T* O = static_cast<const T*>(p.o) + (size_t)b * nq * C + h * HD;
    const float* dO = p.dout + (size_t)b * nq * C + h * HD;
    for (int i = tid; i < nk * HD; i += MB_CHK) {
        Ksh[i / HD][i % HD] = to_f(K[(size_t)(i / HD) * C + i % HD]);
        Vsh[i / HD][i % HD] = to_f(V[(size_t)(i / HD) * C + i % HD]);
    }
    __syncthreads();
    constexpr int NP = (MB_FEW * HD + MB_CHK - 1) / MB_CHK;
    float dka[NP], dva[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) { dka[i] = 0.f; dva[i] = 0.f; }
    for (int c0 = 0; c0 < nq; c0 += MB_CHK) {
        const int qi = c0 + tid;
        const bool ok = qi < nq;
        float qr[HD], dor[HD], dqr[HD];
        float dl = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            qr[d] = ok ? to_f(Q[(size_t)qi * C + d]) : 0.f;
            dor[d] = ok ? dO[(size_t)qi * C + d] : 0.f;
            dl += ok ? dor[d] * to_f(O[(size_t)qi * C + d]) : 0.f;
            dqr[d] = 0.f;
            Qs[tid][d] = qr[d];
            dOs[tid][d] = dor[d];
        }
        float s[MB_FEW], mx = -1.0e30f;
#pragma unroll
        for (int k = 0; k < MB_FEW; ++k) {
            s[k] = -1.0e30f;
            if (k < nk) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) a += qr[d] * Ksh[k][d];
                s[k] = a * p.scale;
                mx = fmaxf(mx, s[k]);
            }
        }
        float l = 0.f;
#pragma unroll
        for (int k = 0; k < MB_FEW; ++k) {
            s[k] = k < nk ? __expf(s[k] - mx) : 0.f;
            l += s[k];
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int k = 0; k < MB_FEW; ++k) {
            float pr = 0.f, ds = 0.f;
            if (k < nk && ok) {
                pr = s[k] * inv;
                float dp = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) dp += dor[d] * Vsh[k][d];
                ds = pr * (dp - dl) * p.scale;
#pragma unroll
                for (int d = 0; d < HD; ++d) dqr[d] += ds * Ksh[k][d];
            }
            Ps[k][tid] = pr;
            dSs[k][tid] = ds;
        }
        if (ok) {
            float* dq = p.dq + ((size_t)b * nq + qi) * C + h * HD;
#pragma unroll
            for (int d = 0; d < HD; d += 4) store4(dq + d, f32x4{dqr[d], dqr[d + 1], dqr[d + 2], dqr[d + 3]});
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int pr = tid + MB_CHK * i, k = pr / HD, d = pr % HD;
            if (k < nk) {
                float a = 0.f, c = 0.f;
                for (int t = 0; t < MB_CHK; ++t) { a += dSs[k][t] * Qs[t][d]; c += Ps[k][t] * dOs[t][d]; }
                dka[i] += a;
                dva[i] += c;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int pr = tid + MB_CHK * i, k = pr / HD, d = pr % HD;
        if (k < nk) {
            p.dk[((size_t)b * nk + k) * C + h * HD + d] = dka[i];
            p.dv[((size_t)b * nk + k) * C + h * HD + d] = dva[i];
        }
    }
}

template <typename T> int launch_mha_bwd(const MhaBwdArgs& a, int hd, hipStream_t s) {
    LSAM_REQUIRE(a.q && a.k && a.v && a.o && a.dout && a.dq && a.dk && a.dv && a.B > 0 && a.nq > 0 && a.nk > 0 && a.nh > 0,
                 LSAM_ERR_BAD_ARG, "mha_bwd: null/empty argument");
    LSAM_REQUIRE(hd == 16 || hd == 32, LSAM_ERR_UNSUPPORTED, "mha_bwd: head_dim %d (16 or 32: the SAM two-way transformer's)", hd);
    LSAM_REQUIRE(a.nq <= MB_FEW || a.nk <= MB_FEW, LSAM_ERR_UNSUPPORTED,
                 "mha_bwd: %d queries x %d keys (one side must be <= %d)", a.nq, a.nk, MB_FEW);
    LSAM_REQUIRE(aligned16(a.dout) && aligned16(a.dq) && aligned16(a.dk) && aligned16(a.dv), LSAM_ERR_BAD_ARG, "mha_bwd: alignment");
    const dim3 grid(a.B * a.nh), blk(a.nq <= MB_FEW ? 256 : MB_CHK);
    if (a.nq <= MB_FEW) {
        if (hd == 16) hipLaunchKernelGGL((mha_bwd_fewq_kernel<T, 16>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((mha_bwd_fewq_kernel<T, 32>), grid, blk, 0, s, a);
    } else {
        if (hd == 16) hipLaunchKernelGGL((mha_bwd_fewk_kernel<T, 16>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((mha_bwd_fewk_kernel<T, 32>), grid, blk, 0, s, a);
    }
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward of lsam_upscale2x2_fwd with respect to g (the ConvTranspose GEMM's output): the pre-activation is recomputed
// from g (+ skip), d_pre = LN'(GELU'(.) d_out) lands on the (dy, dx, c) column block of the input token it came from.
// ---------------------------------------------------------------------------------------------
template <typename T, int C, bool LN>
__global__ __launch_bounds__(256) void upscale2x2_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ g,
                                                             const T* __restrict__ skip, const float* __restrict__ lnw,
                                                             const float* __restrict__ lnb, float eps, float* __restrict__ dg, int B,
                                                             int H, int W) {
    constexpr int LPP = C / 4;
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
    f32x4 d = load4(dout + (size_t)pc * C + 4 * li);
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
        f32x4 xh, a;
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xh[r] = (v[r] - mean) * rstd;
            a[r] = w[r] * d[r] * gelu_grad_t<false>(xh[r] * w[r] + bb[r]);
            m1 += a[r];
            m2 += a[r] * xh[r];
        }
#pragma unroll
        for (int off = LPP / 2; off > 0; off >>= 1) { m1 += __shfl_xor(m1, off, 64); m2 += __shfl_xor(m2, off, 64); }
        m1 /= (float)C;
        m2 /= (float)C;
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = rstd * (a[r] - m1 - xh[r] * m2);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] *= gelu_grad_t<false>(v[r]);
    }
    if (ok) store4(dg + tok * (4 * C) + sub * C + 4 * li, d);
}
template <typename T>
int launch_upscale2x2_bwd(const float* dout, const float* g, const void* skip, const float* lnw, const float* lnb, float eps, float* dg,
                          int B, int H, int W, int C, hipStream_t s) {
    LSAM_REQUIRE(dout && g && dg && B > 0 && H > 0 && W > 0, LSAM_ERR_BAD_ARG, "upscale2x2_bwd: null/empty");
    LSAM_REQUIRE(C == 32 || C == 64, LSAM_ERR_UNSUPPORTED, "upscale2x2_bwd: %d channels (32 or 64)", C);
    LSAM_REQUIRE((lnw == nullptr) == (lnb == nullptr), LSAM_ERR_BAD_ARG, "upscale2x2_bwd: LayerNorm weight and bias go together");
    const long threads = (long)B * 4 * H * W * (C / 4);
    const dim3 grid((unsigned)((threads + 255) / 256)), blk(256);
    const T* sk = static_cast<const T*>(skip);
    if (C == 64 && lnw) hipLaunchKernelGGL((upscale2x2_bwd_kernel<T, 64, true>), grid, blk, 0, s, dout, g, sk, lnw, lnb, eps, dg, B, H, W);
    else if (C == 64) hipLaunchKernelGGL((upscale2x2_bwd_kernel<T, 64, false>), grid, blk, 0, s, dout, g, sk, lnw, lnb, eps, dg, B, H, W);
    else if (lnw) hipLaunchKernelGGL((upscale2x2_bwd_kernel<T, 32, true>), grid, blk, 0, s, dout, g, sk, lnw, lnb, eps, dg, B, H, W);
    else hipLaunchKernelGGL((upscale2x2_bwd_kernel<T, 32, false>), grid, blk, 0, s, dout, g, sk, lnw, lnb, eps, dg, B, H, W);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// ---------------------------------------------------------------------------------------------
// Adjoint of lsam_mask_gate: d_logits fp32 [B][HW][ldl], zero outside the selected columns and for gated frames.
// ---------------------------------------------------------------------------------------------
__global__ void mask_gate_bwd_kernel(const float* __restrict__ dlow, const float* __restrict__ obj, float* __restrict__ dlogits, int ldl,
                                     int B, int HW, int first, int nm) {
    const long total = (long)B * HW * ldl;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % ldl), pix = (int)((i / ldl) % HW), b = (int)(i / ((long)ldl * HW));
        float v = 0.f;
        if (j >= first && j < first + nm && (!obj || obj[b] > 0.f)) v = dlow[((size_t)b * nm + (j - first)) * HW + pix];
        dlogits[i] = v;
    }
}
int launch_mask_gate_bwd(const float* dlow, const float* obj, float* dlogits, int ldl, int B, int HW, int first, int nm, hipStream_t s) {
    LSAM_REQUIRE(dlow && dlogits && B > 0 && HW > 0 && nm > 0 && first >= 0 && first + nm <= ldl, LSAM_ERR_BAD_ARG,
                 "mask_gate_bwd: null/empty or columns out of range");
    const long total = (long)B * HW * ldl;
    hipLaunchKernelGGL(mask_gate_bwd_kernel, dim3((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0, s,
                       dlow, obj, dlogits, ldl, B, HW, first, nm);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// ---------------------------------------------------------------------------------------------
// Adjoint of lsam_bilinear_upsample (align_corners = False): every input pixel gathers, with the forward's own weights,
// from the output pixels whose two-point stencil touches it (a fixed order: no atomics).
// ---------------------------------------------------------------------------------------------
__global__ void bilinear_up_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int BC, int H, int W, int S) {
    const int Ho = H * S, Wo = W * S;
    const long total = (long)BC * H * W;
    const float inv = 1.0f / (float)S;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)(i / ((long)W * H));
        const float* src = dout + (size_t)c * Ho * Wo;
        float acc = 0.f;
        const int Y0 = max(0, (y - 1) * S), Y1 = min(Ho - 1, (y + 1) * S + S - 1);
        const int X0 = max(0, (x - 1) * S), X1 = min(Wo - 1, (x + 1) * S + S - 1);
        for (int Y = Y0; Y <= Y1; ++Y) {
            const float sy = fmaxf(0.f, ((float)Y + 0.5f) * inv - 0.5f);
            const int y0 = (int)sy, y1 = min(y0 + 1, H - 1);
            const float wy1 = sy - (float)y0;
            const float wy = (y0 == y ? 1.0f - wy1 : 0.f) + (y1 == y ? wy1 : 0.f);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int X = X0; X <= X1; ++X) {
                const float sx = fmaxf(0.f, ((float)X + 0.5f) * inv - 0.5f);
                const int x0 = (int)sx, x1 = min(x0 + 1, W - 1);
                const float wx1 = sx - (float)x0;
                const float wx = (x0 == x ? 1.0f - wx1 : 0.f) + (x1 == x ? wx1 : 0.f);
                if (wx != 0.f) row += wx * src[(size_t)Y * Wo + X];
            }
            acc += wy * row;
        }
        din[i] = acc;
    }
}
int launch_bilinear_up_bwd(const float* dout, float* din, int BC, int H, int W, int S, hipStream_t s) {
    LSAM_REQUIRE(dout && din && BC > 0 && H > 0 && W > 0 && S >= 1, LSAM_ERR_BAD_ARG, "bilinear_up_bwd: null/empty");
    const long total = (long)BC * H * W;
    hipLaunchKernelGGL(bilinear_up_bwd_kernel, dim3((unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384)), dim3(256), 0,
                       s, dout, din, BC, H, W, S);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// ---------------------------------------------------------------------------------------------
// out[b][l][c] = sum_m G[b][m][l] U[b][m][c]: the hyper-network side of the mask product's backward (d hyper_in =
// d_masks . upscaled^T, mask_decoder.py:238-239): G fp32 [B][M][L] (L <= 16 mask tokens), U T [B][M][C] (C <= 64), a
// reduction over the M = 65536 pixels of every frame.  BO_SPLIT workgroups per frame sum row slices (tiles of 128 rows
// staged in LDS, one (l, c) pair per thread), a second kernel adds the slices in a fixed order.
// ---------------------------------------------------------------------------------------------
constexpr int BO_SPLIT = 32, BO_ROWS = 128;
template <typename T>
__global__ __launch_bounds__(256) void batched_outer_kernel(const float* __restrict__ G, const T* __restrict__ U, float* __restrict__ part,
                                                            int M, int L, int C) {
    __shared__ float Gs[BO_ROWS][16], Us[BO_ROWS][64];
    const int b = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
    const int per = (M + BO_SPLIT - 1) / BO_SPLIT, m0 = sl * per, m1 = min(M, m0 + per);
    const float* Gb = G + (size_t)b * M * L;
    const T* Ub = U + (size_t)b * M * C;
    const int np = L * C;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = m0; r0 < m1; r0 += BO_ROWS) {
        const int rows = min(BO_ROWS, m1 - r0);
        for (int i = tid; i < BO_ROWS * L; i += 256) Gs[i / L][i % L] = (i / L) < rows ? Gb[(size_t)(r0 + i / L) * L + i % L] : 0.f;
        for (int i = tid; i < BO_ROWS * C; i += 256) Us[i / C][i % C] = (i / C) < rows ? to_f(Ub[(size_t)(r0 + i / C) * C + i % C]) : 0.f;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pr = tid + 256 * k;
            if (pr < np) {
                const int l = pr / C, c = pr % C;
                float a = 0.f;
                for (int r = 0; r < BO_ROWS; ++r) a += Gs[r][l] * Us[r][c];
                acc[k] += a;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int pr = tid + 256 * k;
        if (pr < np) part[((size_t)b * BO_SPLIT + sl) * np + pr] = acc[k];
    }
}
__global__ void batched_outer_final_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int np) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * np) return;
    const int b = i / np, pr = i % np;
    float s = 0.f;
    for (int sl = 0; sl < BO_SPLIT; ++sl) s += part[((size_t)b * BO_SPLIT + sl) * np + pr];
    out[i] = s;
}
size_t batched_outer_ws_floats(int B, int L, int C) { return (size_t)B * BO_SPLIT * L * C; }
template <typename T>
int launch_batched_outer(const float* G, const void* U, float* out, float* ws, int B, int M, int L, int C, hipStream_t s) {
    LSAM_REQUIRE(G && U && out && ws && B > 0 && M > 0, LSAM_ERR_BAD_ARG, "batched_outer: null/empty");
    LSAM_REQUIRE(L >= 1 && L <= 16 && C >= 1 && C <= 64 && L * C <= 1024, LSAM_ERR_UNSUPPORTED, "batched_outer: L <= 16, C <= 64 (L=%d C=%d)", L, C);
    hipLaunchKernelGGL(batched_outer_kernel<T>, dim3(B, BO_SPLIT), dim3(256), 0, s, G, static_cast<const T*>(U), ws, M, L, C);
    LSAM_LAUNCH_CHECK();
    hipLaunchKernelGGL(batched_outer_final_kernel, dim3((B * L * C + 255) / 256), dim3(256), 0, s, ws, out, B, L * C);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// ---------------------------------------------------------------------------------------------
// Mask losses (training/loss_fns.py): the per-mask sums every term is built from, in one pass over the logits:
//   sums[n][m] = { sum focal(x, t),  sum sigmoid(x) t,  sum sigmoid(x),  sum t,  |x > 0 and t > 0|,  |x > 0 or t > 0| }
// with focal = alpha_t * bce(x, t) * (1 - p_t)^gamma (sigmoid_focal_loss :58-98; alpha < 0: no alpha weighting), the dice
// numerator / denominator pieces (:19-55) and the areas of iou_loss (:101-129).  x fp32 [N][M][HW], t fp32 [N][HW] (the
// targets are broadcast over the M candidates, :240).  The backward takes d loss / d sums[., 0..2] (the areas carry no
// gradient) and forms dx = c0 focal'(x) + (c1 t + c2) p (1 - p).  One workgroup per (n, m), fixed-order tree reduction.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void focal_terms(float x, float t, float alpha, float gamma, float& loss, float& dldx, float& p) {
    p = 1.0f / (1.0f + __expf(-x));
    const float ce = fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));   // binary_cross_entropy_with_logits
    const float pt = p * t + (1.0f - p) * (1.0f - t);
    const float om = 1.0f - pt;
    const float mod = gamma == 2.0f ? om * om : (gamma == 0.0f ? 1.0f : __powf(fmaxf(om, 1e-30f), gamma));
    const float at = alpha >= 0.f ? alpha * t + (1.0f - alpha) * (1.0f - t) : 1.0f;
    loss = at * ce * mod;
    // d ce / dx = p - t ;  d pt / dx = (2 t - 1) p (1 - p) ;  d mod / dx = -gamma om^(gamma - 1) d pt / dx
    const float dpt = (2.0f * t - 1.0f) * p * (1.0f - p);
    const float dmod = gamma == 2.0f ? -2.0f * om * dpt : (gamma == 0.0f ? 0.f : -gamma * __powf(fmaxf(om, 1e-30f), gamma - 1.0f) * dpt);
    dldx = at * ((p - t) * mod + ce * dmod);
}
constexpr int ML_SPLIT = 64;   // workgroups per candidate mask (a 1024^2 mask is 1 M pixels: one workgroup per mask took 3 ms)
__global__ __launch_bounds__(256) void mask_loss_sums_kernel(const float* __restrict__ x, const float* __restrict__ t, float* __restrict__ part,
                                                             int M, int HW, float alpha, float gamma) {
    __shared__ float red[6][256];
    const int nm = blockIdx.x, n = nm / M, sl = blockIdx.y;
    const float* xp = x + (size_t)nm * HW;
    const float* tp = t + (size_t)n * HW;
    const int per = (HW + ML_SPLIT - 1) / ML_SPLIT, i0 = sl * per, i1 = min(HW, i0 + per);
    float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const float xv = xp[i], tv = tp[i];
        float l, d, p;
        focal_terms(xv, tv, alpha, gamma, l, d, p);
        a[0] += l;
        a[1] += p * tv;
        a[2] += p;
        a[3] += tv;
        a[4] += (xv > 0.f && tv > 0.f) ? 1.f : 0.f;
        a[5] += (xv > 0.f || tv > 0.f) ? 1.f : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) red[j][threadIdx.x] = a[j];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int j = 0; j < 6; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) part[((size_t)nm * ML_SPLIT + sl) * 6 + threadIdx.x] = red[threadIdx.x][0];
}
__global__ void mask_loss_sums_final_kernel(const float* __restrict__ part, float* __restrict__ sums, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (mask, quantity)
    if (i >= total) return;
    const int nm = i / 6, j = i % 6;
    float s = 0.f;
    for (int sl = 0; sl < ML_SPLIT; ++sl) s += part[((size_t)nm * ML_SPLIT + sl) * 6 + j];   // fixed order
    sums[i] = s;
}
__global__ void mask_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ coef,
                                     float* __restrict__ dx, int M, int HW, float alpha, float gamma) {
    const int nm = blockIdx.y, n = nm / M;
    const float c0 = coef[nm * 3], c1 = coef[nm * 3 + 1], c2 = coef[nm * 3 + 2];
    const float* xp = x + (size_t)nm * HW;
    const float* tp = t + (size_t)n * HW;
    float* dp = dx + (size_t)nm * HW;
    const bool live = c0 != 0.f || c1 != 0.f || c2 != 0.f;   // the candidates the loss did not select
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (live) {
            float l, d, p;
            focal_terms(xp[i], tp[i], alpha, gamma, l, d, p);
            v = c0 * d + (c1 * tp[i] + c2) * p * (1.0f - p);
        }
        dp[i] = v;
    }
}
size_t mask_loss_ws_floats(int N, int M) { return (size_t)N * M * ML_SPLIT * 6; }
int launch_mask_loss_sums(const float* x, const float* t, float* sums, float* ws, int N, int M, int HW, float alpha, float gamma,
                          hipStream_t s) {
    LSAM_REQUIRE(x && t && sums && ws && N > 0 && M > 0 && HW > 0, LSAM_ERR_BAD_ARG, "mask_loss_sums: null/empty");
    hipLaunchKernelGGL(mask_loss_sums_kernel, dim3(N * M, ML_SPLIT), dim3(256), 0, s, x, t, ws, M, HW, alpha, gamma);
    LSAM_LAUNCH_CHECK();
    hipLaunchKernelGGL(mask_loss_sums_final_kernel, dim3((N * M * 6 + 255) / 256), dim3(256), 0, s, ws, sums, N * M * 6);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
int launch_mask_loss_bwd(const float* x, const float* t, const float* coef, float* dx, int N, int M, int HW, float alpha, float gamma,
                         hipStream_t s) {
    LSAM_REQUIRE(x && t && coef && dx && N > 0 && M > 0 && HW > 0, LSAM_ERR_BAD_ARG, "mask_loss_bwd: null/empty");
    hipLaunchKernelGGL(mask_loss_bwd_kernel, dim3((HW + 255) / 256 < 1024 ? (HW + 255) / 256 : 1024, N * M), dim3(256), 0, s, x, t, coef, dx,
                       M, HW, alpha, gamma);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

}  // namespace lsam

using namespace lsam;

extern "C" {

int lsam_act_bwd(int kind, const float* dy, int y_dtype, const void* y, float* dx, long n, lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (y_dtype == LSAM_F32) return launch_act_bwd<float>(kind, dy, y, dx, n, s);
    if (y_dtype == LSAM_BF16) return launch_act_bwd<bf16>(kind, dy, y, dx, n, s);
    set_error("act_bwd: bad dtype %d", y_dtype);
    return LSAM_ERR_BAD_ARG;
}
int lsam_mha_bwd(int dtype, const void* q, const void* k, const void* v, const void* out, const float* dout, float* dq, float* dk,
                 float* dv, int B, int nq, int nk, int nh, int hd, lsam_stream_t stream) {
    MhaBwdArgs a{q, k, v, out, dout, dq, dk, dv, B, nq, nk, nh, 1.0f / sqrtf((float)hd)};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == LSAM_F32) return launch_mha_bwd<float>(a, hd, s);
    if (dtype == LSAM_BF16) return launch_mha_bwd<bf16>(a, hd, s);
    set_error("mha_bwd: bad dtype %d", dtype);
    return LSAM_ERR_BAD_ARG;
}
int lsam_upscale2x2_bwd(int dtype, const float* dout, const float* g, const void* skip, const float* ln_w, const float* ln_b, float eps,
                        float* dg, int B, int H, int W, int C, lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == LSAM_F32) return launch_upscale2x2_bwd<float>(dout, g, skip, ln_w, ln_b, eps, dg, B, H, W, C, s);
    if (dtype == LSAM_BF16) return launch_upscale2x2_bwd<bf16>(dout, g, skip, ln_w, ln_b, eps, dg, B, H, W, C, s);
    set_error("upscale2x2_bwd: bad dtype %d", dtype);
    return LSAM_ERR_BAD_ARG;
}
int lsam_mask_gate_bwd(const float* d_masks, const float* obj_score, float* d_logits, int ldl, int B, int HW, int first, int n_masks,
                       lsam_stream_t stream) {
    return launch_mask_gate_bwd(d_masks, obj_score, d_logits, ldl, B, HW, first, n_masks, static_cast<hipStream_t>(stream));
}
int lsam_bilinear_upsample_bwd(const float* dout, float* din, int BC, int H, int W, int scale, lsam_stream_t stream) {
    return launch_bilinear_up_bwd(dout, din, BC, H, W, scale, static_cast<hipStream_t>(stream));
}
size_t lsam_batched_outer_ws_floats(int B, int L, int C) { return batched_outer_ws_floats(B, L, C); }
int lsam_batched_outer(int u_dtype, const float* G, const void* U, float* out, float* workspace, int B, int M, int L, int C,
                       lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (u_dtype == LSAM_F32) return launch_batched_outer<float>(G, U, out, workspace, B, M, L, C, s);
    if (u_dtype == LSAM_BF16) return launch_batched_outer<bf16>(G, U, out, workspace, B, M, L, C, s);
    set_error("batched_outer: bad dtype %d", u_dtype);
    return LSAM_ERR_BAD_ARG;
}
size_t lsam_mask_loss_ws_floats(int N, int M) { return mask_loss_ws_floats(N, M); }
int lsam_mask_loss_sums(const float* logits, const float* targets, float* sums, float* workspace, int N, int M, int HW, float alpha,
                        float gamma, lsam_stream_t stream) {
    return launch_mask_loss_sums(logits, targets, sums, workspace, N, M, HW, alpha, gamma, static_cast<hipStream_t>(stream));
}
int lsam_mask_loss_bwd(const float* logits, const float* targets, const float* coef, float* d_logits, int N, int M, int HW, float alpha,
                       float gamma, lsam_stream_t stream) {
    return launch_mask_loss_bwd(logits, targets, coef, d_logits, N, M, HW, alpha, gamma, static_cast<hipStream_t>(stream));
}

}  // extern "C"
