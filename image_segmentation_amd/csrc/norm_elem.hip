// HBM-bound kernels of the path: LayerNorm forward/backward (one row per 32- or 64-lane group,
// float4 accesses, xor-shuffle reductions), 2x2 max-pool of the shortcut, casts, PatchEmbed
// im2col, and the LoRA dA/dB token reductions.
#include <mutex>

#include "kernels.hpp"

namespace lsam {

template <int LPR> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm forward  (nn.LayerNorm(eps=1e-6): hieradet.py:100,136,165)
// ---------------------------------------------------------------------------------------------
template <typename T, int VPT, int LPR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     int ldy, float* __restrict__ stats, int M, int C,
                                                     float eps) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, li = lane % LPR;
    const int row = (blockIdx.x * 4 + wave) * RPW + sub;
    const bool rv = row < M;
    const int nch = C >> 2;
    const float* xr = x + (size_t)(rv ? row : 0) * ldx;
    f32x4 v[VPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c = li + i * LPR;
        v[i] = (rv && c < nch) ? load4(xr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = group_sum<LPR>(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c = li + i * LPR;
        if (c < nch) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = v[i][r] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(group_sum<LPR>(q) / (float)C + eps);
    if (!rv) return;
    if (stats && li == 0) {
        stats[2 * (size_t)row] = mean;
        stats[2 * (size_t)row + 1] = rstd;
    }
    T* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c = li + i * LPR;
        if (c < nch) {
            const f32x4 gm = load4(gamma + 4 * c), bt = load4(beta + 4 * c);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v[i][r] - mean) * rstd * gm[r] + bt[r];
            store4(yr + 4 * c, o);
        }
    }
}

template <typename T, int LPR>
static int ln_fwd_dispatch(const float* x, int ldx, const float* g, const float* b, T* y, int ldy, float* st,
                           int M, int C, float eps, hipStream_t s) {
    const int vpt = cdiv(C / 4, LPR);
    const int rows_per_block = 4 * (64 / LPR);
    const dim3 grid(cdiv(M, rows_per_block)), blk(256);
#define LN_CASE(V)                                                                                  \
    case V:                                                                                         \
        hipLaunchKernelGGL((ln_fwd_kernel<T, V, LPR>), grid, blk, 0, s, x, ldx, g, b, y, ldy, st, M, C, eps); \
        break;
    switch (vpt) {
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
        default:
            set_error("layernorm: C=%d too wide", C);
            return LSAM_ERR_UNSUPPORTED;
    }
#undef LN_CASE
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

template <typename T>
int launch_ln_fwd(const float* x, int ldx, const float* gamma, const float* beta, T* y, int ldy, float* stats,
                  int M, int C, float eps, hipStream_t s) {
    LSAM_REQUIRE(x && gamma && beta && y && M > 0 && C > 0, LSAM_ERR_BAD_ARG, "layernorm: null/empty");
    LSAM_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(gamma) &&
                     aligned16(beta) && (reinterpret_cast<uintptr_t>(y) & 7) == 0,
                 LSAM_ERR_BAD_ARG, "layernorm: C, ld must be multiples of 4 and pointers aligned");
    ProfScope ps(s, (double)M * C * (4 + sizeof(T)), PROF_LN_FWD, M, C, 0);
    if (C / 4 <= 32) return ln_fwd_dispatch<T, 32>(x, ldx, gamma, beta, y, ldy, stats, M, C, eps, s);
    return ln_fwd_dispatch<T, 64>(x, ldx, gamma, beta, y, ldy, stats, M, C, eps, s);
}
template int launch_ln_fwd<float>(const float*, int, const float*, const float*, float*, int, float*, int, int, float, hipStream_t);
template int launch_ln_fwd<bf16>(const float*, int, const float*, const float*, bf16*, int, float*, int, int, float, hipStream_t);

// ---------------------------------------------------------------------------------------------
// LayerNorm input gradient (+ incoming residual gradient).  gamma/beta are frozen.
//   g = gamma*dy ; dx = rstd * (g - mean(g) - xhat * mean(g*xhat)) + dres
// ---------------------------------------------------------------------------------------------
template <typename T, int VPT, int LPR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, int lddy,
                                                     const float* __restrict__ x, int ldx,
                                                     const float* __restrict__ stats,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ dres, int lddres,
                                                     float* __restrict__ dx, int lddx, T* __restrict__ dxt,
                                                     int lddxt, int M, int C) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, li = lane % LPR;
    const int row = (blockIdx.x * 4 + wave) * RPW + sub;
    const bool rv = row < M;
    const size_t r0 = rv ? row : 0;
    const int nch = C >> 2;
    const float mean = stats[2 * r0], rstd = stats[2 * r0 + 1];
    f32x4 g[VPT], xh[VPT];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c = li + i * LPR;
        g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        xh[i] = g[i];
        if (rv && c < nch) {
            const f32x4 d = load4(dy + r0 * lddy + 4 * c), gm = load4(gamma + 4 * c), xv = load4(x + r0 * ldx + 4 * c);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                g[i][r] = d[r] * gm[r];
                xh[i][r] = (xv[r] - mean) * rstd;
                a += g[i][r];
                b += g[i][r] * xh[i][r];
            }
        }
    }
    a = group_sum<LPR>(a) / (float)C;
    b = group_sum<LPR>(b) / (float)C;
    if (!rv) return;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c = li + i * LPR;
        if (c < nch) {
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = rstd * (g[i][r] - a - xh[i][r] * b);
            if (dres) o += load4(dres + r0 * lddres + 4 * c);
            if (dx) store4(dx + r0 * lddx + 4 * c, o);
            if (dxt) store4(dxt + r0 * lddxt + 4 * c, o);
        }
    }
}

template <typename T, int LPR>
static int ln_bwd_dispatch(const float* dy, int lddy, const float* x, int ldx, const float* st, const float* g,
                           const float* dres, int lddres, float* dx, int lddx, T* dxt, int lddxt, int M, int C,
                           hipStream_t s) {
    const int vpt = cdiv(C / 4, LPR);
    const dim3 grid(cdiv(M, 4 * (64 / LPR))), blk(256);
#define LN_CASE(V)                                                                                           \
    case V:                                                                                                  \
        hipLaunchKernelGGL((ln_bwd_kernel<T, V, LPR>), grid, blk, 0, s, dy, lddy, x, ldx, st, g, dres, lddres, \
                           dx, lddx, dxt, lddxt, M, C);                                                      \
        break;
    switch (vpt) {
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
        default:
            set_error("layernorm_bwd: C=%d too wide", C);
            return LSAM_ERR_UNSUPPORTED;
    }
#undef LN_CASE
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

template <typename T>
int launch_ln_bwd(const float* dy, int lddy, const float* x, int ldx, const float* stats, const float* gamma,
                  const float* dres, int lddres, float* dx, int lddx, T* dx_t, int lddxt, int M, int C,
                  hipStream_t s) {
    LSAM_REQUIRE(dy && x && stats && gamma && (dx || dx_t) && M > 0 && C > 0, LSAM_ERR_BAD_ARG,
                 "layernorm_bwd: null/empty");
    LSAM_REQUIRE(C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && (!dres || lddres % 4 == 0) &&
                     (!dx || lddx % 4 == 0) && (!dx_t || lddxt % 4 == 0),
                 LSAM_ERR_BAD_ARG, "layernorm_bwd: C and leading dimensions must be multiples of 4");
    ProfScope ps(s, (double)M * C * (8.0 + (dres ? 4 : 0) + (dx ? 4 : 0) + (dx_t ? sizeof(T) : 0)), PROF_LN_BWD, M, C, 0);
    if (C / 4 <= 32)
        return ln_bwd_dispatch<T, 32>(dy, lddy, x, ldx, stats, gamma, dres, lddres, dx, lddx, dx_t, lddxt, M, C, s);
    return ln_bwd_dispatch<T, 64>(dy, lddy, x, ldx, stats, gamma, dres, lddres, dx, lddx, dx_t, lddxt, M, C, s);
}
template int launch_ln_bwd<float>(const float*, int, const float*, int, const float*, const float*, const float*, int, float*, int, float*, int, int, int, hipStream_t);
template int launch_ln_bwd<bf16>(const float*, int, const float*, int, const float*, const float*, const float*, int, float*, int, bf16*, int, int, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// 2x2 max-pool, fp32 NHWC (do_pool on the transition-block shortcut, hieradet.py:140)
// ---------------------------------------------------------------------------------------------
__global__ void maxpool2x2_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int H, int W,
                                  int C4) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long t = i / C4;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho), n = (int)(t / Ho);
        const float* p = x + (((size_t)n * H + 2 * ho) * W + 2 * wo) * (size_t)(C4 * 4) + 4 * c;
        const size_t rs = (size_t)W * C4 * 4;
        const f32x4 a = load4(p), b = load4(p + C4 * 4), d = load4(p + rs), e = load4(p + rs + C4 * 4);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaxf(fmaxf(a[r], b[r]), fmaxf(d[r], e[r]));
        store4(out + i * 4, o);
    }
}
int launch_maxpool2x2(const float* x, float* out, int N, int H, int W, int C, hipStream_t s) {
    LSAM_REQUIRE(x && out && N > 0 && H > 0 && W > 0 && C > 0, LSAM_ERR_BAD_ARG, "maxpool: null/empty");
    LSAM_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, LSAM_ERR_UNSUPPORTED, "maxpool: need even H,W and C%%4==0");
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ProfScope ps(s, (double)N * H * W * C * 5.0, PROF_MAXPOOL, N * H * W, C, 0);
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid), dim3(256), 0, s, x, out, N, H, W, C / 4);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// ---------------------------------------------------------------------------------------------
// per-frame spatial mean of an NHWC map: out[n][c] = mean_{hw} x[n][hw][c]  (the pooled 896-d
// stage-4 feature of the feature-extraction pass, train_cl_kmean.py:189-207 / hieradet.py:284-302)
// One workgroup = one frame x 64 channels: 4 waves stride over the tokens, lanes over channels
// (256-byte coalesced rows), fixed-order LDS reduction -> deterministic.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spatial_mean_kernel(const float* __restrict__ x, float* __restrict__ out, int HW,
                                                           int C) {
    __shared__ float part[4][64];
    const int n = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C) {
        const float* p = x + (size_t)n * HW * C + c;
        for (int t = w; t < HW; t += 4) s += p[(size_t)t * C];
    }
    part[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < C)
        out[(size_t)n * C + c] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x])) / (float)HW;
}

int launch_spatial_mean(const float* x, float* out, int N, int HW, int C, hipStream_t s) {
    LSAM_REQUIRE(x && out && N > 0 && HW > 0 && C > 0, LSAM_ERR_BAD_ARG, "spatial_mean: null/empty");
    LSAM_REQUIRE(N <= 65535, LSAM_ERR_UNSUPPORTED, "spatial_mean: at most 65535 frames per call");
    ProfScope ps(s, (double)N * HW * C * 4.0, PROF_SPATIAL_MEAN, N * HW, C, 0);
    hipLaunchKernelGGL(spatial_mean_kernel, dim3(cdiv(C, 64), N), dim3(256), 0, s, x, out, HW, C);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// ---------------------------------------------------------------------------------------------
// fp32 -> T
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, int lds, T* __restrict__ dst, int ldd, int rows,
                            int c4) {
    const long total = (long)rows * c4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / c4), c = (int)(i % c4);
        store4(dst + (size_t)r * ldd + 4 * c, load4(src + (size_t)r * lds + 4 * c));
    }
}
template <typename T>
int launch_cast(const float* src, int lds, T* dst, int ldd, int rows, int cols, hipStream_t s) {
    LSAM_REQUIRE(src && dst && rows > 0 && cols > 0, LSAM_ERR_BAD_ARG, "cast: null/empty");
    LSAM_REQUIRE(cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0, LSAM_ERR_BAD_ARG, "cast: multiples of 4 required");
    const long total = (long)rows * (cols / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ProfScope ps(s, (double)rows * cols * (4 + sizeof(T)), PROF_CAST, rows, cols, 0);
    hipLaunchKernelGGL(cast_kernel<T>, dim3(grid), dim3(256), 0, s, src, lds, dst, ldd, rows, cols / 4);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
template int launch_cast<float>(const float*, int, float*, int, int, int, hipStream_t);
template int launch_cast<bf16>(const float*, int, bf16*, int, int, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// im2col for PatchEmbed: 7x7 / stride 4 / pad 3, 3 input channels (backbones/utils.py:85-93).
// One thread produces 4 consecutive k of one token.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void im2col_patch_kernel(const float* __restrict__ img, T* __restrict__ cols, int ldc, int N, int S) {
    const int G = S >> 2, k4n = ldc >> 2;
    const long total = (long)N * G * G * k4n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int kc = (int)(i % k4n);
        const long tok = i / k4n;
        long t = tok;
        const int ox = (int)(t % G);
        t /= G;
        const int oy = (int)(t % G), n = (int)(t / G);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kc * 4 + j;
            float v = 0.f;
            if (k < 147) {
                const int c = k / 49, rem = k - c * 49, ky = rem / 7, kx = rem - ky * 7;
                const int iy = 4 * oy - 3 + ky, ix = 4 * ox - 3 + kx;
                if (iy >= 0 && iy < S && ix >= 0 && ix < S) v = img[(((size_t)n * 3 + c) * S + iy) * S + ix];
            }
            o[j] = v;
        }
        store4(cols + (size_t)tok * ldc + 4 * kc, o);
    }
}
template <typename T> int launch_im2col_patch(const float* img, T* cols, int ldc, int N, int S, hipStream_t s) {
    LSAM_REQUIRE(img && cols && N > 0 && S > 0, LSAM_ERR_BAD_ARG, "im2col: null/empty");
    LSAM_REQUIRE(S % 4 == 0 && ldc >= 148 && ldc % 16 == 0, LSAM_ERR_BAD_ARG, "im2col: S%%4, ldc>=148, ldc%%16");
    const long total = (long)N * (S / 4) * (S / 4) * (ldc / 4);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(im2col_patch_kernel<T>, dim3(grid), dim3(256), 0, s, img, cols, ldc, N, S);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
template int launch_im2col_patch<float>(const float*, float*, int, int, int, hipStream_t);
template int launch_im2col_patch<bf16>(const float*, bf16*, int, int, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// G[c][k] = sum_m X[m][c] * Y[m][k],  k < 16: LoRA dA / dB.  Pass 1: each single-wave block owns
// 256 columns x OR_MCH rows and writes an fp32 partial; pass 2 sums the partials in a fixed
// order, so the result is deterministic.
// ---------------------------------------------------------------------------------------------
constexpr int OR_MCH = 128;
size_t outer_reduce_ws(int M, int C) { return (size_t)cdiv(M, OR_MCH) * (size_t)C * 16 * sizeof(float); }

template <typename T>
__global__ __launch_bounds__(64) void outer_partial_kernel(const T* __restrict__ X, int ldx,
                                                           const T* __restrict__ Y, int ldy, int M, int C,
                                                           float* __restrict__ partial) {
    __shared__ float ys[OR_MCH][16];
    const int lane = threadIdx.x;
    const int m0 = blockIdx.y * OR_MCH;
    const int rows = min(OR_MCH, M - m0);
    for (int i = lane; i < OR_MCH * 4; i += 64) {
        const int r = i >> 2, q = i & 3;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (r < rows) v = load4(Y + (size_t)(m0 + r) * ldy + 4 * q);
        *reinterpret_cast<f32x4*>(&ys[r][4 * q]) = v;
    }
    __syncthreads();
    const int c = (blockIdx.x * 64 + lane) * 4;
    if (c >= C) return;
    f32x4 acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rows; ++r) {
        const f32x4 xv = load4(X + (size_t)(m0 + r) * ldx + c);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const f32x4 yv = *reinterpret_cast<const f32x4*>(&ys[r][4 * kq]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * kq + j] += xv * yv[j];
        }
    }
    float* out = partial + ((size_t)blockIdx.y * C + c) * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            f32x4 o;
            o[0] = acc[4 * kq + 0][j]; o[1] = acc[4 * kq + 1][j];
            o[2] = acc[4 * kq + 2][j]; o[3] = acc[4 * kq + 3][j];
            store4(out + j * 16 + 4 * kq, o);
        }
    }
}
__global__ void outer_final_kernel(const float* __restrict__ partial, int nch, int C, int R, float* __restrict__ out,
                                   long out_cs, long out_ks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * R) return;
    const int c = i / R, k = i % R;
    float s = 0.f;
    for (int ch = 0; ch < nch; ++ch) s += partial[((size_t)ch * C + c) * 16 + k];
    out[c * out_cs + k * out_ks] = s;
}
template <typename T>
int launch_outer_reduce(const T* X, int ldx, const T* Y, int ldy, int M, int C, int R, float* out, long out_cs,
                        long out_ks, void* partial, hipStream_t s) {
    LSAM_REQUIRE(X && Y && out && partial && M > 0 && C > 0, LSAM_ERR_BAD_ARG, "outer_reduce: null/empty");
    LSAM_REQUIRE(R >= 1 && R <= 16 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldy >= 16, LSAM_ERR_BAD_ARG,
                 "outer_reduce: need 1<=R<=16, C%%4==0, 16 readable Y columns");
    const int nch = cdiv(M, OR_MCH);
    hipLaunchKernelGGL(outer_partial_kernel<T>, dim3(cdiv(C, 256), nch), dim3(64), 0, s, X, ldx, Y, ldy, M, C,
                       static_cast<float*>(partial));
    LSAM_LAUNCH_CHECK();
    hipLaunchKernelGGL(outer_final_kernel, dim3(cdiv(C * R, 256)), dim3(256), 0, s,
                       static_cast<const float*>(partial), nch, C, R, out, out_cs, out_ks);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
// Four reductions of one LoRA block (dA_q, dB_q, dA_v, dB_v) in ONE partial + ONE final launch: blockIdx.z selects the
// job.  The point is four times the workgroups in flight for a latency-bound kernel and two launches instead of eight.
// RQ = number of 4-column groups of Y that hold the rank (1 for r <= 4, the reference's setting): the columns beyond
// are zero by construction and are neither multiplied nor written (4x less VALU work at r = 4, which is what bounded the
// partial pass).  The final pass sums the chunk partials in a fixed order -- four interleaved slices per output, then the
// slices -- so the result is deterministic (run-to-run bit-identical), though not the same rounding as the one-job kernels.
template <typename T, int RQ>
__global__ __launch_bounds__(64) void outer_partial4_kernel(const OuterJobs j, int M, int C, float* __restrict__ partial,
                                                            size_t job_stride) {
    __shared__ float ys[OR_MCH][4 * RQ];
    const int z = blockIdx.z;
    const T* __restrict__ X = static_cast<const T*>(j.X[z]);
    const T* __restrict__ Y = static_cast<const T*>(j.Y[z]);
    const int ldx = j.ldx[z], ldy = j.ldy[z];
    const int lane = threadIdx.x;
    const int m0 = blockIdx.y * OR_MCH;
    const int rows = min(OR_MCH, M - m0);
    for (int i = lane; i < OR_MCH * RQ; i += 64) {
        const int r = i / RQ, q = i % RQ;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (r < rows) v = load4(Y + (size_t)(m0 + r) * ldy + 4 * q);
        *reinterpret_cast<f32x4*>(&ys[r][4 * q]) = v;
    }
    __syncthreads();
    const int c = (blockIdx.x * 64 + lane) * 4;
    if (c >= C) return;
    f32x4 acc[4 * RQ];
#pragma unroll
    for (int k = 0; k < 4 * RQ; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const T* xp = X + (size_t)m0 * ldx + c;
    int r = 0;
    for (; r + 4 <= rows; r += 4) {   // four rows' loads in flight
        f32x4 xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = load4(xp + (size_t)(r + u) * ldx);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int kq = 0; kq < RQ; ++kq) {
                const f32x4 yv = *reinterpret_cast<const f32x4*>(&ys[r + u][4 * kq]);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[4 * kq + jj] += xv[u] * yv[jj];
            }
    }
    for (; r < rows; ++r) {
        const f32x4 xv = load4(xp + (size_t)r * ldx);
#pragma unroll
        for (int kq = 0; kq < RQ; ++kq) {
            const f32x4 yv = *reinterpret_cast<const f32x4*>(&ys[r][4 * kq]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[4 * kq + jj] += xv * yv[jj];
        }
    }
    float* out = partial + z * job_stride + ((size_t)blockIdx.y * C + c) * 16;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
        for (int kq = 0; kq < RQ; ++kq) {
            f32x4 o;
            o[0] = acc[4 * kq + 0][jj]; o[1] = acc[4 * kq + 1][jj];
            o[2] = acc[4 * kq + 2][jj]; o[3] = acc[4 * kq + 3][jj];
            store4(out + jj * 16 + 4 * kq, o);
        }
    }
}
// 64 outputs per workgroup; wave s of the four sums the chunks ch = s, s + 4, ...; the four slice sums meet in LDS
__global__ __launch_bounds__(256) void outer_final4_kernel(const OuterJobs j, const float* __restrict__ partial, size_t job_stride,
                                                           int nch, int C, int R) {
    __shared__ float sl[4][64];
    const int z = blockIdx.y;
    const int o = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;
    const bool ok = i < C * R;
    const int c = ok ? i / R : 0, k = ok ? i % R : 0;
    const float* p = partial + z * job_stride + (size_t)c * 16 + k;
    float s = 0.f;
    for (int ch = slice; ch < nch; ch += 4) s += p[(size_t)ch * C * 16];
    sl[slice][o] = s;
    __syncthreads();
    if (slice == 0 && ok) j.out[z][c * j.cs[z] + k * j.ks[z]] = (sl[0][o] + sl[1][o]) + (sl[2][o] + sl[3][o]);
}
template <typename T> int launch_outer_reduce4(const OuterJobs& j, int M, int C, int R, void* partial, hipStream_t s) {
    LSAM_REQUIRE(partial && M > 0 && C > 0, LSAM_ERR_BAD_ARG, "outer_reduce4: null/empty");
    for (int z = 0; z < 4; ++z)
        LSAM_REQUIRE(j.X[z] && j.Y[z] && j.out[z] && j.ldx[z] % 4 == 0 && j.ldy[z] % 4 == 0 && j.ldy[z] >= 16, LSAM_ERR_BAD_ARG,
                     "outer_reduce4: job %d operands", z);
    LSAM_REQUIRE(R >= 1 && R <= 16 && C % 4 == 0, LSAM_ERR_BAD_ARG, "outer_reduce4: need 1<=R<=16, C%%4==0");
    const int nch = cdiv(M, OR_MCH);
    const size_t job_stride = (size_t)nch * C * 16;
    ProfScope ps(s, 4.0 * M * (C + 16) * sizeof(T), PROF_OUTER, M, C, R);
    const dim3 grid(cdiv(C, 256), nch, 4);
    float* pp = static_cast<float*>(partial);
    if (R <= 4) hipLaunchKernelGGL((outer_partial4_kernel<T, 1>), grid, dim3(64), 0, s, j, M, C, pp, job_stride);
    else if (R <= 8) hipLaunchKernelGGL((outer_partial4_kernel<T, 2>), grid, dim3(64), 0, s, j, M, C, pp, job_stride);
    else hipLaunchKernelGGL((outer_partial4_kernel<T, 4>), grid, dim3(64), 0, s, j, M, C, pp, job_stride);
    LSAM_LAUNCH_CHECK();
    hipLaunchKernelGGL(outer_final4_kernel, dim3(cdiv(C * R, 64), 4), dim3(256), 0, s, j, static_cast<const float*>(partial),
                       job_stride, nch, C, R);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
template int launch_outer_reduce4<float>(const OuterJobs&, int, int, int, void*, hipStream_t);
template int launch_outer_reduce4<bf16>(const OuterJobs&, int, int, int, void*, hipStream_t);

template int launch_outer_reduce<float>(const float*, int, const float*, int, int, int, int, float*, long, long, void*, hipStream_t);
template int launch_outer_reduce<bf16>(const bf16*, int, const bf16*, int, int, int, int, float*, long, long, void*, hipStream_t);

// ---------------------------------------------------------------------------------------------
// launch_outer_reduce for callers that cannot allocate (code under HIP-graph capture, tight serving
// loops): the fp32 partial sums live in a file-scope device array instead of a caller workspace.
// Every GPU the library runs on holds its own copy of `scratch` (OR_STATIC_BYTES) until the process
// ends.  In host code the NAME `scratch` is only a handle for hipGetSymbolAddress, never a device
// address: the per-device address is resolved once, with that device current, and from then on
// only that pointer is handed to kernels.  Pass 1 overwrites every partial pass 2 reads, so nothing
// has to be cleared between calls; calls that overlap on one GPU (two streams) race on the array.
// ---------------------------------------------------------------------------------------------
constexpr size_t OR_STATIC_BYTES = size_t(8) << 20;
static __device__ __align__(256) float scratch[OR_STATIC_BYTES / sizeof(float)];
constexpr int OR_MAX_DEVICES = 64;
static std::mutex g_scratch_mu;
static float* g_scratch_ptr[OR_MAX_DEVICES] = {};

size_t outer_reduce_static_max_bytes() { return OR_STATIC_BYTES; }

static int scratch_of_current_device(float** out) {
    int dev = -1;
    LSAM_HIP(hipGetDevice(&dev));
    LSAM_REQUIRE(dev >= 0 && dev < OR_MAX_DEVICES, LSAM_ERR_UNSUPPORTED, "outer_reduce_static: device index %d", dev);
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        LSAM_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        LSAM_REQUIRE(p, LSAM_ERR_HIP, "outer_reduce_static: hipGetSymbolAddress returned null on device %d", dev);
        g_scratch_ptr[dev] = static_cast<float*>(p);
    }
    *out = g_scratch_ptr[dev];
    return LSAM_OK;
}

int outer_reduce_static_init() {
    float* p = nullptr;
    return scratch_of_current_device(&p);
}

template <typename T>
int launch_outer_reduce_static(const T* X, int ldx, const T* Y, int ldy, int M, int C, int R, float* out, long out_cs,
                               long out_ks, hipStream_t s) {
    LSAM_REQUIRE(M > 0 && C > 0, LSAM_ERR_BAD_ARG, "outer_reduce_static: null/empty");
    const size_t need = outer_reduce_ws(M, C);
    LSAM_REQUIRE(need <= OR_STATIC_BYTES, LSAM_ERR_WORKSPACE,
                 "outer_reduce_static: M=%d, C=%d needs %zu bytes of partial sums, the built-in scratch holds %zu "
                 "(ceil(M/128)*C*64 <= 8 MiB); use lsam_outer_reduce with a caller workspace",
                 M, C, need, OR_STATIC_BYTES);
    float* part = nullptr;
    LSAM_TRY(scratch_of_current_device(&part));
    return launch_outer_reduce<T>(X, ldx, Y, ldy, M, C, R, out, out_cs, out_ks, part, s);
}
template int launch_outer_reduce_static<float>(const float*, int, const float*, int, int, int, int, float*, long, long, hipStream_t);
template int launch_outer_reduce_static<bf16>(const bf16*, int, const bf16*, int, int, int, int, float*, long, long, hipStream_t);

}  // namespace lsam
