// Lateral 1x1 convolution of the FPN neck for the two high-resolution levels, straight from the fp32 stage output:
//   out[m][:] = bf16( x[m][:] . W^T + b ),  x fp32 [M][K] (NHWC tokens), W [256][K], out bf16 [M][256]
// (FpnNeck.forward, image_encoder.py:107-110: `lateral_features = self.convs[n - i](x)` with kernel_size 1).
// As cast + GEMM the level-0 conv of a 16-frame batch reads x (470 MB), writes and re-reads its bf16 copy and
// writes out: here x is read once (converted to MFMA operand fragments in registers: a lane's two 16-byte
// loads ARE its k-contiguous fragment) and the 256 x K weight matrix stays in LDS in fragment order.
#include "kernels.hpp"

namespace lsam {

namespace {
constexpr int LAT_N = 256, LAT_NT = LAT_N / 16;
}

template <int KS>
__global__ void pack_lateral_kernel(const float* __restrict__ w, int K, char* __restrict__ img) {
    const int total = LAT_NT * KS * 512;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int f = i >> 9, lane = (i & 511) >> 3, j = i & 7;
        const int nt = f / KS, ks = f % KS, g = lane >> 4, lr = lane & 15;
        const int k = 32 * ks + 8 * g + j;
        reinterpret_cast<bf16*>(img)[i] = (bf16)(k < K ? w[(size_t)(16 * nt + lr) * K + k] : 0.f);
    }
}

template <int KS>
__global__ __launch_bounds__(512) void lateral_kernel(const float* __restrict__ x, const char* __restrict__ wimg,
                                                                       const float* __restrict__ bias, bf16* __restrict__ out,
                                                                       int M, int K) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, lr = lane & 15;
    for (int i = tid; i < LAT_NT * KS * 64; i += 512)
        *reinterpret_cast<uint4*>(smem + i * 16) = *reinterpret_cast<const uint4*>(wimg + i * 16);
    __syncthreads();

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int ngroups = (M + 63) / 64;
#pragma unroll 1
    for (int grp = blockIdx.x * 8 + wave; grp < ngroups; grp += gridDim.x * 8) {
        const int m0 = grp * 64;
        bf16x8 xf[4][KS];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float* row = x + (size_t)min(m0 + 16 * mt + lr, M - 1) * K;   // clamped rows are never stored
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c0 = 32 * ks + 8 * g;
                f32x4 a = zero4, b = zero4;
                if (c0 < K) { a = load4(row + c0); b = load4(row + c0 + 4); }
#pragma unroll
                for (int j = 0; j < 4; ++j) { xf[mt][ks][j] = (bf16)a[j]; xf[mt][ks][4 + j] = (bf16)b[j]; }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {   // 64 output channels at a time bound the live accumulators
            f32x4 acc[4][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 b = load4(bias + 16 * (4 * qt + nt) + 4 * g);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = b;
            }
            // weight fragments one k-step ahead of their MFMAs
            bf16x8 wb[2][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                wb[0][nt] = *reinterpret_cast<const bf16x8*>(smem + ((4 * qt + nt) * KS) * 1024 + lane * 16);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        wb[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(smem + ((4 * qt + nt) * KS + ks + 1) * 1024 + lane * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) mma16(wb[ks & 1][nt], xf[mt][ks], acc[mt][nt]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // bf16 stores: two neighbouring column tiles are regrouped across the four lanes of a token so that
            // every lane stores 16 bytes = 8 consecutive channels (as the GEMM epilogue does)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = m0 + 16 * mt + lr;
                bf16* rowp = out + (size_t)min(m, M - 1) * LAT_N + 64 * qt;
#pragma unroll
                for (int nt = 0; nt < 4; nt += 2) {
                    bf16x4 p0, p1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { p0[r] = (bf16)acc[mt][nt][r]; p1[r] = (bf16)acc[mt][nt + 1][r]; }
                    const uint2 xa = __builtin_bit_cast(uint2, p0), ya = __builtin_bit_cast(uint2, p1);
                    auto r0 = __builtin_amdgcn_permlane32_swap(xa.x, ya.x, false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(xa.y, ya.y, false, false);
                    auto q0 = __builtin_amdgcn_permlane16_swap(r0[0], r0[1], false, false);
                    auto q1 = __builtin_amdgcn_permlane16_swap(r1[0], r1[1], false, false);
                    if (m < M) *reinterpret_cast<uint4*>(rowp + nt * 16 + 8 * g) = make_uint4(q0[0], q1[0], q0[1], q1[1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

static int lat_variant(int K, int N) {
    if (N != LAT_N || K % 8 != 0 || K < 16) return -1;
    if (K <= 128) return 0;
    if (K <= 224) return 1;
    return -1;
}
bool lateral_supported(int K, int N) { return lat_variant(K, N) >= 0; }
size_t lateral_image_bytes(int K, int N) {
    switch (lat_variant(K, N)) {
        case 0: return (size_t)LAT_NT * 4 * 1024;
        case 1: return (size_t)LAT_NT * 7 * 1024;
    }
    return 0;
}

int launch_pack_lateral(const float* w, int K, int N, void* img, hipStream_t s) {
    LSAM_REQUIRE(w && img, LSAM_ERR_BAD_ARG, "pack_lateral: null argument");
    const int v = lat_variant(K, N);
    LSAM_REQUIRE(v >= 0, LSAM_ERR_UNSUPPORTED, "pack_lateral: K %d, N %d not supported", K, N);
    if (v == 0) hipLaunchKernelGGL(pack_lateral_kernel<4>, dim3(64), dim3(256), 0, s, w, K, static_cast<char*>(img));
    else hipLaunchKernelGGL(pack_lateral_kernel<7>, dim3(64), dim3(256), 0, s, w, K, static_cast<char*>(img));
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

template <int KS> static int launch_lat(const float* x, const void* img, const float* bias, bf16* out, int M, int K, hipStream_t s) {
    constexpr int LDSB = LAT_NT * KS * 1024;
    auto kern = lateral_kernel<KS>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        attr_set = true;
    }
    const int ngroups = cdiv(M, 64);
    const int grid = cdiv(ngroups, 8) < 256 ? cdiv(ngroups, 8) : 256;   // persistent: one 8-wave workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDSB, s, x, static_cast<const char*>(img), bias, out, M, K);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

int launch_lateral(const float* x, const void* img, const float* bias, void* out, int M, int K, int N, hipStream_t s) {
    LSAM_REQUIRE(x && img && bias && out && M > 0, LSAM_ERR_BAD_ARG, "lateral: null/empty argument");
    const int v = lat_variant(K, N);
    LSAM_REQUIRE(v >= 0, LSAM_ERR_UNSUPPORTED, "lateral: K %d, N %d not supported", K, N);
    LSAM_REQUIRE(aligned16(x) && aligned16(img) && aligned16(bias) && aligned16(out), LSAM_ERR_BAD_ARG,
                 "lateral: pointers must be 16-byte aligned");
    ProfScope ps(s, (double)M * (4.0 * K + 2.0 * N), PROF_LATERAL, M, N, K);
    if (v == 0) return launch_lat<4>(x, img, bias, static_cast<bf16*>(out), M, K, s);
    return launch_lat<7>(x, img, bias, static_cast<bf16*>(out), M, K, s);
}


// ---------------------------------------------------------------------------------------------
// The same scheme for the trunk's SMALL-K projections (K <= 224: the qkv / k | v Linears of stage 2, hieradet.py:59):
//   out[m][n] = bf16( A[m][:] . W[n][:]^T + b[n] ),  A bf16 [M][lda], W [N][K], out bf16 [M][ldo]
// In the tiled GEMM such a product is all epilogue: a 256 x 224 output tile has 3.5 k-tiles of MFMAs (6 k cycles) in front
// of 128 store instructions (11 k cycles of the CU's store path), the eight waves do both in lock-step, and the launch
// runs at 2.4 TB/s of output (262144 x 672 x 224: 170 us).  Here the weights of a 224-column group stay in LDS in fragment
// order (14 tiles x 7 k-steps x 1 KiB = 98 KiB), every wave streams its own 64-row groups -- a lane's 16-byte loads of A
// ARE its k-contiguous B fragments, no staging, no barrier -- and waves drift apart, so one's stores fall under the
// others' MFMAs.  Column groups are separate workgroups of one launch; the workgroups that read the same rows for
// different column groups sit on one XCD (consecutive per-XCD indices), so A crosses HBM once.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int SL_NTG = 14;   // column tiles per group
}

template <int KS>
__global__ void pack_streamlin_kernel(const float* __restrict__ w, int K, int N, char* __restrict__ img) {
    const long total = (long)(N / 16) * KS * 512;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i >> 9), lane = (int)(i & 511) >> 3, j = (int)(i & 7);
        const int nt = f / KS, ks = f % KS, g = lane >> 4, lr = lane & 15;
        const int k = 32 * ks + 8 * g + j;
        reinterpret_cast<bf16*>(img)[i] = (bf16)(k < K ? w[(size_t)(16 * nt + lr) * K + k] : 0.f);
    }
}

template <int KS>
__global__ __launch_bounds__(512) void streamlin_kernel(const bf16* __restrict__ A, int lda, const char* __restrict__ wimg,
                                                        const float* __restrict__ bias, bf16* __restrict__ out, int ldo, int M, int K,
                                                        int N, int ngroups, int nrow_wg) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, lr = lane & 15;
    // workgroup id -> (column group, row slot): the ngroups workgroups of a row slot are consecutive on ONE XCD
    const int xcd = blockIdx.x & 7, j_ = blockIdx.x >> 3;
    const int cg = j_ % ngroups, rslot = (j_ / ngroups) * 8 + xcd;
    if (rslot >= nrow_wg) return;   // workgroup-uniform, before the barrier
    const int nt0 = cg * SL_NTG, ntiles = min(SL_NTG, N / 16 - nt0);
    for (int i = tid; i < ntiles * KS * 64; i += 512)
        *reinterpret_cast<uint4*>(smem + i * 16) = *reinterpret_cast<const uint4*>(wimg + ((size_t)nt0 * KS * 64 + i) * 16);
    __syncthreads();

    constexpr int MT = 2;   // 32 rows per group: two register sets of A fragments fit, so the next group's rows are in flight
    const uint4 zero16 = make_uint4(0u, 0u, 0u, 0u);
    const int nrg = (M + 16 * MT - 1) / (16 * MT), stride = nrow_wg * 8;
    auto load_rows = [&](int grp, bf16x8 (&xf)[MT][KS]) {
        const int m0 = grp * 16 * MT;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bf16* row = A + (size_t)min(m0 + 16 * mt + lr, M - 1) * lda;   // clamped rows are never stored
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c0 = 32 * ks + 8 * g;
                uint4 v = zero16;
                if (c0 < K) v = *reinterpret_cast<const uint4*>(row + c0);
                xf[mt][ks] = __builtin_bit_cast(bf16x8, v);
            }
        }
    };
    auto pass = [&](int grp, const bf16x8 (&xf)[MT][KS], int q0, auto nq_) {   // NQ column tiles of the group from tile q0
        constexpr int NQ = decltype(nq_)::value;
        const int m0 = grp * 16 * MT;
        f32x4 acc[MT][NQ];
#pragma unroll
        for (int nt = 0; nt < NQ; ++nt) {
            const f32x4 b = load4(bias + 16 * (nt0 + q0 + nt) + 4 * g);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = b;
        }
        const char* wq = smem + (size_t)q0 * KS * 1024 + lane * 16;
        bf16x8 wb[2][NQ];   // weight fragments one k-step ahead of their MFMAs
#pragma unroll
        for (int nt = 0; nt < NQ; ++nt) wb[0][nt] = *reinterpret_cast<const bf16x8*>(wq + (nt * KS) * 1024);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int nt = 0; nt < NQ; ++nt) wb[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(wq + (nt * KS + ks + 1) * 1024);
            }
#pragma unroll
            for (int nt = 0; nt < NQ; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mma16(wb[ks & 1][nt], xf[mt][ks], acc[mt][nt]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + 16 * mt + lr;
            bf16* rowp = out + (size_t)min(m, M - 1) * ldo + 16 * (nt0 + q0);
#pragma unroll
            for (int nt = 0; nt < NQ; nt += 2)   // every lane takes part in the swaps; the guard is on the store only
                store_bf16_tile_pair(rowp + nt * 16, g, acc[mt][nt], acc[mt][nt + 1], m < M);
        }
    };
    auto compute = [&](int grp, const bf16x8 (&xf)[MT][KS]) {
#pragma unroll 1
        for (int q0 = 0; q0 < ntiles; q0 += 4) {   // 64 output channels at a time bound the live accumulators
            if (ntiles - q0 >= 4) pass(grp, xf, q0, std::integral_constant<int, 4>{});
            else pass(grp, xf, q0, std::integral_constant<int, 2>{});   // N % 32 == 0: the tail of a group is two tiles
        }
    };
    bf16x8 xa[MT][KS], xb[MT][KS];
    int grp = rslot * 8 + wave;
    if (grp < nrg) load_rows(grp, xa);
#pragma unroll 1
    for (; grp < nrg; grp += 2 * stride) {
        const bool hb = grp + stride < nrg;       // wave-uniform
        if (hb) load_rows(grp + stride, xb);
        compute(grp, xa);
        if (grp + 2 * stride < nrg) load_rows(grp + 2 * stride, xa);
        if (hb) compute(grp + stride, xb);
    }
}

static int sl_ks(int K) { return K <= 128 ? 4 : K <= 160 ? 5 : K <= 224 ? 7 : 0; }
bool streamlin_supported(int K, int N) { return K % 8 == 0 && K >= 16 && sl_ks(K) > 0 && N % 32 == 0 && N >= 32; }
size_t streamlin_image_bytes(int K, int N) { return streamlin_supported(K, N) ? (size_t)(N / 16) * sl_ks(K) * 1024 : 0; }

int launch_pack_streamlin(const float* w, int K, int N, void* img, hipStream_t s) {
    LSAM_REQUIRE(w && img, LSAM_ERR_BAD_ARG, "pack_streamlin: null argument");
    LSAM_REQUIRE(streamlin_supported(K, N), LSAM_ERR_UNSUPPORTED, "pack_streamlin: K %d, N %d not supported", K, N);
    char* d = static_cast<char*>(img);
    switch (sl_ks(K)) {
        case 4: hipLaunchKernelGGL(pack_streamlin_kernel<4>, dim3(128), dim3(256), 0, s, w, K, N, d); break;
        case 5: hipLaunchKernelGGL(pack_streamlin_kernel<5>, dim3(128), dim3(256), 0, s, w, K, N, d); break;
        default: hipLaunchKernelGGL(pack_streamlin_kernel<7>, dim3(128), dim3(256), 0, s, w, K, N, d); break;
    }
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

template <int KS>
static int launch_sl(const bf16* A, int lda, const void* img, const float* bias, bf16* out, int ldo, int M, int K, int N, hipStream_t s) {
    constexpr int LDSB = SL_NTG * KS * 1024;
    auto kern = streamlin_kernel<KS>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        attr_set = true;
    }
    const int ngroups = cdiv(N / 16, SL_NTG);
    const int want = cdiv(cdiv(M, 32), 8);                       // row slots that have at least one 32-row group per wave
    // persistent, one workgroup per CU (its weight slice takes most of the LDS): blockIdx % 8 picks the XCD, so each XCD's
    // 32 CUs hold 32 / ngroups row slots.  One workgroup more than that on an XCD runs after the others: twice the time.
    int nrow_wg = 8 * (32 / ngroups);
    if (nrow_wg < 1) nrow_wg = 1;
    if (nrow_wg > want) nrow_wg = want;
    const int grid = 8 * cdiv(nrow_wg, 8) * ngroups;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDSB, s, A, lda, static_cast<const char*>(img), bias, out, ldo, M, K, N, ngroups, nrow_wg);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

int launch_streamlin(const void* A, int lda, const void* img, const float* bias, void* out, int ldo, int M, int K, int N, hipStream_t s) {
    LSAM_REQUIRE(A && img && bias && out && M > 0, LSAM_ERR_BAD_ARG, "streamlin: null/empty argument");
    LSAM_REQUIRE(streamlin_supported(K, N), LSAM_ERR_UNSUPPORTED, "streamlin: K %d, N %d not supported", K, N);
    LSAM_REQUIRE(aligned16(A) && aligned16(img) && aligned16(bias) && aligned16(out) && lda % 8 == 0 && ldo % 8 == 0 && lda >= K && ldo >= N,
                 LSAM_ERR_BAD_ARG, "streamlin: alignment / leading dimensions");
    ProfScope ps(s, 2.0 * M * N * K, 2016, M, N, K);
    const bf16* a = static_cast<const bf16*>(A);
    bf16* o = static_cast<bf16*>(out);
    switch (sl_ks(K)) {
        case 4: return launch_sl<4>(a, lda, img, bias, o, ldo, M, K, N, s);
        case 5: return launch_sl<5>(a, lda, img, bias, o, ldo, M, K, N, s);
        default: return launch_sl<7>(a, lda, img, bias, o, ldo, M, K, N, s);
    }
}

}  // namespace lsam
