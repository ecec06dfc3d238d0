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

}  // namespace lsam
