// Fused linear layer:  out = epilogue(A . W^T + bias) + res   on MFMA (gfx950).
//
// Both operands are K-contiguous (activations [M][K], nn.Linear weight [N][K]), so a k-tile of
// either is a stack of 128-byte row segments.  A workgroup owns a BM x BN output tile; k-tiles
// go global -> registers -> XOR-swizzled LDS (double-buffered, one barrier per k-tile, the next
// tile's global loads are in flight while the current one is multiplied).  The product is
// formed TRANSPOSED (W fragment as the MFMA A operand, activation fragment as B), so every lane
// ends up with 4 consecutive output columns of one row: bias / residual / GELU / dtype
// conversion are then 16-byte (fp32) or 8-byte (bf16) vector operations in the epilogue.
//
// Hiera-B+ layer widths are all multiples of 112 = 7*16, hence the 128 x 112 tile; other widths
// use 128 x 128 with column guards.
#include <stdlib.h>

#include "kernels.hpp"

namespace lsam {

template <typename T, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const GemmArgs p) {
    constexpr int EPC = Tr<T>::EPC, CPF = Tr<T>::CPF;
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, NTHR = 64 * WM * WN;
    constexpr int BK = 8 * EPC;         // elements per k-tile: one 128-byte row segment
    constexpr int KS = 8 / (4 * CPF);   // K=32 macro-steps per k-tile (bf16: 2, f32: 1)
    constexpr int A_IT = (BM * 8 + NTHR - 1) / NTHR, W_IT = (BN * 8 + NTHR - 1) / NTHR;
    constexpr int STAGE = (BM + BN) * 128;
    using frag = typename Tr<T>::frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int g = lane >> 4, lr = lane & 15;
    const int M = p.M, N = p.N, K = p.K;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (id / tiles_n) * BM, n0 = (id % tiles_n) * BN;
    const T* __restrict__ A = static_cast<const T*>(p.A);
    const T* __restrict__ Wt = static_cast<const T*>(p.W);

    uint4 ra[A_IT], rw[W_IT];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + i * NTHR, row = idx >> 3, c = idx & 7;
            const int gm = m0 + row, gk = k0 + c * EPC;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < BM * 8 && gm < M && gk < K)
                v = *reinterpret_cast<const uint4*>(A + (size_t)gm * p.lda + gk);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            const int idx = tid + i * NTHR, row = idx >> 3, c = idx & 7;
            const int gn = n0 + row, gk = k0 + c * EPC;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < BN * 8 && gn < N && gk < K)
                v = *reinterpret_cast<const uint4*>(Wt + (size_t)gn * p.ldw + gk);
            rw[i] = v;
        }
    };
    auto lstore = [&](int buf) {
        char* a = smem + buf * STAGE;
        char* w = a + BM * 128;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + i * NTHR, row = idx >> 3, c = idx & 7;
            if (idx < BM * 8) *reinterpret_cast<uint4*>(a + row * 128 + (swz(row, c) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            const int idx = tid + i * NTHR, row = idx >> 3, c = idx & 7;
            if (idx < BN * 8) *reinterpret_cast<uint4*>(w + row * 128 + (swz(row, c) << 4)) = rw[i];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto read_frag = [&](const char* base, int row, int c0) -> frag {
        const uint4 x0 = *reinterpret_cast<const uint4*>(base + row * 128 + (swz(row, c0) << 4));
        uint4 x1 = x0;
        if constexpr (CPF == 2)
            x1 = *reinterpret_cast<const uint4*>(base + row * 128 + (swz(row, c0 + 1) << 4));
        return make_frag(x0, x1, T{});
    };
    auto compute = [&](int buf) {
        const char* a = smem + buf * STAGE;
        const char* w = a + BM * 128;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c0 = (ks * 4 + g) * CPF;
            frag af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = read_frag(a, (wm * MT + mt) * 16 + lr, c0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const frag wf = read_frag(w, (wn * NT + nt) * 16 + lr, c0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mma16(wf, af[mt], acc[mt][nt]);
            }
        }
    };

    const int nkt = (K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) gload(kt + 1);
        compute(cur);
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane holds C[m][n..n+3] -------------------------------------------------
    const int flags = p.flags;
    const bool out_f32 = (flags & LSAM_EPI_OUT_F32) || sizeof(T) == 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + (wm * MT + mt) * 16 + lr;
        if (m >= M) continue;
        const size_t rm = p.res_row_mod > 0 ? (size_t)(m % p.res_row_mod) : (size_t)m;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + (wn * NT + nt) * 16 + 4 * g;
            if (n >= N) continue;
            f32x4 v = acc[mt][nt];
            if (p.bias) v += load4(p.bias + n);
            if (flags & LSAM_EPI_SAVE_PRE) store4(static_cast<T*>(p.out2) + (size_t)m * p.ldo2 + n, v);
            if (flags & LSAM_EPI_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
            }
            if (flags & LSAM_EPI_GELU_BWD) {
                const f32x4 x = load4(static_cast<const T*>(p.aux) + (size_t)m * p.ldaux + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= gelu_grad_f(x[r]);
            }
            if (p.res) v += load4(p.res + rm * p.ldres + n);
            if (out_f32)
                store4(static_cast<float*>(p.out) + (size_t)m * p.ldo + n, v);
            else
                store4(static_cast<T*>(p.out) + (size_t)m * p.ldo + n, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Optional per-launch timing of the GEMM kernels (bench.py's live roofline measurement).  OFF by
// default; when armed, every launch is bracketed by a pre-created hipEvent pair on the launch
// stream.  Process-global and not thread-safe by design: a measurement aid, not part of the path.
// ---------------------------------------------------------------------------------------------
struct ProfRec { hipEvent_t e0, e1; double flops; int cfg, m, n, k; };
static ProfRec* g_prof = nullptr;
static int g_prof_cap = 0, g_prof_n = 0;
static bool g_prof_on = false;

int prof_arm(int capacity) {
    if (capacity > g_prof_cap) {
        ProfRec* n = static_cast<ProfRec*>(realloc(g_prof, sizeof(ProfRec) * capacity));
        LSAM_REQUIRE(n, LSAM_ERR_BAD_ARG, "prof: out of host memory");
        g_prof = n;
        for (int i = g_prof_cap; i < capacity; ++i) {
            LSAM_HIP(hipEventCreate(&g_prof[i].e0));
            LSAM_HIP(hipEventCreate(&g_prof[i].e1));
        }
        g_prof_cap = capacity;
    }
    g_prof_n = 0;
    g_prof_on = capacity > 0;
    return LSAM_OK;
}
int prof_read(int max, float* ms, double* flops, int* cfg, int* mnk) {
    g_prof_on = false;
    const int n = g_prof_n < max ? g_prof_n : max;
    for (int i = 0; i < n; ++i) {
        LSAM_HIP(hipEventSynchronize(g_prof[i].e1));
        LSAM_HIP(hipEventElapsedTime(&ms[i], g_prof[i].e0, g_prof[i].e1));
        flops[i] = g_prof[i].flops;
        cfg[i] = g_prof[i].cfg;
        if (mnk) { mnk[3 * i] = g_prof[i].m; mnk[3 * i + 1] = g_prof[i].n; mnk[3 * i + 2] = g_prof[i].k; }
    }
    return n;
}

template <typename T, int MT, int NT, int WM, int WN>
static int launch_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr int LDS = 2 * (BM + BN) * 128;
    auto kern = gemm_kernel<T, MT, NT, WM, WN>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int nblk = cdiv(a.M, BM) * cdiv(a.N, BN);
    const bool prof = g_prof_on && g_prof_n < g_prof_cap;
    if (prof) LSAM_HIP(hipEventRecord(g_prof[g_prof_n].e0, s));
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * WM * WN), LDS, s, a);
    LSAM_LAUNCH_CHECK();
    if (prof) {
        LSAM_HIP(hipEventRecord(g_prof[g_prof_n].e1, s));
        g_prof[g_prof_n].flops = 2.0 * a.M * a.N * a.K;
        g_prof[g_prof_n].cfg = (int)sizeof(T) * 1000 + BN;
        g_prof[g_prof_n].m = a.M; g_prof[g_prof_n].n = a.N; g_prof[g_prof_n].k = a.K;
        ++g_prof_n;
    }
    return LSAM_OK;
}

template <typename T> int launch_gemm(const GemmArgs& a, hipStream_t s) {
    constexpr int EPC = Tr<T>::EPC;
    LSAM_REQUIRE(a.A && a.W && a.out, LSAM_ERR_BAD_ARG, "linear: null operand");
    LSAM_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, LSAM_ERR_BAD_ARG, "linear: empty shape %dx%dx%d", a.M, a.N, a.K);
    LSAM_REQUIRE(a.N % 8 == 0 && a.K % EPC == 0, LSAM_ERR_UNSUPPORTED,
                 "linear: need N%%8==0 and K%%%d==0 (N=%d K=%d)", EPC, a.N, a.K);
    LSAM_REQUIRE(a.lda % EPC == 0 && a.ldw % EPC == 0 && aligned16(a.A) && aligned16(a.W),
                 LSAM_ERR_BAD_ARG, "linear: operand rows must be 16-byte aligned");
    LSAM_REQUIRE(a.lda >= a.K && a.ldw >= a.K && a.ldo >= a.N, LSAM_ERR_BAD_ARG, "linear: bad leading dimension");
    LSAM_REQUIRE(a.ldo % 4 == 0 && aligned16(a.out) && (!a.bias || aligned16(a.bias)), LSAM_ERR_BAD_ARG,
                 "linear: out/bias alignment");
    LSAM_REQUIRE(!a.res || (a.ldres % 4 == 0 && aligned16(a.res)), LSAM_ERR_BAD_ARG, "linear: res alignment");
    LSAM_REQUIRE(!(a.flags & LSAM_EPI_GELU_BWD) || (a.aux && a.ldaux % 4 == 0), LSAM_ERR_BAD_ARG,
                 "linear: GELU_BWD needs aux");
    LSAM_REQUIRE(!(a.flags & LSAM_EPI_SAVE_PRE) || (a.out2 && a.ldo2 % 4 == 0), LSAM_ERR_BAD_ARG,
                 "linear: SAVE_PRE needs out2");
    if (a.N % 112 == 0) return launch_cfg<T, 2, 7, 4, 1>(a, s);
    return launch_cfg<T, 4, 4, 2, 2>(a, s);
}
template int launch_gemm<float>(const GemmArgs&, hipStream_t);
template int launch_gemm<bf16>(const GemmArgs&, hipStream_t);

// ---------------------------------------------------------------------------------------------
// dst[r][c] = (T) src[r*rs + c*cs], zero-filled up to ldd
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_kernel(const float* __restrict__ src, long rs, long cs, int rows, int cols,
                            T* __restrict__ dst, int ldd) {
    const long total = (long)rows * ldd;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / ldd), c = (int)(i % ldd);
        dst[i] = from_f<T>(c < cols ? src[r * rs + c * cs] : 0.f);
    }
}
template <typename T>
int launch_pack_matrix(const float* src, long rs, long cs, int rows, int cols, T* dst, int ldd, hipStream_t s) {
    LSAM_REQUIRE(src && dst && rows > 0 && cols > 0 && ldd >= cols, LSAM_ERR_BAD_ARG, "pack_matrix: bad args");
    const long total = (long)rows * ldd;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_kernel<T>, dim3(grid), dim3(256), 0, s, src, rs, cs, rows, cols, dst, ldd);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
template int launch_pack_matrix<float>(const float*, long, long, int, int, float*, int, hipStream_t);
template int launch_pack_matrix<bf16>(const float*, long, long, int, int, bf16*, int, hipStream_t);

}  // namespace lsam
