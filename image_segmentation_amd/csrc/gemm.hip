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

#include <type_traits>

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
                for (int r = 0; r < 4; ++r) v[r] = gelu_t<sizeof(T) == 2>(v[r]);
            }
            if (flags & LSAM_EPI_GELU_BWD) {
                const f32x4 x = load4(static_cast<const T*>(p.aux) + (size_t)m * p.ldaux + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= gelu_grad_t<sizeof(T) == 2>(x[r]);
            }
            if (flags & LSAM_EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (flags & LSAM_EPI_SIGMOID) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = 1.0f / (1.0f + expf(-v[r]));
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
// v2: 256 x 112 tile, 4 waves of 64 x 112 (28 accumulator tiles each), operands staged by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write), single LDS stage of 46 KiB so three
// workgroups share a CU and hide each other's load latency.  One wave-instruction of the DMA writes
// 1 KiB = 8 rows x 128 B linearly, so the XOR swizzle is applied to the per-lane SOURCE address
// (guide rule 21); out-of-range rows / k-chunks read a 16-byte zero word instead.
// ---------------------------------------------------------------------------------------------
#ifndef LSAM_ABL
#define LSAM_ABL 0
#endif
// 1: issue the next k-tile's DMA instructions between groups of MFMAs instead of in one burst in front of them.
// Measured on the stage-3 qkv shape (65536 x 1344 x 480): 125 us spread vs 114 us burst -- the loop is bound by
// the L2 -> LDS stream and the epilogue, not by DMA issue slots, and the burst starts that stream earlier.
#ifndef GEMM2_SPREAD_DMA
#define GEMM2_SPREAD_DMA 0
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ uint4 g_zero16 = {0u, 0u, 0u, 0u};

//
// WN = 1: 256 x 112 tile, 4 waves, one 46 KiB LDS stage, 3 workgroups per CU (used when N is an odd
//         multiple of 112: the HBM-bound stage-1 shapes).
// WN = 2: 256 x 224 tile, 8 waves (4 x 2), two 60 KiB LDS stages: the DMA of the next k-tile is
//         issued right after the barrier that publishes the current one and lands while it is
//         multiplied (one barrier per k-tile).
// Both are PERSISTENT: a workgroup walks output tiles t = b, b + G, ...  The epilogue's stores are
// fire-and-forget, so a tile's write-back (and, for WN = 2, the first k-tile DMA of the NEXT output
// tile, issued before the epilogue) overlaps the next tile's main loop.  With one output tile per
// workgroup every CU ran "all MFMA, then all stores" in lock-step and K <= 448 GEMMs spent ~70 % of
// their time in that store phase.
template <typename T, int WN, bool OUTF32, bool POOL = false>
__global__ __launch_bounds__(256 * WN, 2) void gemm2_kernel(const GemmArgs p) {
    static_assert(!POOL || (WN == 2 && OUTF32 && sizeof(T) == 2), "pooled epilogue: bf16, 256 x 224 tile, fp32 out");
    constexpr int EPC = Tr<T>::EPC, CPF = Tr<T>::CPF;
    constexpr int MT = 4, NT = 7, BM = 256, BN = 112 * WN, ROWS = BM + BN, NW = 4 * WN;
    constexpr int BK = 8 * EPC, KS = 8 / (4 * CPF);
    constexpr int A_IT = 32 / NW;                      // A: 32 DMA wave-instructions per k-tile
    constexpr int W_INSTR = BN / 8, W_IT = (W_INSTR + NW - 1) / NW;
    constexpr int STAGE = ROWS * 128;
    using frag = typename Tr<T>::frag;

    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int g = lane >> 4, lr = lane & 15;
    const int M = p.M, N = p.N, K = p.K;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int slot = xcd_remap(blockIdx.x, G);   // consecutive slots (same A rows) share an XCD's L2
    const T* __restrict__ A = static_cast<const T*>(p.A);
    const T* __restrict__ Wt = static_cast<const T*>(p.W);

    // per-lane DMA sources: A instruction j = wave + NW i covers A-tile rows 8j .. 8j+7 (j < 32),
    // W instruction j likewise for the W tile, so consecutive i are 8 NW rows apart.
    // Buffer addressing: 32-bit byte offsets, and the hardware range check turns every
    // out-of-range row / k-chunk (offset forced to 0xFFFFFFFF) into zeros.
    const int lrow = lane >> 3, ck = ((lane & 7) ^ lrow) * EPC;   // source k offset of this lane's chunk
    const unsigned astep = 8u * NW * (unsigned)p.lda * (unsigned)sizeof(T), wstep = 8u * NW * (unsigned)p.ldw * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(A), 0,
                                                         (int)(((size_t)(M - 1) * p.lda + K) * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Wt), 0,
                                                         (int)(((size_t)(N - 1) * p.ldw + K) * sizeof(T)), 0x00020000);
    // POOL: the 256 rows of an output tile are a TH x TW patch of the token grid (TW = min(W, 128), TH = 256 / TW)
    // laid out so that wave row-block wm owns a 2-row x 32-column sub-patch: tile row 64 wm + 16 mt + lr is token
    // (y0 + 2 wy + (mt >> 1), x0 + 32 wx + 16 (mt & 1) + lr), (wx, wy) = (wm % (TW/32), wm / (TW/32)).  The four
    // partners of a 2x2 pool are then two accumulator tiles (mt, mt + 2) of ONE lane and its neighbour lane lr ^ 1.
    const int pTW = POOL ? (p.pool_w < 128 ? p.pool_w : 128) : 1, pTH = POOL ? BM / pTW : 1, pWX = POOL ? pTW / 32 : 1;
    const int ptx = POOL ? p.pool_w / pTW : 1, pty = POOL ? p.pool_h / pTH : 1;
    struct TileAddr { int m0, n0, arow0, wrow0; unsigned aoff0, woff0; unsigned aoffp[POOL ? A_IT : 1]; int pn, py0, px0; };
    auto tile_addr = [&](int t) {
        TileAddr a;
        a.m0 = (t / tiles_n) * BM;
        a.n0 = (t % tiles_n) * BN;
        a.arow0 = a.m0 + 8 * wave + lrow;
        a.wrow0 = a.n0 + 8 * wave + lrow;
        a.aoff0 = ((unsigned)a.arow0 * (unsigned)p.lda + ck) * (unsigned)sizeof(T);
        a.woff0 = ((unsigned)a.wrow0 * (unsigned)p.ldw + ck) * (unsigned)sizeof(T);
        if constexpr (POOL) {
            const int tm = t / tiles_n;
            a.pn = tm / (ptx * pty);
            a.py0 = ((tm / ptx) % pty) * pTH;
            a.px0 = (tm % ptx) * pTW;
            // DMA instruction j = wave + 8 i covers tile rows 8 j .. 8 j + 7: wm = i, mt = wave >> 1, half = wave & 1
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int y = a.py0 + 2 * (i / pWX) + (wave >> 2), x = a.px0 + 32 * (i % pWX) + 16 * ((wave >> 1) & 1) + 8 * (wave & 1) + lrow;
                const unsigned tok = ((unsigned)a.pn * (unsigned)p.pool_h + (unsigned)y) * (unsigned)p.pool_w + (unsigned)x;
                a.aoffp[i] = (tok * (unsigned)p.lda + ck) * (unsigned)sizeof(T);
            }
        }
        return a;
    };
    auto read_frag = [&](const char* base, int row, int c0) -> frag {
        const uint4 x0 = *reinterpret_cast<const uint4*>(base + row * 128 + (swz(row, c0) << 4));
        uint4 x1 = x0;
        if constexpr (CPF == 2) x1 = *reinterpret_cast<const uint4*>(base + row * 128 + (swz(row, c0 + 1) << 4));
        return make_frag(x0, x1, T{});
    };
    auto issue = [&](const TileAddr& ta, int kt, int buf) {
#if LSAM_ABL & 1
        return;
#endif
        const int k0 = kt * BK;
        const bool kok = k0 + ck < K;
        const unsigned kb = (unsigned)k0 * (unsigned)sizeof(T);
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            unsigned vo;
            if constexpr (POOL) vo = kok ? ta.aoffp[i] + kb : 0xFFFFFFFFu;   // whole patches only: every row exists
            else vo = (ta.arow0 + 8 * NW * i < M && kok) ? ta.aoff0 + i * astep + kb : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_ptr_t)(base + (wave + NW * i) * 1024), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            const int j = wave + NW * i;
            if (j < W_INSTR) {
                const unsigned vo = (ta.wrow0 + 8 * NW * i < N && kok) ? ta.woff0 + i * wstep + kb : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(base + (32 + j) * 1024), 16, vo, 0, 0, 0);
            }
        }
    };

    // one DMA wave-instruction of a k-tile: idx < A_IT -> A piece idx, else W piece idx - A_IT
    auto issue_piece = [&](const TileAddr& ta, int kt, int buf, int idx) {
#if LSAM_ABL & 1
        return;
#endif
        const int k0 = kt * BK;
        const bool kok = k0 + ck < K;
        const unsigned kb = (unsigned)k0 * (unsigned)sizeof(T);
        char* base = smem + buf * STAGE;
        if (idx < A_IT) {
            const int i = idx;
            unsigned vo;
            if constexpr (POOL) vo = kok ? ta.aoffp[i] + kb : 0xFFFFFFFFu;
            else vo = (ta.arow0 + 8 * NW * i < M && kok) ? ta.aoff0 + i * astep + kb : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_ptr_t)(base + (wave + NW * i) * 1024), 16, vo, 0, 0, 0);
        } else {
            const int i = idx - A_IT, j = wave + NW * i;
            if (j < W_INSTR) {
                const unsigned vo = (ta.wrow0 + 8 * NW * i < N && kok) ? ta.woff0 + i * wstep + kb : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(base + (32 + j) * 1024), 16, vo, 0, 0, 0);
            }
        }
    };

    f32x4 acc[MT][NT];
    // All fragment reads of a k-tile (bf16: 22 ds_read_b128 per lane) are issued before its first MFMA,
    // and (WN = 2) before the DMA of the next k-tile: hipcc drains vmcnt(0) in front of any LDS read
    // that follows an LDS-DMA in program order, so reads placed after the DMA would serialise copy
    // and multiply.  The MFMAs then wait on counted lgkmcnt(N) as the reads return in order.
    auto load_frags = [&](int buf, frag (&af)[KS][MT], frag (&wf)[KS][NT]) {
        const char* base = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c0 = (ks * 4 + g) * CPF;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[ks][mt] = read_frag(base, wm * 64 + mt * 16 + lr, c0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[ks][nt] = read_frag(base, BM + wn * 112 + nt * 16 + lr, c0);
        }
    };
    // fp32 fragments are twice as wide (8 VGPRs): hoisting all of them would spill, so the fp32 path
    // (accuracy path, not the benchmarked one) reads one W fragment per 4 MFMAs.
    auto compute_interleaved = [&](int buf) {
        const char* base = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c0 = (ks * 4 + g) * CPF;
            frag a4[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a4[mt] = read_frag(base, wm * 64 + mt * 16 + lr, c0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const frag w1 = read_frag(base, BM + wn * 112 + nt * 16 + lr, c0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mma16(w1, a4[mt], acc[mt][nt]);
            }
        }
    };
    auto mfmas = [&](const frag (&af)[KS][MT], const frag (&wf)[KS][NT]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mma16(wf[ks][nt], af[ks][mt], acc[mt][nt]);
    };

    // ---- epilogue.  Lane (lr, g) holds C[m][16 nt + 4 g .. +3].  fp32 results go out as 16-byte
    // stores (64 contiguous bytes per row and instruction).  For bf16 results two neighbouring
    // column tiles are regrouped across the four lanes of a row with v_permlane32_swap +
    // v_permlane16_swap so every lane stores 16 bytes = 8 consecutive columns.
    // The global reads of the epilogue are batched: the 7 residual / GELU-backward operands of a
    // 16-row block are requested in one go (one memory round trip per row block instead of one per
    // 16 x 16 tile); the bias is already in the accumulators (init_acc).
    const int flags = p.flags;
    constexpr bool out_f32 = OUTF32 || sizeof(T) == 4;
    const int lr_c = lr, g_c = g;
    auto epilogue = [&](int m0, int n0) {
        // opaque per-tile copies of the lane coordinates: keeps the compiler from hoisting the 28
        // tiles' worth of lane-constant address arithmetic out of the persistent tile loop (which
        // costs > 100 VGPRs and spills the accumulators)
        int lr = lr_c, g = g_c;
        asm volatile("" : "+v"(lr), "+v"(g));
#if LSAM_ABL & 4
        {
            float sacc = 0.f;
            for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (sacc != 1.2345e-30f) return;
        }
#endif
        const int nc = n0 + wn * 112 + 4 * g;          // this lane's column in column tile 0
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            __builtin_amdgcn_sched_barrier(0);   // bound the scheduler's load hoisting: one row block at a time
            const int m = m0 + wm * 64 + mt * 16 + lr;
            const bool mok = m < M;
            const int mc = mok ? m : M - 1;   // clamp: every lane stays active for the cross-lane swaps
            auto pack2 = [](const f32x4& v) -> uint2 {
                bf16x4 o;
                o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
                return __builtin_bit_cast(uint2, o);
            };
            auto store_row = [&](auto* base, int ld, const f32x4 (&val)[NT]) {
                auto* rowp = base + (size_t)m * ld + n0 + wn * 112;
                if constexpr (sizeof(*base) == 4) {
                    if (mok) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) store4(rowp + nt * 16 + 4 * g, val[nt]);
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt + 1 < NT; nt += 2) {
                        const uint2 x = pack2(val[nt]), y = pack2(val[nt + 1]);
                        auto r0 = __builtin_amdgcn_permlane32_swap(x.x, y.x, false, false);
                        auto r1 = __builtin_amdgcn_permlane32_swap(x.y, y.y, false, false);
                        auto q0 = __builtin_amdgcn_permlane16_swap(r0[0], r0[1], false, false);
                        auto q1 = __builtin_amdgcn_permlane16_swap(r1[0], r1[1], false, false);
                        if (mok) *reinterpret_cast<uint4*>(rowp + nt * 16 + 8 * g) = make_uint4(q0[0], q1[0], q0[1], q1[1]);
                    }
                    if (mok) store4(rowp + (NT - 1) * 16 + 4 * g, val[NT - 1]);
                }
            };
            f32x4 rv[NT], v[NT];
            if (p.res) {
                const float* rp = p.res + (p.res_row_mod > 0 ? (size_t)(mc % p.res_row_mod) : (size_t)mc) * p.ldres + nc;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) rv[nt] = load4(rp + 16 * nt);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) v[nt] = acc[mt][nt];
            if (flags & LSAM_EPI_SAVE_PRE) store_row(static_cast<T*>(p.out2), p.ldo2, v);
            if (flags & LSAM_EPI_GELU) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[nt][r] = gelu_t<sizeof(T) == 2>(v[nt][r]);
            }
            if (flags & LSAM_EPI_GELU_BWD) {
                const T* xp = static_cast<const T*>(p.aux) + (size_t)mc * p.ldaux + nc;
                f32x4 x[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) x[nt] = load4(xp + 16 * nt);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[nt][r] *= gelu_grad_t<sizeof(T) == 2>(x[nt][r]);
            }
            if (p.res) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) v[nt] += rv[nt];
            }
            if constexpr (out_f32) store_row(static_cast<float*>(p.out), p.ldo, v);
            else store_row(static_cast<bf16*>(p.out), p.ldo, v);
        }
    };
    // POOL epilogue: out[pooled token][n] = max over the 2x2 token patch (the bias is already in the accumulators
    // and is the same for the four partners).  Even lanes store 16 bytes; rows = the pooled grid [N][H/2][W/2].
    auto epilogue_pool = [&](const TileAddr& ta) {
        int lr = lr_c, g = g_c;
        asm volatile("" : "+v"(lr), "+v"(g));
        const int nc = ta.n0 + wn * 112 + 4 * g;
        const int hh = p.pool_h >> 1, hw = p.pool_w >> 1;
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {
            const int py = (ta.py0 >> 1) + wm / pWX, px = (ta.px0 + 32 * (wm % pWX) + 16 * mp + lr) >> 1;
            float* rowp = static_cast<float*>(p.out) + (((size_t)ta.pn * hh + py) * hw + px) * p.ldo + nc;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = fmaxf(acc[mp][nt][r], acc[mp + 2][nt][r]);
                    const float b = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));   // lane ^ 1
                    v[r] = fmaxf(a, b);
                }
                if (!(lr & 1)) store4(rowp + 16 * nt, v);
            }
        }
    };
    // the accumulators start at the bias (one batched read per output tile) instead of zero
    auto init_acc = [&](int n0) {
        f32x4 bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = load4(p.bias + n0 + wn * 112 + 4 * g + 16 * nt);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = bv[j];
    };

    const int nkt = (K + BK - 1) / BK;
    constexpr bool HOIST = sizeof(T) == 2;
    frag af[HOIST ? KS : 1][HOIST ? MT : 1], wf[HOIST ? KS : 1][HOIST ? NT : 1];
    if constexpr (WN == 1) {
        for (int t = slot; t < ntiles; t += G) {
            const TileAddr ta = tile_addr(t);
            init_acc(ta.n0);
#pragma nounroll
            for (int kt = 0; kt < nkt; ++kt) {
                __syncthreads();           // every wave has finished reading the previous k-tile
                issue(ta, kt, 0);
                lds_dma_wait_barrier();    // DMA landed (vmcnt(0)) and visible to all waves
#if !(LSAM_ABL & 2)
                if constexpr (HOIST) {
                    load_frags(0, af, wf);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(af, wf);
                } else {
                    compute_interleaved(0);
                }
#endif
            }
            epilogue(ta.m0, ta.n0);
        }
    } else {
        int st = 0;                        // LDS stage holding the k-tile about to be consumed
        int t = slot;
        TileAddr ta = tile_addr(t < ntiles ? t : 0);
        if (t < ntiles) issue(ta, 0, 0);
        while (t < ntiles) {
            const int tn = t + G;
            const TileAddr tb = tile_addr(tn < ntiles ? tn : 0);
            init_acc(ta.n0);
#pragma nounroll
            for (int kt = 0; kt < nkt; ++kt) {
                lds_dma_wait_barrier();    // k-tile landed; every wave is done with the other stage
#if !(LSAM_ABL & 2)
                if constexpr (HOIST) load_frags(st, af, wf);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (HOIST && GEMM2_SPREAD_DMA) {
                    // the DMA instructions of the next k-tile are issued between groups of MFMAs (their issue cost
                    // falls into MFMA shadows) instead of in one burst in front of them.  No LDS read follows them
                    // before the next barrier, so hipcc has no reason to drain vmcnt early.
                    constexpr int NPIECE = A_IT + W_IT, NMMA = KS * NT * MT, PER = (NMMA + NPIECE - 1) / NPIECE;
                    const bool same = kt + 1 < nkt, nexttile = !same && tn < ntiles;
#pragma unroll
                    for (int c = 0; c < NPIECE; ++c) {
                        if (same) issue_piece(ta, kt + 1, st ^ 1, c);
                        else if (nexttile) issue_piece(tb, 0, st ^ 1, c);
#if !(LSAM_ABL & 2)
#pragma unroll
                        for (int e = c * PER; e < (c + 1) * PER && e < NMMA; ++e) {
                            const int ks = e / (NT * MT), nt = (e % (NT * MT)) / MT, mt = e % MT;
                            mma16(wf[ks][nt], af[ks][mt], acc[mt][nt]);
                        }
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    if (kt + 1 < nkt) issue(ta, kt + 1, st ^ 1);
                    else if (tn < ntiles) issue(tb, 0, st ^ 1);   // next output tile's first k-tile
                    __builtin_amdgcn_sched_barrier(0);
#if !(LSAM_ABL & 2)
                    if constexpr (HOIST) mfmas(af, wf);
                    else compute_interleaved(st);
#endif
                }
                st ^= 1;
            }
            if constexpr (POOL) epilogue_pool(ta);
            else epilogue(ta.m0, ta.n0);
            ta = tb;
            t = tn;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// v3 (bf16, N % 224 == 0): the 256 x 224 tile, DMA scheme and epilogues of v2 with the fragment reads
// software-pipelined ACROSS the per-k-tile barrier.  In v2 all eight waves leave the barrier together, read their
// 22 fragments and only then start their 56 MFMAs: the matrix pipes idle for the barrier plus the read latency of
// every k-tile (measured: the MFMA-only loop ran at 48 % of the MFMA rate).  Here the barrier sits between the two
// k-steps of a k-tile:
//     read ks1(q) | MFMA ks0(q) | wait DMA(q+1), barrier | read ks0(q+1), issue DMA(q+2) | MFMA ks1(q)
// so the first MFMAs behind a barrier have their operands in registers already, and every read has 28 MFMAs to
// land under.  The reads are inline asm (hipcc drains vmcnt(0) in front of any LDS read it can see behind an
// LDS-DMA, which would serialise copy and multiply) and are ordered by hand with counted waits.
// Further differences from v2:
//  * every wave issues exactly 8 DMA pieces per k-tile (the four spare ones go to the stage's padding rows), so
//    counted vmcnt waits are exact: behind an epilogue the wait for the next k-tile skips the epilogue's stores
//    (vmcnt counts in issue order), which then drain under the next tile's first k-tile instead of in front of it;
//  * the bias slice of an output tile rides with its first k-tile as one of those spare pieces (1 KiB of LDS by
//    tile parity) and seeds the accumulators as the C operand of the tile's first MFMAs -- no global read sits
//    between an epilogue's stores and the next tile's first MFMA;
//  * a last k-tile with K % 64 <= 32 skips its empty second k-step (K = 224, 480).
// ---------------------------------------------------------------------------------------------
#ifndef GEMM3
#define GEMM3 1
#endif
// Where a k-tile's 8 DMA pieces are issued: 0 = in one burst right behind the barrier, in front of the second MFMA
// block; 1 = in one burst behind that block; 2 = one piece in front of every 3-4 MFMAs of that block.  Stand-alone,
// 1 is 5-11 % faster than 0 on some shapes (16384 x 896 x 3584) and 6 % slower on others, and over the whole
// training step it loses (837 vs 851 frames/s); 2 is the best of the three there (852.6 vs 850.2, A/B/A/B on one box).
#ifndef GEMM3_DMA_POS
#define GEMM3_DMA_POS 2
#endif
typedef __attribute__((address_space(3))) const char* gemm_lds_cp;
template <int B, int E, typename F> __device__ __forceinline__ void g3_static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        g3_static_for<B + 1, E>(f);
    }
}
template <int OFF, typename V> __device__ __forceinline__ void g3_rd(unsigned addr, V& dst) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

// NGUARD: N need not be a multiple of 224 (Hiera-L widths, multiples of 144): W rows >= N are zero-filled by the DMA's
// range check and every epilogue access is guarded by its column.
// NT = column tiles per wave: 7 (224-wide output tiles) or 6 (192-wide: N = 576, 1152, 1728, 2304 ... of Hiera-L divide by 192
// and not by 224, so the narrower tile wastes no MFMAs on guard columns).
template <bool OUTF32, bool POOL, bool NGUARD = false, int NT = 7>
__global__ __launch_bounds__(512, 2) void gemm3_kernel(const GemmArgs p) {
    static_assert(NT == 7 || (NT == 6 && !NGUARD), "192-wide tiles: N % 192 == 0");
    using T = bf16;
    constexpr int MT = 4, BM = 256, BN = 32 * NT, WNC = 16 * NT, NW = 8, BK = 64, A_IT = 4;   // WNC: columns per wave
    constexpr int STAGE = 65536;       // 512 rows x 128 B; rows 480..511 only take the padding DMA pieces
    constexpr int BIAS0 = 2 * STAGE;   // 4 x 1 KiB: fp32 bias slice [224] of a tile, slot = tile count & 3

    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(gemm_lds_cp)(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int g = lane >> 4, lr = lane & 15;
    const int M = p.M, N = p.N, K = p.K;
    const int tiles_n = NGUARD ? (N + BN - 1) / BN : N / BN, tiles_m = (M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int slot = xcd_remap(blockIdx.x, G);
    if (slot >= ntiles) return;
    const T* __restrict__ A = static_cast<const T*>(p.A);
    const T* __restrict__ Wt = static_cast<const T*>(p.W);
    const int nkt = (K + BK - 1) / BK;
    const bool half_last = K - (nkt - 1) * BK <= 32;

    const int lrow = lane >> 3, ck = ((lane & 7) ^ lrow) * 8;
    const unsigned astep = 64u * (unsigned)p.lda * 2u, wstep = 64u * (unsigned)p.ldw * 2u;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(A), 0,
                                                         (int)(((size_t)(M - 1) * p.lda + K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Wt), 0,
                                                         (int)(((size_t)(N - 1) * p.ldw + K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias : reinterpret_cast<const float*>(A)), 0, p.bias ? N * 4 : 0, 0x00020000);

    // POOL: see gemm2_kernel
    const int pTW = POOL ? (p.pool_w < 128 ? p.pool_w : 128) : 1, pTH = POOL ? BM / pTW : 1, pWX = POOL ? pTW / 32 : 1;
    const int ptx = POOL ? p.pool_w / pTW : 1, pty = POOL ? p.pool_h / pTH : 1;
    struct DmaAddr { int n0, arow0, wrow0; unsigned aoff0, woff0; unsigned aoffp[POOL ? A_IT : 1]; };
    auto dma_addr = [&](int t) {
        DmaAddr a;
        const int m0 = (t / tiles_n) * BM;
        a.n0 = (t % tiles_n) * BN;
        a.arow0 = m0 + 8 * wave + lrow;
        a.wrow0 = a.n0 + 8 * wave + lrow;
        a.aoff0 = ((unsigned)a.arow0 * (unsigned)p.lda + ck) * 2u;
        a.woff0 = ((unsigned)a.wrow0 * (unsigned)p.ldw + ck) * 2u;
        if constexpr (POOL) {
            const int tm = t / tiles_n;
            const int pn = tm / (ptx * pty), py0 = ((tm / ptx) % pty) * pTH, px0 = (tm % ptx) * pTW;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int y = py0 + 2 * (i / pWX) + (wave >> 2), x = px0 + 32 * (i % pWX) + 16 * ((wave >> 1) & 1) + 8 * (wave & 1) + lrow;
                const unsigned tok = ((unsigned)pn * (unsigned)p.pool_h + (unsigned)y) * (unsigned)p.pool_w + (unsigned)x;
                a.aoffp[i] = (tok * (unsigned)p.lda + ck) * 2u;
            }
        }
        return a;
    };
    // 8 DMA wave-instructions: A rows 8 (wave + 8 i), W rows 8 (wave + 8 i); waves 4..7 have no fourth W piece: theirs
    // is a zero fill of the stage's padding rows, or (wave 4, first k-tile of an output tile) the tile's bias slice.
    auto issue = [&](const DmaAddr& ta, int kt, int stage, int bslot) {
#if LSAM_ABL & 1
        return;
#endif
        const int k0 = kt * BK;
        const bool kok = k0 + ck < K;
        const unsigned kb = (unsigned)k0 * 2u;
        char* base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            unsigned vo;
            if constexpr (POOL) vo = kok ? ta.aoffp[i] + kb : 0xFFFFFFFFu;
            else vo = (ta.arow0 + 64 * i < M && kok) ? ta.aoff0 + i * astep + kb : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_ptr_t)(base + (wave + NW * i) * 1024), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned vo = (kok && (!NGUARD || ta.wrow0 + 64 * i < N)) ? ta.woff0 + i * wstep + kb : 0xFFFFFFFFu;   // N % 224 == 0: every W row exists
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(base + (32 + wave + NW * i) * 1024), 16, vo, 0, 0, 0);
        }
        if (NT == 7 && wave < 4) {
            const unsigned vo = (kok && (!NGUARD || ta.wrow0 + 192 < N)) ? ta.woff0 + 3 * wstep + kb : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(base + (32 + wave + NW * 3) * 1024), 16, vo, 0, 0, 0);
        } else if (wave == 4 && kt == 0) {
            const unsigned vo = lane < 8 * NT ? (unsigned)(ta.n0 + 4 * lane) * 4u : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_ptr_t)(smem + BIAS0 + bslot * 1024), 16, vo, 0, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_ptr_t)(base + (32 + wave + NW * 3) * 1024), 16, 0xFFFFFFFFu, 0, 0, 0);
        }
    };

    // one of the 8 pieces (compile-time index), for the interleaved placement (GEMM3_DMA_POS == 2)
    auto issue_piece = [&](const DmaAddr& ta, int kt, int stage, int bslot, auto idx_) {
        constexpr int idx = decltype(idx_)::value;
        const int k0 = kt * BK;
        const bool kok = k0 + ck < K;
        const unsigned kb = (unsigned)k0 * 2u;
        char* base = smem + stage * STAGE;
        if constexpr (idx < A_IT) {
            unsigned vo;
            if constexpr (POOL) vo = kok ? ta.aoffp[idx] + kb : 0xFFFFFFFFu;
            else vo = (ta.arow0 + 64 * idx < M && kok) ? ta.aoff0 + idx * astep + kb : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_ptr_t)(base + (wave + NW * idx) * 1024), 16, vo, 0, 0, 0);
        } else if constexpr (idx < A_IT + 3) {
            constexpr int i = idx - A_IT;
            const unsigned vo = (kok && (!NGUARD || ta.wrow0 + 64 * i < N)) ? ta.woff0 + i * wstep + kb : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(base + (32 + wave + NW * i) * 1024), 16, vo, 0, 0, 0);
        } else {
            if (NT == 7 && wave < 4) {
                const unsigned vo = (kok && (!NGUARD || ta.wrow0 + 192 < N)) ? ta.woff0 + 3 * wstep + kb : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(base + (32 + wave + NW * 3) * 1024), 16, vo, 0, 0, 0);
            } else if (wave == 4 && kt == 0) {
                const unsigned vo = lane < 8 * NT ? (unsigned)(ta.n0 + 4 * lane) * 4u : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_ptr_t)(smem + BIAS0 + bslot * 1024), 16, vo, 0, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_ptr_t)(base + (32 + wave + NW * 3) * 1024), 16, 0xFFFFFFFFu, 0, 0, 0);
            }
        }
    };

    // per-lane fragment read addresses in stage 0 (k-step 1 is chunk ^ 4 of k-step 0)
    const unsigned chunk0 = (unsigned)((g ^ (lr & 7)) << 4);
    const unsigned aA0 = lds0 + (unsigned)(wm * 64 + lr) * 128u + chunk0, aA1 = aA0 ^ 64u;
    const unsigned aW0 = lds0 + (unsigned)(BM + wn * WNC + lr) * 128u + chunk0, aW1 = aW0 ^ 64u;

    // Fragments are held as 4 x u32 and re-typed at the MFMA: as <8 x bf16> values carried round the loop hipcc
    // re-packs some of them element by element (v_lshrrev / v_perm) at the loop head, i.e. BEFORE the counted wait
    // that makes the asm read's result valid.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[MT][NT];
    u32x4 af0[MT], wf0[NT], af1[MT], wf1[NT];
    auto rd_set = [&](unsigned aA, unsigned aW, u32x4 (&af)[MT], u32x4 (&wf)[NT]) {
        g3_static_for<0, MT>([&](auto i_) { g3_rd<decltype(i_)::value * 2048>(aA, af[decltype(i_)::value]); });
        g3_static_for<0, NT>([&](auto i_) { g3_rd<decltype(i_)::value * 2048>(aW, wf[decltype(i_)::value]); });
    };
    auto mfmas = [&](const u32x4 (&af)[MT], const u32x4 (&wf)[NT]) {
#if LSAM_ABL & 2
        return;
#endif
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                mma16(__builtin_bit_cast(bf16x8, wf[nt]), __builtin_bit_cast(bf16x8, af[mt]), acc[mt][nt]);
    };

    // ---- epilogues: as in gemm2_kernel, the bias is already in the accumulators ---------------------------
    const int flags = p.flags;
    const int lr_c = lr, g_c = g;
    auto epilogue = [&](int m0, int n0) {
        int lr = lr_c, g = g_c;
        asm volatile("" : "+v"(lr), "+v"(g));
        const int nc = n0 + wn * WNC + 4 * g;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            __builtin_amdgcn_sched_barrier(0);
            const int m = m0 + wm * 64 + mt * 16 + lr;
            const bool mok = m < M;
            const int mc = mok ? m : M - 1;
            // NGUARD: column c of this wave's 112 is inside the matrix while c < nlim (N % 8 == 0: a 16- or 8-byte store
            // is inside or outside as a whole)
            const int nlim = NGUARD ? N - (n0 + wn * WNC) : WNC;
            auto store_row = [&](auto* base, int ld, const f32x4 (&val)[NT]) {
                auto* rowp = base + (size_t)mc * ld + n0 + wn * WNC;
                if constexpr (sizeof(*base) == 4) {
                    if (mok) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            if (!NGUARD || nt * 16 + 4 * g < nlim) store4(rowp + nt * 16 + 4 * g, val[nt]);
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt + 1 < NT; nt += 2)
                        store_bf16_tile_pair(rowp + nt * 16, g, val[nt], val[nt + 1], mok && (!NGUARD || nt * 16 + 8 * g < nlim));
                    if constexpr (NT % 2 == 1) {
                        if (mok && (!NGUARD || (NT - 1) * 16 + 4 * g < nlim)) store4(rowp + (NT - 1) * 16 + 4 * g, val[NT - 1]);
                    }
                }
            };
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            f32x4 v[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) v[nt] = acc[mt][nt];
            if (flags & LSAM_EPI_SAVE_PRE) store_row(static_cast<T*>(p.out2), p.ldo2, v);
            if (flags & LSAM_EPI_GELU) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[nt][r] = gelu_t<true>(v[nt][r]);
            }
            if (flags & LSAM_EPI_GELU_BWD) {
                const T* xp = static_cast<const T*>(p.aux) + (size_t)mc * p.ldaux + nc;
                f32x4 x[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) x[nt] = (!NGUARD || nt * 16 + 4 * g < nlim) ? load4(xp + 16 * nt) : z4;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[nt][r] *= gelu_grad_t<true>(x[nt][r]);
            }
            if (p.res) {
                const float* rp = p.res + (p.res_row_mod > 0 ? (size_t)(mc % p.res_row_mod) : (size_t)mc) * p.ldres + nc;
                f32x4 rv[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) rv[nt] = (!NGUARD || nt * 16 + 4 * g < nlim) ? load4(rp + 16 * nt) : z4;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) v[nt] += rv[nt];
            }
            if constexpr (OUTF32) store_row(static_cast<float*>(p.out), p.ldo, v);
            else store_row(static_cast<bf16*>(p.out), p.ldo, v);
        }
    };
    auto epilogue_pool = [&](int t) {
        int lr = lr_c, g = g_c;
        asm volatile("" : "+v"(lr), "+v"(g));
        const int tm = t / tiles_n, n0 = (t % tiles_n) * BN;
        const int pn = tm / (ptx * pty), py0 = ((tm / ptx) % pty) * pTH, px0 = (tm % ptx) * pTW;
        const int nc = n0 + wn * WNC + 4 * g;
        const int hh = p.pool_h >> 1, hw = p.pool_w >> 1;
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {
            const int py = (py0 >> 1) + wm / pWX, px = (px0 + 32 * (wm % pWX) + 16 * mp + lr) >> 1;
            const bool second = p.split_n > 0 && n0 >= p.split_n;   // workgroup-uniform: a column tile belongs to one output
            float* rowp = second ? static_cast<float*>(p.out2) + (((size_t)pn * hh + py) * hw + px) * p.ldo2 + (nc - p.split_n)
                                 : static_cast<float*>(p.out) + (((size_t)pn * hh + py) * hw + px) * p.ldo + nc;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = fmaxf(acc[mp][nt][r], acc[mp + 2][nt][r]);
                    const float b = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));
                    v[r] = fmaxf(a, b);
                }
                if (!(lr & 1) && (!NGUARD || nc + 16 * nt < N)) store4(rowp + 16 * nt, v);   // N % 8 == 0: whole 16-byte groups
            }
        }
    };

    // ---- DMA cursor: two k-tiles ahead of the multiply -------------------------------------------------------
    int dt = slot, dkt = 0, dstage = 0, dit = 0;
    DmaAddr da = dma_addr(dt);
    auto dma_advance = [&]() {
        dstage ^= 1;
        if (++dkt == nkt) {
            dkt = 0;
            dt += G;
            ++dit;
            if (dt < ntiles) da = dma_addr(dt);
        }
    };
    auto dma_next = [&]() {
        if (dt < ntiles) issue(da, dkt, dstage, dit & 3);
        dstage ^= 1;
        if (++dkt == nkt) {
            dkt = 0;
            dt += G;
            ++dit;
            if (dt < ntiles) da = dma_addr(dt);
        }
    };
    dma_next();
    const bool second = dt < ntiles;
    dma_next();
    if (second) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    rd_set(aA0, aW0, af0, wf0);

    int st = 0, it = 0;
    int after = 0;   // vector-memory operations this wave issued behind the DMA the next barrier waits for
    for (int t = slot; t < ntiles; t += G, ++it) {
#pragma nounroll
        for (int kt = 0; kt < nkt; ++kt) {
            const unsigned so = (unsigned)st << 16, sn = so ^ 0x10000u;
            __builtin_amdgcn_sched_barrier(0);
            if (kt == 0) {
                // the bias slice lands in row block 0's accumulators and is the C operand of the tile's first
                // MFMAs (row block 0 last: it overwrites the bias)
                // (address rebuilt from the lane id here: kept live across the tile loop it costs a spill whose reload
                // would put a vmcnt(0) in front of every tile)
                unsigned ln;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
                const unsigned ab = lds0 + BIAS0 + (unsigned)(wn * WNC) * 4u + (ln >> 4) * 16u + (unsigned)(it & 3) * 1024u;
                g3_static_for<0, NT>([&](auto i_) { g3_rd<decltype(i_)::value * 64>(ab, acc[0][decltype(i_)::value]); });
                rd_set(aA1 + so, aW1 + so, af1, wf1);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT) : "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int mt = 1; mt < MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf0[nt]),
                                                                              __builtin_bit_cast(bf16x8, af0[mt]), acc[0][nt], 0, 0, 0);
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf0[nt]),
                                                                         __builtin_bit_cast(bf16x8, af0[0]), acc[0][nt], 0, 0, 0);
                }
            } else {
                rd_set(aA1 + so, aW1 + so, af1, wf1);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                mfmas(af0, wf0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // the next k-tile has landed (this wave's pieces; the barrier covers the others') and every wave is
            // done reading this stage
            if (kt == 0 && after >= 24) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (kt == 0 && after >= 20) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (kt == 0 && after >= 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (kt == 0 && after >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            rd_set(aA0 + sn, aW0 + sn, af0, wf0);
            __builtin_amdgcn_sched_barrier(0);
#if GEMM3_DMA_POS == 0
            dma_next();
            __builtin_amdgcn_sched_barrier(0);
#endif
#if GEMM3_DMA_POS == 2
            {
                const bool live = dt < ntiles, mm = !(half_last && kt == nkt - 1);
                g3_static_for<0, 8>([&](auto c_) {
                    constexpr int c = decltype(c_)::value;
                    if (live) issue_piece(da, dkt, dstage, dit & 3, c_);
                    if (mm) {
#pragma unroll
                        for (int e = (MT * NT * c) / 8; e < (MT * NT * (c + 1)) / 8; ++e) {
                            const int nt = e / MT, mt = e % MT;
                            mma16(__builtin_bit_cast(bf16x8, wf1[nt]), __builtin_bit_cast(bf16x8, af1[mt]), acc[mt][nt]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                dma_advance();
            }
#else
            if (!(half_last && kt == nkt - 1)) mfmas(af1, wf1);
#endif
#if GEMM3_DMA_POS == 1
            __builtin_amdgcn_sched_barrier(0);
            dma_next();
#endif
            st ^= 1;
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next tile's first fragments are in registers
        __builtin_amdgcn_sched_barrier(0);
        const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
#if LSAM_ABL & 4
        if (acc[0][0][0] == 1.2345e-30f && acc[3][6][3] == 1.2345e-30f) epilogue(m0, n0);
        after = 0;
        continue;
#endif
        if constexpr (POOL) {
            epilogue_pool(t);
            // 14 stores (NT = 6: 12), counted a little short so that a store the compiler fuses or drops cannot matter; a
            // column-guarded last tile issues fewer: wait for everything
            after = (!NGUARD || (t % tiles_n) * BN + BN <= N) ? (NT == 7 ? 12 : 8) : 0;
        } else {
            epilogue(m0, n0);
            // full tile: 28 (fp32) / 16 (bf16) stores (NT = 6: 24 / 12), counted short; a partial tile issues fewer: wait for everything
            after = (m0 + BM <= M && (!NGUARD || n0 + BN <= N)) ? (NT == 7 ? (OUTF32 ? 24 : 12) : (OUTF32 ? 20 : 8)) : 0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
            LSAM_HIP(hipEventCreateWithFlags(&g_prof[i].e0, hipEventDisableSystemFence));
            LSAM_HIP(hipEventCreateWithFlags(&g_prof[i].e1, hipEventDisableSystemFence));
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

// generic form for the non-GEMM MFMA kernels (tailblock.hip): returns a record index or -1
int prof_begin(hipStream_t s) {
    if (!(g_prof_on && g_prof_n < g_prof_cap)) return -1;
    if (hipEventRecord(g_prof[g_prof_n].e0, s) != hipSuccess) return -1;
    return g_prof_n;
}
void prof_end(int rec, hipStream_t s, double flops, int cfg, int m, int n, int k) {
    if (rec < 0 || rec != g_prof_n) return;
    if (hipEventRecord(g_prof[rec].e1, s) != hipSuccess) return;
    g_prof[rec].flops = flops;
    g_prof[rec].cfg = cfg;
    g_prof[rec].m = m; g_prof[rec].n = n; g_prof[rec].k = k;
    ++g_prof_n;
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

template <typename T, int WN, bool OUTF32, bool POOL = false> static int launch_v2(const GemmArgs& a, hipStream_t s) {
    constexpr int LDS = WN * (256 + 112 * WN) * 128;
    auto kern = gemm2_kernel<T, WN, OUTF32, POOL>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int ntiles = cdiv(a.M, 256) * cdiv(a.N, 112 * WN);
    constexpr int PER_CU = WN == 1 ? 2 : 1;
    const int nblk = ntiles < 256 * PER_CU ? ntiles : 256 * PER_CU;   // persistent: one wave of workgroups
    const bool prof = g_prof_on && g_prof_n < g_prof_cap;
    if (prof) LSAM_HIP(hipEventRecord(g_prof[g_prof_n].e0, s));
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256 * WN), LDS, s, a);
    LSAM_LAUNCH_CHECK();
    if (prof) {
        LSAM_HIP(hipEventRecord(g_prof[g_prof_n].e1, s));
        g_prof[g_prof_n].flops = 2.0 * a.M * a.N * a.K;
        g_prof[g_prof_n].cfg = (int)sizeof(T) * 1000 + 112 * WN + (OUTF32 && sizeof(T) == 2 ? 1 : 0);
        g_prof[g_prof_n].m = a.M; g_prof[g_prof_n].n = a.N; g_prof[g_prof_n].k = a.K;
        ++g_prof_n;
    }
    return LSAM_OK;
}

template <bool OUTF32, bool POOL = false, bool NGUARD = false, int NT = 7> static int launch_v3(const GemmArgs& a, hipStream_t s) {
    constexpr int LDS = 2 * 65536 + 4096;
    auto kern = gemm3_kernel<OUTF32, POOL, NGUARD, NT>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int ntiles = cdiv(a.M, 256) * cdiv(a.N, 32 * NT);
    const int nblk = ntiles < 256 ? ntiles : 256;   // persistent: one workgroup per CU
    const int rec = prof_begin(s);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), LDS, s, a);
    LSAM_LAUNCH_CHECK();
    prof_end(rec, s, 2.0 * a.M * a.N * a.K, 2000 + 224 + (OUTF32 ? 1 : 0) + (NGUARD || NT == 6 ? 2 : 0), a.M, a.N, a.K);
    return LSAM_OK;
}

// ---------------------------------------------------------------------------------------------
// Skinny products out[M][32] = A[M][K] . W[32][K]^T (the LoRA u = xn . [A_q | A_v]^T and t = dqkv . [B_q | B_v] columns,
// lora_qkv.py:104-105 and their adjoints): 2.8 GFLOP over an 88 MB operand is a streaming read, and a 128-wide tile
// kernel runs it on 128 workgroups with three quarters of every tile empty (55 us).  Here a workgroup owns 16 rows, its
// four waves split K, every operand fragment is a lane's own 16-byte global load (no LDS staging, no barrier in the
// loop), and the four partial tiles meet in LDS once.  Swapped product: a lane ends with 4 consecutive outputs of one row.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void skinny32_kernel(const GemmArgs p) {
    using frag = typename Tr<T>::frag;
    constexpr int EPC = Tr<T>::EPC, CPF = Tr<T>::CPF, UNR = sizeof(T) == 2 ? 7 : 4;
    __shared__ __attribute__((aligned(16))) float part[3][2][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, lr = lane & 15;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 32;
    const T* __restrict__ A = static_cast<const T*>(p.A) + (size_t)min(m0 + lr, p.M - 1) * p.lda;   // clamped rows are never stored
    const T* __restrict__ W0 = static_cast<const T*>(p.W) + (size_t)min(n0 + lr, p.N - 1) * p.ldw;
    const T* __restrict__ W1 = static_cast<const T*>(p.W) + (size_t)min(n0 + 16 + lr, p.N - 1) * p.ldw;
    const int nks = (p.K + 31) / 32, per = (nks + 3) / 4;
    const int ks0 = wave * per, ks1 = min(nks, ks0 + per);
    auto ldf = [&](const T* row, int ks) -> frag {
        const int k0 = 32 * ks + 8 * g;
        uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
        if (k0 < p.K) {
            c0 = *reinterpret_cast<const uint4*>(row + k0);
            if constexpr (CPF == 2) {   // fp32: K % 4 == 0 only, so the fragment's second chunk has a guard of its own
                if (k0 + EPC < p.K) c1 = *reinterpret_cast<const uint4*>(row + k0 + EPC);
            }
        }
        return make_frag(c0, c1, T{});
    };
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    for (int ks = ks0; ks < ks1; ks += UNR) {
        frag a[UNR], w0[UNR], w1[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {   // all loads of the group requested before the first product
            const int k = min(ks + u, nks - 1);
            a[u] = ldf(A, k);
            w0[u] = ldf(W0, k);
            w1[u] = ldf(W1, k);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (ks + u < ks1) {
                mma16(w0[u], a[u], acc0);
                mma16(w1[u], a[u], acc1);
            }
    }
    if (wave > 0) {
        *reinterpret_cast<f32x4*>(part[wave - 1][0][lane]) = acc0;
        *reinterpret_cast<f32x4*>(part[wave - 1][1][lane]) = acc1;
    }
    __syncthreads();
    if (wave == 0 && m0 + lr < p.M) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            acc0 += *reinterpret_cast<const f32x4*>(part[w][0][lane]);
            acc1 += *reinterpret_cast<const f32x4*>(part[w][1][lane]);
        }
        // lane: row m0 + lr, columns n0 + 4 g .. + 3 (acc0) and n0 + 16 + 4 g .. (acc1); N % 4 == 0
        const int m = m0 + lr;
        auto finish = [&](f32x4 v, int n) {
            if (n >= p.N) return;
            if (p.bias) v += load4(p.bias + n);
            if (p.flags & LSAM_EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (p.flags & LSAM_EPI_SIGMOID) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = 1.0f / (1.0f + __expf(-v[r]));
            }
            if (p.res) v += load4(p.res + (size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldres + n);
            if ((p.flags & LSAM_EPI_OUT_F32) || sizeof(T) == 4) store4(static_cast<float*>(p.out) + (size_t)m * p.ldo + n, v);
            else store4(static_cast<T*>(p.out) + (size_t)m * p.ldo + n, v);
        };
        finish(acc0, n0 + 4 * g);
        finish(acc1, n0 + 16 + 4 * g);
    }
}

template <typename T> static int launch_skinny32(const GemmArgs& a, hipStream_t s) {
    const int rec = prof_begin(s);
    hipLaunchKernelGGL(skinny32_kernel<T>, dim3(cdiv(a.M, 16), cdiv(a.N, 32)), dim3(256), 0, s, a);
    LSAM_LAUNCH_CHECK();
    prof_end(rec, s, 2.0 * a.M * a.N * a.K, (int)sizeof(T) * 1000 + 32, a.M, a.N, a.K);
    return LSAM_OK;
}

// pooled epilogue: whole TH x TW patches (TW = min(W, 128) a multiple of 32, TH = 256 / TW), 224-wide column tiles
bool gemm_pool_supported(int H, int W, int N) {
    // N % 224 == 0: full 224-wide tiles; N % 192 == 0: 192-wide ones; any other N >= 144 with N % 8 == 0: column-guarded
    if (H <= 0 || W <= 0 || W % 32 != 0 || N % 8 != 0 || N < 144) return false;
    const int tw = W < 128 ? W : 128;
    if (tw != 32 && tw != 64 && tw != 128) return false;
    return W % tw == 0 && H % (256 / tw) == 0;
}

template <typename T> int launch_gemm(const GemmArgs& a_in, hipStream_t s) {
    constexpr int EPC = Tr<T>::EPC;
    const bool few_rows = (a_in.flags & LSAM_EPI_FEW_ROWS) != 0;
    GemmArgs a = a_in;
    a.flags &= ~LSAM_EPI_FEW_ROWS;
    LSAM_REQUIRE(a.A && a.W && a.out, LSAM_ERR_BAD_ARG, "linear: null operand");
    LSAM_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, LSAM_ERR_BAD_ARG, "linear: empty shape %dx%dx%d", a.M, a.N, a.K);
    LSAM_REQUIRE(a.N % 8 == 0 && a.K % EPC == 0, LSAM_ERR_UNSUPPORTED,
                 "linear: need N%%8==0 and K%%%d==0 (N=%d K=%d)", EPC, a.N, a.K);
    LSAM_REQUIRE(a.lda % EPC == 0 && a.ldw % EPC == 0 && aligned16(a.A) && aligned16(a.W),
                 LSAM_ERR_BAD_ARG, "linear: operand rows must be 16-byte aligned");
    LSAM_REQUIRE(a.lda >= a.K && a.ldw >= a.K && a.ldo >= (a.split_n > 0 ? a.split_n : a.N), LSAM_ERR_BAD_ARG, "linear: bad leading dimension");
    LSAM_REQUIRE(a.split_n == 0 || a.pool_w > 0, LSAM_ERR_BAD_ARG, "linear: split_n goes with the pooled epilogue only");
    LSAM_REQUIRE(a.ldo % 4 == 0 && aligned16(a.out) && (!a.bias || aligned16(a.bias)), LSAM_ERR_BAD_ARG,
                 "linear: out/bias alignment");
    LSAM_REQUIRE(!a.res || (a.ldres % 4 == 0 && aligned16(a.res)), LSAM_ERR_BAD_ARG, "linear: res alignment");
    LSAM_REQUIRE(!(a.flags & LSAM_EPI_GELU_BWD) || (a.aux && a.ldaux % 4 == 0), LSAM_ERR_BAD_ARG,
                 "linear: GELU_BWD needs aux");
    LSAM_REQUIRE(!(a.flags & LSAM_EPI_SAVE_PRE) || (a.out2 && a.ldo2 % 4 == 0), LSAM_ERR_BAD_ARG,
                 "linear: SAVE_PRE needs out2");
    // The LDS-DMA kernels address A with 32-bit buffer offsets: an operand of 2 GB or more (Hiera-L at 64 frames: 4.2 M
    // tokens x 576 channels) is cut into row slices, one launch each, instead of falling back to the slow generic kernel.
    if (a.pool_w == 0 && a.res_row_mod == 0 && a.M >= 2048 && (double)a.M * a.lda * sizeof(T) >= 2.0e9 &&
        (double)a.N * a.ldw * sizeof(T) < 2.0e9) {
        long rows = (long)(1.9e9 / ((double)a.lda * sizeof(T)));
        rows -= rows % 256;
        if (rows >= 1024) {
            const bool of32 = (a.flags & LSAM_EPI_OUT_F32) || sizeof(T) == 4;
            for (long m0 = 0; m0 < a.M; m0 += rows) {
                GemmArgs b = a;
                b.M = (int)((a.M - m0) < rows ? (a.M - m0) : rows);
                b.A = static_cast<const T*>(a.A) + (size_t)m0 * a.lda;
                if (a.res) b.res = a.res + (size_t)m0 * a.ldres;
                if (a.aux) b.aux = static_cast<const T*>(a.aux) + (size_t)m0 * a.ldaux;
                if (a.out2) b.out2 = static_cast<T*>(a.out2) + (size_t)m0 * a.ldo2;
                b.out = of32 ? static_cast<void*>(static_cast<float*>(a.out) + (size_t)m0 * a.ldo)
                             : static_cast<void*>(static_cast<T*>(a.out) + (size_t)m0 * a.ldo);
                LSAM_TRY(launch_gemm<T>(b, s));
            }
            return LSAM_OK;
        }
    }
    // LSAM_EPI_FEW_ROWS (the SAM heads' token stream: 8 tokens per frame; the caller says so, the choice never depends on M, so
    // results do not depend on the batch size) or N = 32 (the LoRA u / t columns): the streaming kernel -- 16 x 32 outputs per
    // workgroup, K split over its waves, no LDS staging -- instead of a tile kernel's serial k-loop on one or two workgroups
    const bool stream_ok = a.pool_w == 0 && !a.aux && !a.out2 && (a.flags & ~(LSAM_EPI_OUT_F32 | LSAM_EPI_RELU | LSAM_EPI_SIGMOID)) == 0;
    LSAM_REQUIRE(!few_rows || stream_ok, LSAM_ERR_UNSUPPORTED, "linear: FEW_ROWS goes with bias / ReLU / sigmoid / OUT_F32 / res only");
    if (few_rows || (stream_ok && a.N == 32 && a.M >= 1024 && !a.bias && !a.res && (a.flags & ~LSAM_EPI_OUT_F32) == 0))
        return launch_skinny32<T>(a, s);
    // ReLU / sigmoid epilogues (the SAM heads' small MLPs) exist in the generic kernel only
    if (a.flags & (LSAM_EPI_RELU | LSAM_EPI_SIGMOID)) {
        LSAM_REQUIRE(a.pool_w == 0, LSAM_ERR_UNSUPPORTED, "linear: ReLU / sigmoid with the pooled epilogue");
        return a.N % 112 == 0 ? launch_cfg<T, 2, 7, 4, 1>(a, s) : launch_cfg<T, 4, 4, 2, 2>(a, s);
    }
    if (a.pool_w > 0) {
        if constexpr (sizeof(T) == 2) {
            LSAM_REQUIRE(a.split_n == 0 || (GEMM3 && a.split_n % 224 == 0 && a.split_n < a.N && a.out2 && a.ldo2 % 4 == 0 &&
                                            a.ldo2 >= a.N - a.split_n && a.ldo >= a.split_n && aligned16(a.out2)),
                         LSAM_ERR_BAD_ARG, "linear: bad split of the pooled output");
            LSAM_REQUIRE(gemm_pool_supported(a.pool_h, a.pool_w, a.N) && a.M % (a.pool_h * a.pool_w) == 0 &&
                             (a.flags & ~LSAM_EPI_OUT_F32) == 0 && (a.flags & LSAM_EPI_OUT_F32) && !a.res &&
                             (double)a.M * a.lda * sizeof(T) < 2.0e9 && (double)a.N * a.ldw * sizeof(T) < 2.0e9,
                         LSAM_ERR_UNSUPPORTED, "linear: pooled epilogue not available for this shape");
            if (GEMM3 && a.N % 224 != 0) {   // Hiera-L widths (288, 576, 1152)
                LSAM_REQUIRE(a.split_n == 0, LSAM_ERR_UNSUPPORTED, "linear: the split pooled output needs N %% 224 == 0");
                if (a.N % 192 == 0) return launch_v3<true, true, false, 6>(a, s);
                return launch_v3<true, true, true>(a, s);
            }
            LSAM_REQUIRE(a.N % 224 == 0, LSAM_ERR_UNSUPPORTED, "linear: pooled epilogue needs N %% 224 == 0 in this build");
            if (GEMM3) return launch_v3<true, true>(a, s);
            return launch_v2<T, 2, true, true>(a, s);
        } else {
            set_error("linear: pooled epilogue is bf16 only");
            return LSAM_ERR_UNSUPPORTED;
        }
    }
    if constexpr (sizeof(T) == 2) {
        // widths that are not multiples of 112 (Hiera-T/S/L: multiples of 96 / 144): the 256 x 224 kernel with column guards
        // once a tile row is reasonably full
        if (GEMM3 && a.N % 112 != 0 && a.N >= 144 && a.M >= 1024 &&
            (double)a.M * a.lda * sizeof(T) < 2.0e9 && (double)a.N * a.ldw * sizeof(T) < 2.0e9 && a.ldo % 8 == 0 &&
            (!(a.flags & LSAM_EPI_SAVE_PRE) || (a.ldo2 % 8 == 0 && aligned16(a.out2))) &&
            (!(a.flags & LSAM_EPI_GELU_BWD) || a.ldaux % 4 == 0) && (!a.res || a.ldres % 4 == 0)) {
            const bool of32 = (a.flags & LSAM_EPI_OUT_F32) != 0;
#ifndef GEMM3_NT6
#define GEMM3_NT6 1
#endif
            // N a multiple of 192 (Hiera-L stages 3 / 4: 576, 1152, 1728, 2304, 3456, 4608): 192-wide tiles, no guard columns
            // (unless the 224-wide tiling wastes under 3 % of its columns: 4608 -> 2.1 %, where the wider tile is 2 % faster)
            if (GEMM3_NT6 && a.N % 192 == 0 && (long)(cdiv(a.N, 224) * 224 - a.N) * 100 >= 3L * a.N) return of32 ? launch_v3<true, false, false, 6>(a, s) : launch_v3<false, false, false, 6>(a, s);
            return of32 ? launch_v3<true, false, true>(a, s) : launch_v3<false, false, true>(a, s);
        }
    }
    if (a.N % 112 == 0 && (double)a.M * a.lda * sizeof(T) < 2.0e9 && (double)a.N * a.ldw * sizeof(T) < 2.0e9 && a.M >= 1024 &&
        a.ldo % 8 == 0 && (!(a.flags & LSAM_EPI_SAVE_PRE) || (a.ldo2 % 8 == 0 && aligned16(a.out2))))
    {
        const bool of32 = (a.flags & LSAM_EPI_OUT_F32) || sizeof(T) == 4;
#ifndef GEMM2_FORCE_WN1
#define GEMM2_FORCE_WN1 0
#endif
        if constexpr (sizeof(T) == 2) {
            if (GEMM3 && a.N % 224 == 0 && !GEMM2_FORCE_WN1) return of32 ? launch_v3<true>(a, s) : launch_v3<false>(a, s);
        }
        if (a.N % 224 == 0 && !GEMM2_FORCE_WN1) return of32 ? launch_v2<T, 2, true>(a, s) : launch_v2<T, 2, sizeof(T) == 4>(a, s);
        return of32 ? launch_v2<T, 1, true>(a, s) : launch_v2<T, 1, sizeof(T) == 4>(a, s);
    }
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
