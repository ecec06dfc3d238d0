// Fused MLP half of a MultiScaleBlock (hieradet.py:165, sam2_utils.py:112-136) for the narrow,
// HBM-bound stages:   y = x + fc2(GELU(fc1(LN2(x))))   in ONE kernel: x is read once and y written
// once; LN2's output, the 4d-wide hidden activation and its GELU never exist in HBM.
//
// A wave owns 16*MT consecutive tokens and keeps all of their activations in registers as MFMA
// operands.  The hidden dimension is walked in chunks of 64:
//   acc1 = W1[chunk] . xn^T            (swapped product: lane = token, 4 consecutive hidden units)
//   h    = GELU(acc1 + b1)             -> bf16, which IS the B-operand layout of the next product once
//                                         the K order of W2 is permuted to match (done when packing)
//   acc2 += W2[:, chunk] . h^T         (lane = token, 4 consecutive output channels)
// The weight chunks (W1 rows + W2 columns + the chunk's fc1 biases, 30-60 KiB) are packed once into the exact LDS image and
// streamed global -> LDS by LDS-DMA, double-buffered: all waves of the workgroup walk the chunks in
// lock-step (one barrier per chunk) and the DMA of chunk c+1 lands while chunk c is multiplied.
#include <type_traits>

#include "kernels.hpp"

namespace lsam {

#ifndef MLP_ABL
#define MLP_ABL 0
#endif
#ifndef MLP_OPAQUE_LANE_ALL
#define MLP_OPAQUE_LANE_ALL 0
#endif
#ifndef MLP_PROJ_LDS
#define MLP_PROJ_LDS 1   // PROJ: W_proj as one more chunk of the LDS stream (1) or read from L2 as fragments (0: 595 us against 524, 36 spills)
#endif
typedef __attribute__((address_space(3))) void* mlp_lds_p;
typedef const __attribute__((address_space(1))) void* mlp_gbl_p;
typedef __attribute__((address_space(3))) const char* mlp_lds_cp;

namespace {
template <int B, int E, typename F> __device__ __forceinline__ void mlp_static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        mlp_static_for<B + 1, E>(f);
    }
}
// LDS reads the compiler cannot see (it waits vmcnt(0), i.e. for the chunk DMA just issued, in front of every
// LDS read it can see after an LDS-DMA); ordered by hand with mlp_lds_wait() and scheduling barriers.
template <int OFF> __device__ __forceinline__ void mlp_rd(unsigned addr, bf16x8& dst) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void mlp_rd(unsigned addr, f32x4& dst) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void mlp_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
}  // namespace

template <int KS, int NTO> struct MlpImg {
    static constexpr int NSUB = (KS + 1) / 2;                  // [64][64] sub-tiles of a W1 chunk
    static constexpr int W1_B = NSUB * 64 * 128;
    static constexpr int W2_B = NTO * 16 * 128;
    static constexpr int B1_OFF = W1_B + W2_B;                 // the chunk's 64 fc1 biases (fp32), padded to 1 KiB
    static constexpr int CHUNK_B = W1_B + W2_B + 1024;         // multiple of 1024
};

__host__ __device__ inline int mlp_slot_to_ch(int s) {
    const int ks = s >> 5, g = (s >> 3) & 3, t = (s >> 2) & 1, r = s & 3;
    return 32 * ks + 16 * t + 4 * g + r;
}

template <int KS, int NTO>
__global__ void pack_mlpblock_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                     const float* __restrict__ w2, const float* __restrict__ b2, int d, int nch,
                                     char* __restrict__ img, float* __restrict__ bias_p) {
    using I = MlpImg<KS, NTO>;
    const int hid = 4 * d;
    const long total = (long)nch * I::CHUNK_B / 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long byte = 2 * i;
        const int c = (int)(byte / I::CHUNK_B), in = (int)(byte % I::CHUNK_B);
        float v = 0.f;
        if (in >= I::B1_OFF) {   // fp32 biases are written by the loop below
            continue;
        } else if (in < I::W1_B) {
            const int kk = in / 8192, r8 = in % 8192;
            // K order of fc1 = the accumulator layout of the residual tile (see the kernel): k-slot s of a
            // 64-wide sub-tile holds channel mlp_slot_to_ch(s)
            const int row = r8 / 128, chunk = ((r8 % 128) >> 4) ^ (row & 7), k = 64 * kk + mlp_slot_to_ch(8 * chunk + ((r8 & 15) >> 1));
            const int hrow = c * 64 + row;
            if (hrow < hid && k < d) v = w1[(size_t)hrow * d + k];
        } else {
            const int r8 = in - I::W1_B;
            const int row = r8 / 128, chunk = ((r8 % 128) >> 4) ^ (row & 7), slot = 8 * chunk + ((r8 & 15) >> 1);
            const int hcol = c * 64 + mlp_slot_to_ch(slot);
            if (row < d && hcol < hid) v = w2[(size_t)row * hid + hcol];
        }
        reinterpret_cast<bf16*>(img)[i] = (bf16)v;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nch * 256; i += gridDim.x * blockDim.x) {
        const int c = i / 256, e = i % 256;   // 256 floats = the chunk's 1 KiB bias slot
        reinterpret_cast<float*>(img + (size_t)c * I::CHUNK_B + I::B1_OFF)[e] = (e < 64 && c * 64 + e < hid) ? b1[c * 64 + e] : 0.f;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NTO * 16; i += gridDim.x * blockDim.x)
        bias_p[i] = i < d ? b2[i] : 0.f;
}

// PROJ: the attention projection in front -- y = x + b_proj + a . W_proj^T takes the place of x (hieradet.py:79,163).  W_proj is
// one more chunk of the weight stream (fragment order, NTO x KS KiB: fits a chunk buffer for the 64- and 112-wide variants),
// consumed between the x loads and LayerNorm 2; the stage-1 attention kernel then stops at its attention output (winblock.hip,
// ATTN_ONLY), whose proj + residual phase cost it a third of its time at one wave per SIMD, while this kernel is VALU-bound.
// W_proj as one chunk: fragment (nt, ks) at (nt KS + ks) KiB, lane (g, lr) = row 16 nt + lr, k = 32 ks + 8 g + j (natural K order:
// the attention rows are read as they lie in memory); the rest of the chunk is zero.  pbias: b_proj padded to 16 NTO.
template <int KS, int NTO>
__global__ void pack_mlpproj_kernel(const float* __restrict__ wproj, const float* __restrict__ bproj, int d, char* __restrict__ img,
                                    float* __restrict__ pbias) {
    constexpr int CB = MlpImg<KS, NTO>::CHUNK_B;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < CB / 2; i += gridDim.x * blockDim.x) {
        const int f = i >> 9, lane = (i & 511) >> 3, j = i & 7;
        float v = 0.f;
        if (f < NTO * KS) {
            const int nt = f / KS, ks = f % KS, row = 16 * nt + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + j;
            if (row < d && k < d) v = wproj[(size_t)row * d + k];
        }
        reinterpret_cast<bf16*>(img)[i] = (bf16)v;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 16 * NTO; i += gridDim.x * blockDim.x) pbias[i] = i < d ? bproj[i] : 0.f;
}

template <int KS, int NTO, int MT, int NW, bool PROJ = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void mlpblock_kernel(const MlpBlockArgs p) {
    static_assert(!PROJ || NTO * KS * 1024 <= MlpImg<KS, NTO>::CHUNK_B, "the proj fragments must fit a chunk buffer");
    using I = MlpImg<KS, NTO>;
    constexpr int CB = I::CHUNK_B, TPW = 16 * MT, TPG = TPW * NW;
    constexpr int NDMA = CB / 1024;   // 1 KiB DMA wave-instructions per chunk
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lr = lane & 15;
    const int d = p.d, M = p.M, nch = p.nch;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ b2p = p.bias_p;
    const char* __restrict__ img = static_cast<const char*>(p.img);

    constexpr bool OPQ = PROJ || MLP_OPAQUE_LANE_ALL;
    auto issue = [&](int c, int buf) {
#if MLP_ABL & 1
        return;
#endif
        const char* src = img + (size_t)c * CB;
        char* dst = smem + buf * CB;
        // PROJ: the lane's byte offset is rebuilt here from an opaque lane id -- kept live across the token loop the per-piece
        // source addresses are spilled, and a scratch reload waits for vmcnt(0), i.e. for the DMA piece issued just before it
        unsigned lo = 0;
        if constexpr (OPQ) {
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
            lo *= 16;
        }
#pragma unroll
        for (int i = 0; i < (NDMA + NW - 1) / NW; ++i) {
            const int j = wave + NW * i;
            if (j < NDMA) {
                if constexpr (OPQ) __builtin_amdgcn_global_load_lds((mlp_gbl_p)(src + j * 1024 + lo), (mlp_lds_p)(dst + j * 1024), 16, 0, 0);
                else __builtin_amdgcn_global_load_lds((mlp_gbl_p)(src + j * 1024 + lane * 16), (mlp_lds_p)(dst + j * 1024), 16, 0, 0);
            }
        }
    };
    auto issue_proj = [&](int buf) {   // the W_proj chunk (PROJ): same size, its own image
        const char* src = static_cast<const char*>(p.pimg);
        char* dst = smem + buf * CB;
        unsigned lo;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
        lo *= 16;
#pragma unroll
        for (int i = 0; i < (NDMA + NW - 1) / NW; ++i) {
            const int j = wave + NW * i;
            if (j < NDMA)
                __builtin_amdgcn_global_load_lds((mlp_gbl_p)(src + j * 1024 + lo), (mlp_lds_p)(dst + j * 1024), 16, 0, 0);
        }
    };
    // chunk after MLP chunk c in stream order: c + 1, or the first chunk of the next token group (W_proj under PROJ)
    auto issue_next = [&](int c, int buf) {
        if constexpr (PROJ && MLP_PROJ_LDS) {
            if (c + 1 < nch) issue(c + 1, buf);
            else issue_proj(buf);
        } else {
            issue(c + 1 < nch ? c + 1 : 0, buf);   // (one inlined copy of the issue loop, as before)
        }
    };
    auto wfrag = [&](const char* base, int row, int c) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(base + row * 128 + ((c ^ (row & 7)) << 4));
    };

    const int ngroups = (M + TPG - 1) / TPG;
    int buf = 0;
    if constexpr (PROJ && MLP_PROJ_LDS) issue_proj(0);
    else issue(0, 0);
#pragma unroll 1
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int m0 = grp * TPG + wave * TPW;
        // ---- x is read ONCE, in accumulator layout (lane: token lr, channels 16 nt + 4 g + r): it seeds the
        // fc2 accumulators (the residual never leaves them) and LayerNorm 2 is taken from the same registers.
        // A lane's accumulator channels are the 8 k-slots 8g..8g+7 of k-step ks if slot 8g + 4t + r means
        // channel 32 ks + 16 t + 4 g + r: fc1's K order is packed that way, so no transpose is needed.
        // Rows past M are clamped and never stored.
        bf16x8 xn[MT][KS];
        f32x4 acc2[MT][NTO];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = min(m0 + mt * 16 + lr, M - 1);
            const float* row = p.x + (size_t)m * d;
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) {
                const int c0 = 16 * nt + 4 * g;
                acc2[mt][nt] = c0 < d ? load4(row + c0) : zero4;
            }
        }
        if constexpr (PROJ) {
            // ---- y = x + b_proj + a . W_proj^T: the token's attention rows are the B operand as they lie in memory (lane:
            //      token lr, channels 32 ks + 8 g ..), the W_proj tile the A operand; the result lands in acc2's layout ----------
            bf16x8 af[MT][KS];
            const bf16* __restrict__ ap = static_cast<const bf16*>(p.a);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bf16* row = ap + (size_t)min(m0 + mt * 16 + lr, M - 1) * p.lda;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int k0 = 32 * ks + 8 * g;
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (k0 < d) v = *reinterpret_cast<const uint4*>(row + k0);
                    af[mt][ks] = __builtin_bit_cast(bf16x8, v);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) {
                const int c0 = 16 * nt + 4 * g;
                if (c0 < d) {
                    const f32x4 b = load4(p.pbias + c0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc2[mt][nt] += b;
                }
            }
            if constexpr (MLP_PROJ_LDS) {
                lds_dma_wait_barrier();   // the W_proj chunk landed; every wave is done with the other buffer
                // (asm reads, ordered by hand, for the reason given at mlp_rd; fragment (nt, ks) at (nt KS + ks) KiB + lane * 16)
                const unsigned pb = (unsigned)(size_t)(mlp_lds_cp)(smem) + buf * CB + lane * 16;
                // one k-step's NTO fragments are read while the previous k-step multiplies
                bf16x8 pw[2][NTO];
                mlp_static_for<0, NTO>([&](auto nt_) { mlp_rd<(decltype(nt_)::value * KS) * 1024>(pb, pw[0][decltype(nt_)::value]); });
                issue(0, buf ^ 1);   // the first MLP chunk
                mlp_static_for<0, KS>([&](auto ks_) {
                    constexpr int ks = decltype(ks_)::value;
                    mlp_lds_wait();
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (ks + 1 < KS)
                        mlp_static_for<0, NTO>([&](auto nt_) { mlp_rd<(decltype(nt_)::value * KS + ks + 1) * 1024>(pb, pw[(ks + 1) & 1][decltype(nt_)::value]); });
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) mma16(pw[ks & 1][nt], af[mt][ks], acc2[mt][nt]);
                    __builtin_amdgcn_sched_barrier(0);
                });
                buf ^= 1;
            } else {
                // W_proj fragments straight from L2 (28 KiB, hot: every token group of every workgroup reads them), one k-step
                // ahead of the MFMAs that use them: no extra step in the chunk stream, no extra barrier (as one more chunk of
                // the LDS stream -- MLP_PROJ_LDS, the default -- the kernel takes 524 us against 442 without the projection; this form 595)
                const bf16x8* __restrict__ pf = reinterpret_cast<const bf16x8*>(p.pimg) + lane;
                bf16x8 pw[2][NTO];
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) pw[0][nt] = pf[(nt * KS) * 64];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 1 < KS) {
#pragma unroll
                        for (int nt = 0; nt < NTO; ++nt) pw[(ks + 1) & 1][nt] = pf[(nt * KS + ks + 1) * 64];
                    }
#pragma unroll
                    for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) mma16(pw[ks & 1][nt], af[mt][ks], acc2[mt][nt]);
                }
            }
        }
        {
            float mean[MT], rstd[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float s = 0.f;
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) s += (acc2[mt][nt][0] + acc2[mt][nt][1]) + (acc2[mt][nt][2] + acc2[mt][nt][3]);
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                mean[mt] = s / (float)d;
                float q = 0.f;
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
                    if (16 * nt + 4 * g < d) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float dd = acc2[mt][nt][r] - mean[mt];
                            q += dd * dd;
                        }
                    }
                q += __shfl_xor(q, 16, 64);
                q += __shfl_xor(q, 32, 64);
                rstd[mt] = 1.0f / sqrtf(q / (float)d + p.eps);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int nt = 2 * ks + t, c0 = 16 * nt + 4 * g;
                    const bool ok = nt < NTO && c0 < d;
                    f32x4 gm = zero4, bt = zero4;
                    if (ok) { gm = load4(p.gamma + c0); bt = load4(p.beta + c0); }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            xn[mt][ks][4 * t + r] = (bf16)(ok ? (acc2[mt][nt < NTO ? nt : 0][r] - mean[mt]) * rstd[mt] * gm[r] + bt[r] : 0.f);
                }
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) {   // + fc2 bias
                const int c0 = 16 * nt + 4 * g;
                if (c0 < d) {
                    const f32x4 b = load4(b2p + c0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc2[mt][nt] += b;
                }
            }
        }

#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            lds_dma_wait_barrier();   // chunk c landed; every wave is done with the other buffer
            // next chunk in stream order (wraps to chunk 0 of the next token group)
            const bool more = c + 1 < nch || grp + (int)gridDim.x < ngroups;
            if constexpr (KS == 4 && NTO == 7) {
                // the 112-wide stage: hand-ordered LDS reads, so that the DMA of the next chunk really
                // runs under this chunk's MFMAs and GELU.  Fragment (row tile nt, 16-byte column c) sits at
                // row * 128 + ((c ^ (row & 7)) << 4), row = 16 nt + lr: two per-lane bases (c = g, c = 4 + g)
                // plus immediates.
                const unsigned cb = (unsigned)(size_t)(mlp_lds_cp)(smem) + buf * CB;
                const unsigned sw0 = lr * 128 + ((g ^ (lr & 7)) << 4), sw1 = lr * 128 + (((4 + g) ^ (lr & 7)) << 4);
                const unsigned a0 = cb + sw0, a1 = cb + sw1, ab = cb + I::B1_OFF + g * 16;
                bf16x8 wa[8], wb[8], w2f[2 * NTO];
                f32x4 acc1[MT][4], bi[4];   // the chunk's fc1 biases are the C operand of each tile's first MFMA
                mlp_static_for<0, 4>([&](auto nt_) { mlp_rd<decltype(nt_)::value * 64>(ab, bi[decltype(nt_)::value]); });
                mlp_static_for<0, 8>([&](auto i_) {   // k-steps 0, 1
                    constexpr int i = decltype(i_)::value, ks = i >> 2, nt = i & 3;
                    mlp_rd<(ks >> 1) * 8192 + nt * 2048>((ks & 1) ? a1 : a0, wa[i]);
                });
                if (more) issue_next(c, buf ^ 1);
                mlp_lds_wait();
                mlp_static_for<0, 8>([&](auto i_) {   // k-steps 2, 3
                    constexpr int i = decltype(i_)::value, ks = 2 + (i >> 2), nt = i & 3;
                    mlp_rd<(ks >> 1) * 8192 + nt * 2048>((ks & 1) ? a1 : a0, wb[i]);
                });
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if (i < 4) acc1[mt][i] = bi[i];
                        mma16(wa[i], xn[mt][i >> 2], acc1[mt][i & 3]);
                    }
                __builtin_amdgcn_sched_barrier(0);
                mlp_lds_wait();
                mlp_static_for<0, 2 * NTO>([&](auto i_) {   // all fc2 fragments of the chunk: they land under the GELU
                    constexpr int i = decltype(i_)::value, k2 = i / NTO, nt = i % NTO;
                    mlp_rd<I::W1_B + nt * 2048>(k2 ? a1 : a0, w2f[i]);
                });
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) mma16(wb[i], xn[mt][2 + (i >> 2)], acc1[mt][i & 3]);
                bf16x8 hf[MT][2];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#if MLP_ABL & 4
                                hf[mt][k2][4 * t + r] = (bf16)(acc1[mt][2 * k2 + t][r]);
#else
                                hf[mt][k2][4 * t + r] = (bf16)gelu_t<true>(acc1[mt][2 * k2 + t][r]);
#endif
                __builtin_amdgcn_sched_barrier(0);
                mlp_lds_wait();
                __builtin_amdgcn_sched_barrier(0);   // without it the first fc2 MFMA is scheduled above the wait (tools/asm_lds_hazard.py)
#pragma unroll
                for (int i = 0; i < 2 * NTO; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) mma16(w2f[i], hf[mt][i / NTO], acc2[mt][i % NTO]);
                __builtin_amdgcn_sched_barrier(0);
                buf ^= 1;
                continue;
            }
            if (more) issue_next(c, buf ^ 1);
#if MLP_ABL & 2
            buf ^= 1;
            continue;
#endif
            const char* w1b = smem + buf * CB;
            const char* w2b = w1b + I::W1_B;
            f32x4 acc1[MT][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(w1b + I::B1_OFF + (16 * nt + 4 * g) * 4);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc1[mt][nt] = b;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const bf16x8 wf = wfrag(w1b + (ks >> 1) * 8192, 16 * nt + lr, (ks & 1) * 4 + g);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) mma16(wf, xn[mt][ks], acc1[mt][nt]);
                }
            __builtin_amdgcn_sched_barrier(0);   // keep the W2 fragment reads from being hoisted above the GELU
            bf16x8 hf[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    bf16x8 f;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#if MLP_ABL & 4
                            f[4 * t + r] = (bf16)(acc1[mt][2 * k2 + t][r]);
#else
                            f[4 * t + r] = (bf16)gelu_t<true>(acc1[mt][2 * k2 + t][r]);
#endif
                    hf[mt][k2] = f;
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) {
                    const bf16x8 wf = wfrag(w2b, 16 * nt + lr, k2 * 4 + g);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) mma16(wf, hf[mt][k2], acc2[mt][nt]);
                }
            buf ^= 1;
        }
        // ---- store: the accumulators already hold x + b2 + fc2(...) ------------------------------------------
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + mt * 16 + lr;
            if (m < M) {
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) {
                    const int c0 = 16 * nt + 4 * g;
                    if (c0 < d) store4(p.y + (size_t)m * d + c0, acc2[mt][nt]);
                }
            }
        }
        // ---- optional LayerNorm 1 of the next block, from the same registers (hieradet.py:136) --------------
        if (p.xn_next) {   // kernel-uniform
            bf16* __restrict__ xo = static_cast<bf16*>(p.xn_next);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float sacc = 0.f;
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
                    if (16 * nt + 4 * g < d) sacc += (acc2[mt][nt][0] + acc2[mt][nt][1]) + (acc2[mt][nt][2] + acc2[mt][nt][3]);
                const float mean = group4_sum(sacc) / (float)d;
                float q = 0.f;
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
                    if (16 * nt + 4 * g < d) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float dd = acc2[mt][nt][r] - mean;
                            q += dd * dd;
                        }
                    }
                const float rstd = 1.0f / sqrtf(group4_sum(q) / (float)d + p.eps);
                const int m = m0 + mt * 16 + lr;
                bf16* rowp = xo + (size_t)min(m, M - 1) * p.ldn;
                if (d == 16 * NTO && (p.ldn & 7) == 0) {   // kernel-uniform: full width, 16-byte rows
                    // two column tiles per 16-byte store (store instructions, not bytes, are what the write-back costs)
#pragma unroll
                    for (int nt = 0; nt + 1 < NTO; nt += 2) {
                        f32x4 v[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int c0 = 16 * (nt + t) + 4 * g;
                            const f32x4 gm = load4(p.gamma_n + c0), bt = load4(p.beta_n + c0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[t][r] = (acc2[mt][nt + t][r] - mean) * rstd * gm[r] + bt[r];
                        }
                        store_bf16_tile_pair(rowp + 16 * nt, g, v[0], v[1], m < M);
                    }
                    if constexpr (NTO % 2 == 1) {
                        const int c0 = 16 * (NTO - 1) + 4 * g;
                        const f32x4 gm = load4(p.gamma_n + c0), bt = load4(p.beta_n + c0);
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (acc2[mt][NTO - 1][r] - mean) * rstd * gm[r] + bt[r];
                        if (m < M) store4(rowp + c0, v);
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < NTO; ++nt) {
                        const int c0 = 16 * nt + 4 * g;
                        if (c0 < d && m < M) {
                            const f32x4 gm = load4(p.gamma_n + c0), bt = load4(p.beta_n + c0);
                            f32x4 v;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = (acc2[mt][nt][r] - mean) * rstd * gm[r] + bt[r];
                            store4(rowp + c0, v);
                        }
                    }
                }
            }
        }
    }
}

// supported widths: d <= 64 (KS 2), d <= 128 (KS 4), d <= 144 (KS 5), d <= 224 (KS 7); d % 8 == 0
static int mlp_variant(int d) {
    if (d % 8 != 0 || d < 16) return -1;
    if (d <= 64) return 0;
    if (d <= 112) return 1;
    if (d > 128 && d <= 144) return 3;   // Hiera-L stage 1 (144): 5 k-steps and 9 column tiles instead of the 224-wide 7 and 14
    if (d > 144 && d <= 224) return 2;
    return -1;
}
bool mlpblock_supported(int d) { return mlp_variant(d) >= 0; }
// the projection in front: where the W_proj fragments fit one chunk buffer (the 64- and 112-wide variants)
bool mlpblock_proj_supported(int d) { return mlp_variant(d) == 0 || mlp_variant(d) == 1; }
size_t mlpblock_proj_image_bytes(int d) {
    return mlp_variant(d) == 0 ? (size_t)MlpImg<2, 4>::CHUNK_B : mlp_variant(d) == 1 ? (size_t)MlpImg<4, 7>::CHUNK_B : 0;
}
int launch_pack_mlpblock_proj(const float* wproj, const float* bproj, int d, void* pimg, float* pbias, hipStream_t s) {
    LSAM_REQUIRE(wproj && bproj && pimg && pbias, LSAM_ERR_BAD_ARG, "pack_mlpblock_proj: null argument");
    LSAM_REQUIRE(mlpblock_proj_supported(d), LSAM_ERR_UNSUPPORTED, "pack_mlpblock_proj: width %d", d);
    if (mlp_variant(d) == 0) hipLaunchKernelGGL((pack_mlpproj_kernel<2, 4>), dim3(32), dim3(256), 0, s, wproj, bproj, d, static_cast<char*>(pimg), pbias);
    else hipLaunchKernelGGL((pack_mlpproj_kernel<4, 7>), dim3(32), dim3(256), 0, s, wproj, bproj, d, static_cast<char*>(pimg), pbias);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
int mlpblock_chunks(int d) { return (4 * d + 63) / 64; }
size_t mlpblock_image_bytes(int d) {
    const int nch = mlpblock_chunks(d);
    switch (mlp_variant(d)) {
        case 0: return (size_t)nch * MlpImg<2, 4>::CHUNK_B;
        case 1: return (size_t)nch * MlpImg<4, 7>::CHUNK_B;
        case 2: return (size_t)nch * MlpImg<7, 14>::CHUNK_B;
        case 3: return (size_t)nch * MlpImg<5, 9>::CHUNK_B;
    }
    return 0;
}
size_t mlpblock_bias_floats(int) { return 16 * 14; }

int launch_pack_mlpblock(const float* w1, const float* b1, const float* w2, const float* b2, int d, void* img,
                         float* bias_p, hipStream_t s) {
    LSAM_REQUIRE(w1 && b1 && w2 && b2 && img && bias_p, LSAM_ERR_BAD_ARG, "pack_mlpblock: null argument");
    const int v = mlp_variant(d), nch = mlpblock_chunks(d);
    LSAM_REQUIRE(v >= 0, LSAM_ERR_UNSUPPORTED, "pack_mlpblock: width %d", d);
    char* im = static_cast<char*>(img);
    if (v == 0) hipLaunchKernelGGL((pack_mlpblock_kernel<2, 4>), dim3(128), dim3(256), 0, s, w1, b1, w2, b2, d, nch, im, bias_p);
    if (v == 1) hipLaunchKernelGGL((pack_mlpblock_kernel<4, 7>), dim3(128), dim3(256), 0, s, w1, b1, w2, b2, d, nch, im, bias_p);
    if (v == 2) hipLaunchKernelGGL((pack_mlpblock_kernel<7, 14>), dim3(128), dim3(256), 0, s, w1, b1, w2, b2, d, nch, im, bias_p);
    if (v == 3) hipLaunchKernelGGL((pack_mlpblock_kernel<5, 9>), dim3(128), dim3(256), 0, s, w1, b1, w2, b2, d, nch, im, bias_p);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

template <int KS, int NTO, int MT, int NW, bool PROJ = false> static int launch_mlp(const MlpBlockArgs& a, hipStream_t s) {
    constexpr int LDSB = 2 * MlpImg<KS, NTO>::CHUNK_B;
    auto kern = mlpblock_kernel<KS, NTO, MT, NW, PROJ>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        attr_set = true;
    }
    const int ngroups = cdiv(a.M, 16 * MT * NW);
    // persistent; two 4-wave workgroups share a CU when their LDS fits, so that one workgroup's
    // memory phases (LayerNorm reads, residual + store) overlap the other's MFMA/GELU phases
    const int per_cu = (NW == 4 && 2 * LDSB <= 160 * 1024) ? 2 : 1;
    const int grid = ngroups < 256 * per_cu ? ngroups : 256 * per_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), LDSB, s, a);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

int launch_mlpblock(const MlpBlockArgs& a, hipStream_t s) {
    LSAM_REQUIRE(a.x && a.y && a.gamma && a.beta && a.img && a.bias_p, LSAM_ERR_BAD_ARG, "mlpblock: null argument");
    LSAM_REQUIRE(a.M > 0 && a.nch == mlpblock_chunks(a.d), LSAM_ERR_BAD_ARG, "mlpblock: bad shape");
    LSAM_REQUIRE(aligned16(a.x) && aligned16(a.y) && aligned16(a.img) && aligned16(a.gamma) && aligned16(a.beta) &&
                     aligned16(a.bias_p), LSAM_ERR_BAD_ARG, "mlpblock: pointers must be 16-byte aligned");
    if (a.xn_next)
        LSAM_REQUIRE(a.gamma_n && a.beta_n && a.ldn >= a.d && a.ldn % 4 == 0 && aligned16(a.gamma_n) && aligned16(a.beta_n) &&
                         (reinterpret_cast<uintptr_t>(a.xn_next) & 15) == 0,
                     LSAM_ERR_BAD_ARG, "mlpblock: next-LayerNorm output needs gamma, beta, 16-byte alignment and ldn %% 4 == 0");
    if (a.a)
        LSAM_REQUIRE(mlpblock_proj_supported(a.d) && a.pimg && a.pbias && a.lda >= a.d && a.lda % 8 == 0 && aligned16(a.a) &&
                         aligned16(a.pimg) && aligned16(a.pbias),
                     LSAM_ERR_BAD_ARG, "mlpblock: the projection in front needs width <= 112, its packed image and bias, lda %% 8 == 0");
    ProfScope ps(s, (double)a.M * a.d * (8.0 + (a.xn_next ? 2.0 : 0.0) + (a.a ? 2.0 : 0.0)), PROF_MLPBLOCK, a.M, a.d, 0);   // x in + y out (+ bf16 next LN, + bf16 a)
    switch (mlp_variant(a.d)) {
        case 0: return a.a ? launch_mlp<2, 4, 2, 4, true>(a, s) : launch_mlp<2, 4, 2, 4>(a, s);
        case 1: return a.a ? launch_mlp<4, 7, 2, 4, true>(a, s) : launch_mlp<4, 7, 2, 4>(a, s);
        case 2: return launch_mlp<7, 14, 2, 8>(a, s);
        case 3: return launch_mlp<5, 9, 2, 8>(a, s);
    }
    set_error("mlpblock: width %d not supported", a.d);
    return LSAM_ERR_UNSUPPORTED;
}

}  // namespace lsam
