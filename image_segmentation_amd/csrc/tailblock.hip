// Fused tail of a MultiScaleBlock for the 224- and 448-wide stages (bf16 MFMA), ONE kernel for
//   y       = res + proj(a)                          (hieradet.py:79,163: attention projection + shortcut)
//   out     = y + fc2(GELU(fc1(LayerNorm2(y))))      (hieradet.py:165, MLP: sam2_utils.py:131-136)
//   xn_next = LayerNorm1_next(out)                   (optional: the next block's norm1, hieradet.py:136)
// Un-fused, this is five launches (proj GEMM, LN2, fc1, fc2, next LN1) whose intermediates (y twice,
// LN2 out, the 4d-wide hidden activation) cross HBM: ~1 GB per stage-3 block at batch 16, which makes
// fc2 and proj HBM-bound.  Here only a (bf16), res and out (fp32) and xn_next (bf16) touch HBM.
//
// A wave owns 32 consecutive tokens and keeps ALL of their state in registers as MFMA operands
// (one wave per SIMD, ~400 VGPRs at d = 448):
//   acc  [2][NTO] f32x4  swapped-product accumulators: lane (g, lr) holds token lr, channels
//                        16 nt + 4 g + r.  Initialised with res + b_proj, they take proj(a) -> y, and
//                        keep accumulating the fc2 product on top of y: the residual never leaves them.
//   xf   [2][KS] bf16x8  B-operand fragments: first a (natural channel order), after LN2 the normalised
//                        y.  A lane of `acc` owns exactly the 8 channels of k-slots 8g..8g+7 of k-step ks
//                        if slot 8g + 4t + r is made to mean channel 32 ks + 16 t + 4 g + r, so xn is built
//                        from acc in registers (no LDS transpose); fc1's K order is permuted to match when
//                        it is packed.  The same trick turns GELU(fc1) into fc2's B operand.
// Weights are streamed as 1 KiB MFMA A-operand fragments (already in lane order: conflict-free
// ds_read_b128 at lane * 16) in equal chunks, global -> LDS by LDS-DMA, double-buffered, one barrier
// per chunk.  Stream per token group: ceil(KS/2) proj chunks (two k-steps of all NTO row tiles), then
// NCH + 1 MLP chunks; MLP chunk c holds fc1 rows of hidden units 32c..32c+31 (+ their biases) and the
// fc2 columns of hidden chunk c - 1, so that the GELU of chunk c (VALU) is issued between the fc2
// MFMAs of chunk c - 1 instead of in front of its own.
#include <type_traits>

#include "kernels.hpp"

namespace lsam {

#ifndef TAIL_ABL
#define TAIL_ABL 0
#endif
typedef __attribute__((address_space(3))) void* tail_lds_p;
typedef __attribute__((address_space(3))) const char* tail_lds_cp;

#ifndef TAIL_BS224
#define TAIL_BS224 2
#endif
#ifndef TAIL_INTERLEAVE
#define TAIL_INTERLEAVE 1
#endif
#ifndef TAIL_DMA_FRAC
#define TAIL_DMA_FRAC 4   // whole step, A/B on one box: 1 -> 859, 2 -> 861, 4 -> 862 frames/s
#endif
// number of leading fragment batches (of nb) that carry the chunk's DMA pieces
constexpr int DMAB(int nb) { return nb / TAIL_DMA_FRAC > 0 ? nb / TAIL_DMA_FRAC : 1; }

template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
// N consecutive 1 KiB fragments starting at fragment F0 of the chunk at LDS byte address `base`
// (which already includes lane * 16): asm reads with immediate offsets, see the kernel for why.
template <int F0, int N> __device__ __forceinline__ void lds_frags(unsigned base, bf16x8* dst) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(base), "n"(F0 * 1024));
    if constexpr (N > 1) lds_frags<F0 + 1, N - 1>(base, dst + 1);
}
__device__ __forceinline__ f32x4 lds_f32x4(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

typedef const __attribute__((address_space(1))) void* tail_gbl_p;

template <int KS, int NTO> struct TailImg {
    static constexpr int NPC = (KS + 1) / 2;   // proj chunks
    static constexpr int W1F = 2 * KS;         // fc1 fragments of an MLP chunk (2 row tiles x KS k-steps, tile major)
    static constexpr int W2F = NTO;            // fc2 fragments of an MLP chunk
    static constexpr int FR = (2 * NTO > W1F + W2F) ? 2 * NTO : W1F + W2F;
    static constexpr int B1_OFF = FR * 1024;   // the chunk's 32 fc1 biases (fp32) in a 1 KiB slot
    static constexpr int CHUNK_B = (FR + 1) * 1024;
    static constexpr int VEC = 16 * NTO;       // padded length of each packed fp32 vector
};

template <int KS, int NTO>
__global__ void pack_tailblock_kernel(const float* __restrict__ wproj, const float* __restrict__ bproj,
                                      const float* __restrict__ g2, const float* __restrict__ be2,
                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                      const float* __restrict__ w2, const float* __restrict__ b2, int d, int nch,
                                      char* __restrict__ img, float* __restrict__ vec) {
    using I = TailImg<KS, NTO>;
    const int hid = 4 * d, nck = I::NPC + nch + 1;
    const long total = (long)nck * I::CHUNK_B / 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long byte = 2 * i;
        const int ck = (int)(byte / I::CHUNK_B), in = (int)(byte % I::CHUNK_B);
        if (in >= I::B1_OFF) continue;   // fp32 bias slot: written below
        const int f = in >> 10, lane = (in & 1023) >> 4, j = (in & 15) >> 1;
        const int g = lane >> 4, lr = lane & 15;
        float v = 0.f;
        if (ck < I::NPC) {   // proj: rows = output channels, natural K order
            if (f < 2 * NTO) {
                const int kl = f / NTO, nt = f % NTO, ks = 2 * ck + kl;
                const int row = 16 * nt + lr, k = 32 * ks + 8 * g + j;
                if (ks < KS && row < d && k < d) v = wproj[(size_t)row * d + k];
            }
        } else {
            const int c = ck - I::NPC;
            const int perm = 16 * (j >> 2) + 4 * g + (j & 3);   // k-slot 8g + j <-> this offset inside a 32-block
            if (f < I::W1F) {
                const int t = f / KS, ks = f % KS;   // row-tile major: the first tile's GELU can start early
                const int hrow = 32 * c + 16 * t + lr, ch = 32 * ks + perm;
                if (c < nch && hrow < hid && ch < d) v = w1[(size_t)hrow * d + ch];
            } else if (f < I::W1F + I::W2F) {
                const int nt = f - I::W1F, row = 16 * nt + lr, hcol = 32 * (c - 1) + perm;
                if (c >= 1 && row < d && hcol < hid) v = w2[(size_t)row * hid + hcol];
            }
        }
        reinterpret_cast<bf16*>(img)[i] = (bf16)v;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nck * 256; i += gridDim.x * blockDim.x) {
        const int ck = i / 256, e = i % 256, c = ck - I::NPC;
        const bool ok = c >= 0 && c < nch && e < 32 && 32 * c + e < hid;
        reinterpret_cast<float*>(img + (size_t)ck * I::CHUNK_B + I::B1_OFF)[e] = ok ? b1[32 * c + e] : 0.f;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 4 * I::VEC; i += gridDim.x * blockDim.x) {
        const int which = i / I::VEC, e = i % I::VEC;
        const float* src = which == 0 ? bproj : which == 1 ? g2 : which == 2 ? be2 : b2;
        vec[i] = e < d ? src[e] : 0.f;
    }
}

template <int KS, int NTO, int MT, int NW, int WPC>
__global__ __launch_bounds__(64 * NW, WPC) void tailblock_kernel(const TailBlockArgs p) {
    using I = TailImg<KS, NTO>;
    constexpr int CB = I::CHUNK_B, NF = CB / 1024, TPW = 16 * MT, TPG = TPW * NW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lr = lane & 15;
    static_assert(32 * KS == 16 * NTO, "the kernel is specialised for d = 32 KS = 16 NTO (no channel guards)");
    constexpr int d = 16 * NTO;
    const int M = p.M, nch = p.nch;
    const int nck = I::NPC + nch + 1;
    constexpr float inv_d = 1.0f / (float)d;
    const char* __restrict__ img = static_cast<const char*>(p.img);
    const bf16* __restrict__ A = static_cast<const bf16*>(p.a);
    const float* __restrict__ vproj = p.vec;
    const float* __restrict__ vg2 = p.vec + I::VEC;
    const float* __restrict__ vb2 = p.vec + 2 * I::VEC;
    const float* __restrict__ vfc2 = p.vec + 3 * I::VEC;

    // LDS-DMA of one chunk: 1 KiB pieces wave, wave + NW, ...; buffer addressing keeps the per-lane part
    // of the address in ONE VGPR (lane * 16 + wave * 1024) and the moving part in the scalar offset.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(img), 0, nck * CB, 0x00020000);
    const unsigned dma_vo = (unsigned)(lane * 16 + wave * 1024);
    // The pieces of a chunk are not issued in one burst after the barrier but spread over the chunk's
    // fragment batches (issue_part<b, NBT>), so that their issue cost falls into MFMA shadows.
    auto issue_piece = [&](int ck, int buf, int i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (tail_lds_p)(smem + buf * CB + (wave + NW * i) * 1024), 16, dma_vo,
                                                 ck * CB + NW * i * 1024, 0, 0);
    };
    auto issue_part = [&](auto b_, auto nbt_, int ck, int buf) {
#if TAIL_ABL & 1
        return;
#endif
        constexpr int b = decltype(b_)::value, NBT = decltype(nbt_)::value;
        constexpr int NP = NF / NW, PPB = (NP + NBT - 1) / NBT;   // full pieces per wave, pieces per batch
        if constexpr (NF % NW != 0 && b == 0) {   // the odd piece first: whatever is issued last is what the next barrier waits for
            if (wave + NW * NP < NF) issue_piece(ck, buf, NP);
        }
#pragma unroll
        for (int i = b * PPB; i < (b + 1) * PPB && i < NP; ++i) issue_piece(ck, buf, i);
    };
    auto issue = [&](int ck, int buf) {   // whole chunk at once (kernel start only)
        static_for<0, 4>([&](auto b_) { issue_part(b_, std::integral_constant<int, 4>{}, ck, buf); });
    };
    // Weight fragments come out of LDS through inline-asm reads: hipcc puts s_waitcnt vmcnt(0) in front
    // of every LDS read it can see after an LDS-DMA (it assumes the DMA's LDS write may alias the
    // read), which would serialise the chunk stream and the MFMAs.  The reads it cannot see are
    // ordered by hand: lds_wait() before the first use of a batch, scheduling barriers around the users.
    const unsigned lds0 = (unsigned)(size_t)(tail_lds_cp)(smem);
    const unsigned lds_lane = lds0 + (unsigned)lane * 16u;
    constexpr int BS = (MT == 1 && NTO == 28) ? 4 : (MT >= 2 && NTO == 14) ? TAIL_BS224 : 7;   // weight fragments per prefetch batch (4 VGPRs each, two register sets)
    static_assert(NTO % BS == 0 && I::W1F % BS == 0 && I::FR == I::W1F + I::W2F, "batching assumes BS | NTO, BS | 2 KS");

    const int ngroups = (M + TPG - 1) / TPG;
    int buf = 0;
    if ((int)blockIdx.x < ngroups) issue(0, 0);
#pragma unroll 1
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int m0 = grp * TPG + wave * TPW;
        int mrow[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) mrow[mt] = min(m0 + mt * 16 + lr, M - 1);   // clamped rows are never stored

        // ---- operands of the projection: a fragments; accumulators start at res + b_proj ------------
        bf16x8 xf[MT][KS];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                xf[mt][ks] = *reinterpret_cast<const bf16x8*>(A + (size_t)mrow[mt] * p.lda + 32 * ks + 8 * g);
            }
        f32x4 acc[MT][NTO];
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) {
            const int c0 = 16 * nt + 4 * g;
            const f32x4 bp = load4(vproj + c0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = load4(p.res + (size_t)mrow[mt] * d + c0) + bp;
        }
        // ---- y = res + b + W_proj . a^T ------------------------------------------------------------
        // Weight fragments are read BS at a time into one of two register sets, one batch ahead of
        // the MFMAs that use them; the scheduling barriers keep hipcc from sinking each read down to
        // its use (it otherwise waits lgkmcnt(0) in front of every MFMA pair: one wave per SIMD has
        // nothing else to hide the LDS latency).
        static_for<0, I::NPC>([&](auto pc_) {
            constexpr int pc = decltype(pc_)::value;
            lds_dma_wait_barrier();   // chunk landed; every wave is done with the other buffer
            const unsigned base = lds_lane + buf * CB;
            constexpr int NBP = 2 * NTO / BS;
            bf16x8 w[2][BS];
            lds_frags<0, BS>(base, w[0]);
            static_for<0, NBP>([&](auto b_) {
                constexpr int b = decltype(b_)::value;
                constexpr int ks = 2 * pc + (b * BS) / NTO, ksc = ks < KS ? ks : KS - 1;
                lds_wait();   // batch b is in registers
                if constexpr (b + 1 < NBP && 2 * pc + ((b + 1) * BS) / NTO < KS) lds_frags<(b + 1) * BS, BS>(base, w[(b + 1) & 1]);
                // all pieces go out in the first DMAB batches, so that the last of them has the rest of the
                // chunk to land before the next barrier's vmcnt(0)
                if constexpr (b < DMAB(NBP)) issue_part(b_, std::integral_constant<int, DMAB(NBP)>{}, pc + 1, buf ^ 1);   // pc + 1 <= NPC < nck
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ks < KS) {   // KS may be odd: the last half chunk is empty
#pragma unroll
                    for (int i = 0; i < BS; ++i)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) mma16(w[b & 1][i], xf[mt][ksc], acc[mt][(b * BS + i) % NTO]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            buf ^= 1;
        });
        // ---- LayerNorm 2 of y, straight from the accumulators into fc1's B operands ---------------
        {
            float mean[MT], rstd[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float s = 0.f;
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) s += (acc[mt][nt][0] + acc[mt][nt][1]) + (acc[mt][nt][2] + acc[mt][nt][3]);
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                mean[mt] = s * inv_d;
                float q = 0.f;
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dd = acc[mt][nt][r] - mean[mt];
                        q = fmaf(dd, dd, q);
                    }
                q += __shfl_xor(q, 16, 64);
                q += __shfl_xor(q, 32, 64);
                rstd[mt] = 1.0f / sqrtf(q * inv_d + p.eps);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int nt = 2 * ks + t;
                    const int c0 = 16 * nt + 4 * g;
                    const f32x4 gm = load4(vg2 + c0), bt = load4(vb2 + c0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            xf[mt][ks][4 * t + r] = (bf16)fmaf((acc[mt][nt][r] - mean[mt]) * rstd[mt], gm[r], bt[r]);
                }
        }
        // ---- MLP: hidden units in chunks of 32, GELU of chunk c between the fc2 MFMAs of chunk c-1 --
        bf16x8 hprev[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 8; ++j) hprev[mt][j] = (bf16)0.f;
#pragma unroll 1
        for (int c = 0; c <= nch; ++c) {
            lds_dma_wait_barrier();
            // next chunk of the stream; after the very last chunk of this workgroup chunk 0 is fetched once
            // more (unused; drained before the kernel ends) so that the loop body stays branch-free
            const int next_ck = I::NPC + c + 1 < nck ? I::NPC + c + 1 : 0;
#if TAIL_ABL & 2
            buf ^= 1;
            continue;
#endif
            // the lane id is re-derived here (2 VALU ops) so that no loop-invariant address has to stay
            // in a register across the loop: under this register pressure hipcc spills it, and a scratch
            // reload waits on vmcnt(0), i.e. on the chunk DMA that was just issued
            unsigned ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            const unsigned cbase = lds0 + buf * CB;
            const unsigned base = cbase + ln * 16u;
            constexpr int NB1 = I::W1F / BS, NB2 = I::W2F / BS, NB = NB1 + NB2;
            bf16x8 w[2][BS];
            f32x4 acc1[MT][2];
            {   // the chunk's fc1 biases seed the accumulators (lane owns hidden units 16 t + 4 g + r)
                const unsigned baddr = cbase + I::B1_OFF + (ln >> 4) * 16u;
                const f32x4 b0 = lds_f32x4(baddr), b1 = lds_f32x4(baddr + 64);
                lds_frags<0, BS>(base, w[0]);
                lds_wait();
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { acc1[mt][0] = b0; acc1[mt][1] = b1; }
            }
            bf16x8 hnew[MT];
            static_for<0, NB>([&](auto b_) {
                constexpr int b = decltype(b_)::value;
                lds_wait();   // batch b is in registers
                if constexpr (b + 1 < NB) lds_frags<(b + 1) * BS, BS>(base, w[(b + 1) & 1]);
                if constexpr (b < DMAB(NB)) issue_part(b_, std::integral_constant<int, DMAB(NB)>{}, next_ck, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                // GELU work items: (mt, t, half) = 2 values each.  Row tile t = 0 is complete after the first
                // half of the fc1 batches, t = 1 after all of them; the items are spread over the batches that
                // follow so that their VALU instructions issue in the shadow of those batches' MFMAs.
                auto gelu_item = [&](int mt, int t, int half) {
                    float g0, g1;
#if TAIL_ABL & 4
                    g0 = acc1[mt][t][2 * half]; g1 = acc1[mt][t][2 * half + 1];
#else
                    g0 = gelu_t<true>(acc1[mt][t][2 * half]); g1 = gelu_t<true>(acc1[mt][t][2 * half + 1]);
#endif
                    asm volatile("" : "+v"(g0), "+v"(g1));   // pin to this batch (LLVM otherwise sinks it to the last use)
                    hnew[mt][4 * t + 2 * half] = (bf16)g0;
                    hnew[mt][4 * t + 2 * half + 1] = (bf16)g1;
                };
                constexpr int NIT = 2 * MT;                       // items per row tile
                constexpr int H1 = NB1 / 2;                       // batches [H1, NB1): t = 0 items; [NB1, NB): t = 1 items
                constexpr int PB0 = (NIT + (NB1 - H1) - 1) / (NB1 - H1), PB1 = (NIT + NB2 - 1) / NB2;
                if constexpr (b >= H1 && b < NB1) {
#pragma unroll
                    for (int q = (b - H1) * PB0; q < (b - H1 + 1) * PB0 && q < NIT; ++q) gelu_item(q >> 1, 0, q & 1);
                }
                if constexpr (b >= NB1) {
#pragma unroll
                    for (int q = (b - NB1) * PB1; q < (b - NB1 + 1) * PB1 && q < NIT; ++q) gelu_item(q >> 1, 1, q & 1);
                }
                if constexpr (b < NB1) {   // fc1: fragment f = t KS + ks
#pragma unroll
                    for (int i = 0; i < BS; ++i) {
                        const int f = b * BS + i, t = f / KS, ks = f % KS;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) mma16(w[b & 1][i], xf[mt][ks], acc1[mt][t]);
                    }
                } else {         // fc2 of the previous hidden chunk
#pragma unroll
                    for (int i = 0; i < BS; ++i)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) mma16(w[b & 1][i], hprev[mt], acc[mt][(b - NB1) * BS + i]);
                }
#if TAIL_INTERLEAVE
                // issue order inside the batch: one MFMA, then the VALU instructions that fit in its shadow
                // (an MFMA blocks vector issue for 8 of its 16 cycles).  Left alone, hipcc emits the GELU
                // work as one block between MFMA runs and the matrix pipe idles meanwhile.
                if constexpr (b >= H1) {
                    constexpr int items = b < NB1 ? PB0 : PB1, per = (items * 24 + BS * MT - 1) / (BS * MT);
#pragma unroll
                    for (int i = 0; i < BS * MT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, per, 0);   // VALU
                    }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) hprev[mt] = hnew[mt];
            buf ^= 1;
        }
        // ---- + fc2 bias, store; optional LayerNorm 1 of the next block -------------------------------
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) {
            const f32x4 b = load4(vfc2 + 16 * nt + 4 * g);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] += b;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + mt * 16 + lr;
            if (m < M) {
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) store4(p.out + (size_t)m * d + 16 * nt + 4 * g, acc[mt][nt]);
            }
        }
        if (p.xn_next) {   // kernel-uniform
            bf16* __restrict__ xo = static_cast<bf16*>(p.xn_next);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float s = 0.f;
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) s += (acc[mt][nt][0] + acc[mt][nt][1]) + (acc[mt][nt][2] + acc[mt][nt][3]);
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                const float mean = s * inv_d;
                float q = 0.f;
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dd = acc[mt][nt][r] - mean;
                        q = fmaf(dd, dd, q);
                    }
                q += __shfl_xor(q, 16, 64);
                q += __shfl_xor(q, 32, 64);
                const float rstd = 1.0f / sqrtf(q * inv_d + p.eps);
                const int m = m0 + mt * 16 + lr;
                static_assert(NTO % 2 == 0, "bf16 rows are stored as pairs of column tiles");
#pragma unroll
                for (int nt = 0; nt < NTO; nt += 2) {
                    f32x4 v[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int c0 = 16 * (nt + t) + 4 * g;
                        const f32x4 gm = load4(p.gamma_n + c0), bt = load4(p.beta_n + c0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[t][r] = fmaf((acc[mt][nt + t][r] - mean) * rstd, gm[r], bt[r]);
                    }
                    store_bf16_tile_pair(xo + (size_t)min(m, M - 1) * p.ldn + 16 * nt, g, v[0], v[1], m < M);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last (unused) chunk prefetch must land before the LDS is released
}

// =====================================================================================================
// Two waves per SIMD for the 448-wide tail: a PAIR of waves shares a 32-token tile.
// The one-wave-per-SIMD kernel above cannot overlap anything: its MFMAs, the GELU VALU work, the DMA issue
// and the LDS waits of a chunk simply add up (rocprof: matrix pipe busy ~36 %).  Here a workgroup has 8 waves
// (two per SIMD, 256 registers each) and the two waves of a pair split the 32-token tile's state:
//   wave hh keeps output channels [224 hh, 224 hh + 224): acc[2][NTO/2] (proj, fc2, residual) and, because a
//   lane's accumulator channels ARE its k-slots (see above), the LayerNorm-2 output for the SAME channels,
//   i.e. the K half hh of fc1's B operand.
//   fc1 of a hidden chunk is therefore a K-split: each wave produces a partial 32-hidden x 32-token product
//   over its 224 channels; the partials are exchanged through LDS as bf16 (2 KiB per wave and chunk, double
//   buffered, riding on the chunk barrier), both waves add them in the same order, apply bias + GELU
//   (redundantly: 16 values per lane) and so hold the identical fc2 B operand.
//   fc2 is an N-split: wave hh accumulates its 224 output channels.
// Pipeline per wave and iteration c (one barrier each): fc1 partial of chunk c | GELU of chunk c-1 from the
// exchanged partials | fc2 of chunk c-2.  The three are independent, so with two waves per SIMD the hardware
// overlaps one wave's VALU/LDS time with the other's MFMAs.
// wave-pair kernel: fraction of a chunk's fragment batches that carry the next chunk's DMA pieces (see DMAB)
#ifndef PAIR_DMA_FRAC
#define PAIR_DMA_FRAC 2   // whole step, A/B/A/B on one box: 1 -> 862.7, 2 -> 869.2, 4 -> 866.4 frames/s
#endif
constexpr int PDMAB(int nb) { return nb / PAIR_DMA_FRAC > 0 ? nb / PAIR_DMA_FRAC : 1; }
#ifndef PAIR_BSP
#define PAIR_BSP 4
#endif
// non-temporal hints on the activation streams (bit 0: residual loads, 1: a loads, 2: out stores), so that they do not
// evict the weight image from the XCD's L2
#ifndef PAIR_NT
#define PAIR_NT 0
#endif
#if PAIR_NT & 2
#define PAIR_LDA(ptr) __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(ptr))
#else
#define PAIR_LDA(ptr) (*reinterpret_cast<const bf16x8*>(ptr))
#endif
#ifndef PAIR_BSM_ODD
#define PAIR_BSM_ODD 2
#endif
#ifndef PAIR_BSM
#define PAIR_BSM 2   // whole step, A/B on one box (with PAIR_DMA_FRAC 2): 4 -> 870, 2 -> 876, 1 -> 872 frames/s
#endif
template <int KS, int NTO> struct PairImg {
    // Odd KS (d = 224: 7 k-steps, 7 column tiles per wave): a wave's 112 channels are 3 1/2 k-steps of fc1's K -- its last
    // k-step has the upper 16 k-slots empty (zero operand slots, zero weights: 8 fc1 fragments where 7 would do), and the last
    // proj chunk holds one k-step.
    static_assert(NTO == 2 * KS, "d = 32 KS = 16 NTO");
    static constexpr int KSH = (KS + 1) / 2, NTH = NTO / 2;
    static constexpr int NPC = (KS + 1) / 2;        // proj chunks: two k-steps of all row tiles
    static constexpr int HFP = 2 * NTH, HFM = 2 * KSH + NTH;   // fragments a wave uses of a proj / an MLP chunk
    static constexpr int HF = HFP > HFM ? HFP : HFM;           //   proj: [kl][ntl]; MLP: fc1 [ksl][t] (2 KSH), then fc2 [ntl] (NTH)
    static constexpr int FR = 2 * HF;               // fragments per chunk; wave half hh owns [hh HF, (hh+1) HF)
    static constexpr int B1_OFF = FR * 1024;        // fc1 biases of the chunk (they seed the fc1 accumulators)
    static constexpr int CHUNK_B = (FR + 1) * 1024;
    static constexpr int VEC = 16 * NTO;
    static constexpr int XCH_OFF = 2 * CHUNK_B;     // exchange: [2 parities]{fc1 partials [NW waves][1 KiB], GELU halves [NW waves][1 KiB]}
    static constexpr int stat_off(int nw) { return XCH_OFF + 2 * nw * 2048; }   // LayerNorm statistics: [2][NW waves][32] floats
    static constexpr int lds_bytes(int nw) { return stat_off(nw) + 2 * nw * 32 * 4; }
};

template <int KS, int NTO>
__global__ void pack_tailpair_kernel(const float* __restrict__ wproj, const float* __restrict__ bproj,
                                     const float* __restrict__ g2, const float* __restrict__ be2,
                                     const float* __restrict__ w1, const float* __restrict__ b1,
                                     const float* __restrict__ w2, const float* __restrict__ b2, int d, int nch,
                                     char* __restrict__ img, float* __restrict__ vec) {
    using I = PairImg<KS, NTO>;
    const int hid = 4 * d, nck = I::NPC + nch + 2;
    const long total = (long)nck * I::CHUNK_B / 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long byte = 2 * i;
        const int ck = (int)(byte / I::CHUNK_B), in = (int)(byte % I::CHUNK_B);
        if (in >= I::B1_OFF) continue;
        const int f = in >> 10, lane = (in & 1023) >> 4, j = (in & 15) >> 1;
        const int g = lane >> 4, lr = lane & 15;
        const int hh = f / I::HF, fl = f % I::HF;
        float v = 0.f;
        if (ck < I::NPC) {
            const int kl = fl / I::NTH, ntl = fl % I::NTH, ks = 2 * ck + kl;
            const int row = 16 * (I::NTH * hh + ntl) + lr, k = 32 * ks + 8 * g + j;
            if (fl < I::HFP && ks < KS) v = wproj[(size_t)row * d + k];
        } else {
            const int c = ck - I::NPC;
            if (fl < 2 * I::KSH) {   // fc1 of hidden chunk c, K half hh; row tile 0 of a half is the one it finalises
                // k-slot j of the wave's k-step ksl = channel 4 g + (j & 3) of its column tile 2 ksl + (j >> 2)
                const int ksl = fl >> 1, t = (fl & 1) ^ hh, ctile = 2 * ksl + (j >> 2);
                const int hrow = 32 * c + 16 * t + lr, ch = 16 * (I::NTH * hh + ctile) + 4 * g + (j & 3);
                if (c < nch && ctile < I::NTH) v = w1[(size_t)hrow * d + ch];
            } else if (fl < I::HFM) {   // fc2 of hidden chunk c - 2, output half hh; k-slots 0..3 = its own hidden tile
                const int ntl = fl - 2 * I::KSH, row = 16 * (I::NTH * hh + ntl) + lr;
                const int hcol = 32 * (c - 2) + 16 * ((j >> 2) ^ hh) + 4 * g + (j & 3);
                if (c >= 2 && c - 2 < nch) v = w2[(size_t)row * hid + hcol];
            }
        }
        reinterpret_cast<bf16*>(img)[i] = (bf16)v;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nck * 256; i += gridDim.x * blockDim.x) {
        const int ck = i / 256, e = i % 256, c = ck - I::NPC;   // biases of the chunk whose fc1 runs in this iteration
        const bool ok = c >= 0 && c < nch && e < 32;
        reinterpret_cast<float*>(img + (size_t)ck * I::CHUNK_B + I::B1_OFF)[e] = ok ? b1[32 * c + e] : 0.f;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 4 * I::VEC; i += gridDim.x * blockDim.x) {
        const int which = i / I::VEC, e = i % I::VEC;
        const float* src = which == 0 ? bproj : which == 1 ? g2 : which == 2 ? be2 : b2;
        vec[i] = e < d ? src[e] : 0.f;
    }
}

__device__ __forceinline__ void lds_write_b128(unsigned addr, const bf16x8& v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ bf16x8 lds_read_b128(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

// NW = 8: one workgroup per CU; NW = 4: two (they drift apart, so one's HBM phase -- residual and a loads, stores -- falls
// under the other's MFMAs, at twice the weight traffic from L2)
template <int KS, int NTO, int NW>
__global__ __launch_bounds__(64 * NW, 2) void tailpair_kernel(const TailBlockArgs p) {
    using I = PairImg<KS, NTO>;
    constexpr int KSH = I::KSH, NTH = I::NTH, HF = I::HF, MT = 2, CB = I::CHUNK_B, NF = CB / 1024;
    constexpr int XPAR = NW * 2048, XGELU = NW * 1024;   // exchange area: bytes per parity, offset of the GELU halves
    constexpr int TPG = 16 * MT * (NW / 2), d = 16 * NTO;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wave >> 1, hh = wave & 1;
    const int g = lane >> 4, lr = lane & 15;
    const int M = p.M, nch = p.nch;
    const int nck = I::NPC + nch + 2;
    constexpr float inv_d = 1.0f / (float)d;
    const char* __restrict__ img = static_cast<const char*>(p.img);
    const bf16* __restrict__ A = static_cast<const bf16*>(p.a);
    const float* __restrict__ vproj = p.vec;
    const float* __restrict__ vg2 = p.vec + I::VEC;
    const float* __restrict__ vb2 = p.vec + 2 * I::VEC;
    const float* __restrict__ vfc2 = p.vec + 3 * I::VEC;
    const int cbase0 = 16 * NTH * hh + 4 * g;   // first of this lane's channels; tile ntl adds 16 ntl

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(img), 0, nck * CB, 0x00020000);
    const unsigned dma_vo = (unsigned)(lane * 16 + wave * 1024);
    auto issue_piece = [&](int ck, int buf, int i) {
#if TAIL_ABL & 32
        ck &= 1;   // timing only: the stream stays on two L2-hot chunks
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (tail_lds_p)(smem + buf * CB + (wave + NW * i) * 1024), 16, dma_vo,
                                                 ck * CB + NW * i * 1024, 0, 0);
    };
    auto issue_part = [&](auto b_, auto nbt_, int ck, int buf) {
#if TAIL_ABL & 1
        return;
#endif
        constexpr int b = decltype(b_)::value, NBT = decltype(nbt_)::value;
        constexpr int NP = NF / NW, PPB = (NP + NBT - 1) / NBT;
        if constexpr (NF % NW != 0 && b == 0) {   // the odd piece first: whatever is issued last is what the barrier waits for
            if (wave + NW * NP < NF) issue_piece(ck, buf, NP);
        }
#pragma unroll
        for (int i = b * PPB; i < (b + 1) * PPB && i < NP; ++i) issue_piece(ck, buf, i);
    };
    const unsigned lds0 = (unsigned)(size_t)(tail_lds_cp)(smem);
    float* stat = reinterpret_cast<float*>(smem + I::stat_off(NW));
    // LayerNorm statistics of the pair's 32 tokens: each wave reduces its 224 channels, the halves meet in LDS
    auto pair_stats = [&](const f32x4 (&acc)[MT][NTH], float (&mean)[MT], float (&rstd)[MT]) {
        float s[MT], q[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            s[mt] = 0.f;
#pragma unroll
            for (int nt = 0; nt < NTH; ++nt) s[mt] += (acc[mt][nt][0] + acc[mt][nt][1]) + (acc[mt][nt][2] + acc[mt][nt][3]);
            s[mt] += __shfl_xor(s[mt], 16, 64);
            s[mt] += __shfl_xor(s[mt], 32, 64);
            if (g == 0) stat[wave * 32 + mt * 16 + lr] = s[mt];
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            mean[mt] = (stat[(wave & ~1) * 32 + mt * 16 + lr] + stat[(wave | 1) * 32 + mt * 16 + lr]) * inv_d;
            q[mt] = 0.f;
#pragma unroll
            for (int nt = 0; nt < NTH; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dd = acc[mt][nt][r] - mean[mt];
                    q[mt] = fmaf(dd, dd, q[mt]);
                }
            q[mt] += __shfl_xor(q[mt], 16, 64);
            q[mt] += __shfl_xor(q[mt], 32, 64);
            if (g == 0) stat[NW * 32 + wave * 32 + mt * 16 + lr] = q[mt];
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float qq = stat[NW * 32 + (wave & ~1) * 32 + mt * 16 + lr] + stat[NW * 32 + (wave | 1) * 32 + mt * 16 + lr];
            rstd[mt] = 1.0f / sqrtf(qq * inv_d + p.eps);
        }
    };

    // the exchange area is read one iteration before its first real contents exist: it must hold finite numbers
    for (int i = tid; i < 2 * XPAR / 16; i += 64 * NW)
        *reinterpret_cast<f32x4*>(smem + I::XCH_OFF + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ngroups = (M + TPG - 1) / TPG;
    int buf = 0;
    if ((int)blockIdx.x < ngroups)
        static_for<0, 4>([&](auto b_) { issue_part(b_, std::integral_constant<int, 4>{}, 0, 0); });
#pragma unroll 1
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int m0 = grp * TPG + pr * (16 * MT);
        int mrow[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) mrow[mt] = min(m0 + mt * 16 + lr, M - 1);

        f32x4 acc[MT][NTH];
        {   // ---- y = res + b + W_proj . a^T, this wave's 224 output channels over the full K ------------
            // a fragments are fetched one chunk (two k-steps) ahead: holding all of them would push the asm-read
            // weight fragments into scratch, and a register spilled between its ds_read and the wait is stale
            const bf16* arow[MT];
            bf16x8 afn[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                arow[mt] = A + (size_t)mrow[mt] * p.lda + 8 * g;
#pragma unroll
                for (int kl = 0; kl < 2; ++kl) afn[mt][kl] = PAIR_LDA(arow[mt] + 32 * kl);   // KS >= 2
            }
#pragma unroll
            for (int nt = 0; nt < NTH; ++nt) {
                const int c0 = cbase0 + 16 * nt;
                const f32x4 bp = load4(vproj + c0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#if PAIR_NT & 1
                    acc[mt][nt] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.res + (size_t)mrow[mt] * d + c0)) + bp;
#else
                    acc[mt][nt] = load4(p.res + (size_t)mrow[mt] * d + c0) + bp;
#endif
                }
            }
            static_for<0, I::NPC>([&](auto pc_) {
                constexpr int pc = decltype(pc_)::value;
                lds_dma_wait_barrier();
                bf16x8 af[MT][2];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int kl = 0; kl < 2; ++kl) {
                        af[mt][kl] = afn[mt][kl];
                        if constexpr (pc + 1 < I::NPC) {
                            if (2 * (pc + 1) + kl < KS)   // compile-time after unrolling; odd KS: the last chunk has one k-step
                                afn[mt][kl] = PAIR_LDA(arow[mt] + 32 * (2 * (pc + 1) + kl));
                        }
                    }
                unsigned ln;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
                const unsigned base = lds0 + buf * CB + hh * (HF * 1024) + ln * 16u;
                constexpr int BS = PAIR_BSP, NBP = (HF + BS - 1) / BS;
                constexpr int LIM = 2 * pc + 1 < KS ? I::HFP : NTH;   // fragments of this chunk that are used (odd KS: one k-step in the last)
                bf16x8 w[2][BS];
                lds_frags<0, BS>(base, w[0]);
                static_for<0, NBP>([&](auto b_) {
                    constexpr int b = decltype(b_)::value;
                    lds_wait();
                    // only fragments that are used: a register an asm read targets and nothing consumes is reused at once by the
                    // compiler, and the late LDS return would land on its new contents
                    if constexpr (b + 1 < NBP && (b + 1) * BS < LIM)
                        lds_frags<(b + 1) * BS, (LIM - (b + 1) * BS < BS ? LIM - (b + 1) * BS : BS)>(base, w[(b + 1) & 1]);
                    if constexpr (b < PDMAB(NBP)) issue_part(b_, std::integral_constant<int, PDMAB(NBP)>{}, pc + 1, buf ^ 1);   // pc + 1 <= NPC < nck
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < BS; ++i) {
                        const int fl = b * BS + i, kl = fl / NTH, ntl = fl % NTH;
                        if (fl < LIM) {
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) mma16(w[b & 1][i], af[mt][kl & 1], acc[mt][ntl]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                buf ^= 1;
            });
        }
        // ---- LayerNorm 2: this wave's channels become its K half of fc1's B operand ---------------------
        bf16x8 xn[MT][KSH];
        {
            float mean[MT], rstd[MT];
            pair_stats(acc, mean, rstd);
#pragma unroll
            for (int ksl = 0; ksl < KSH; ++ksl)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int nt = 2 * ksl + t, c0 = cbase0 + 16 * nt;
                    if (nt < NTH) {
                        const f32x4 gm = load4(vg2 + c0), bt = load4(vb2 + c0);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                xn[mt][ksl][4 * t + r] = (bf16)fmaf((acc[mt][nt][r] - mean[mt]) * rstd[mt], gm[r], bt[r]);
                    } else {   // odd KS: the empty half of the wave's last k-step
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) xn[mt][ksl][4 * t + r] = (bf16)0.f;
                    }
                }
        }
        // ---- MLP ---------------------------------------------------------------------------------------
        // iteration c: fc1 partials of chunk c | GELU of this wave's hidden tile of chunk c - 1 | fc2 of chunk c - 2
        f32x4 own[MT];      // this wave's fp32 partial of the tile it finalises (chunk c - 1)
        bf16x8 hown;        // its GELU output [mt][r] (chunk c - 2), the other half comes from the partner
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) own[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) hown[j] = (bf16)0.f;
#pragma unroll 1
        for (int c = 0; c <= nch + 1; ++c) {
            lds_wait();               // this wave's exchange writes of the previous iteration have reached LDS
#if !(TAIL_ABL & 8)
            lds_dma_wait_barrier();   // chunk landed, the partner's writes are visible, the other buffers are free
#endif
            const int next_ck = I::NPC + c + 1 < nck ? I::NPC + c + 1 : 0;
            unsigned ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            const unsigned cbase = lds0 + buf * CB;
            const unsigned base = cbase + hh * (HF * 1024) + ln * 16u;
            const unsigned xw = lds0 + I::XCH_OFF + (c & 1) * XPAR + wave * 1024 + ln * 16u;               // mine, this iteration
            const unsigned xr = lds0 + I::XCH_OFF + ((c & 1) ^ 1) * XPAR + (wave ^ 1) * 1024 + ln * 16u;   // partner's, previous
            const unsigned baddr = cbase + I::B1_OFF + hh * 64 + (ln >> 4) * 16u;
            constexpr int BS = KS % 2 ? PAIR_BSM_ODD : PAIR_BSM;
            constexpr int NB = (HF + BS - 1) / BS, F1 = 2 * KSH;   // fragments [0, F1): fc1, [F1, HFM): fc2
            constexpr int BG0 = 1, BGN = NB > 6 ? 4 : 2;   // GELU batches: 8 values over BGN batches
            constexpr int BPW = (F1 + BS - 1) / BS;     // first batch after the last fc1 fragment
            static_assert(BG0 + BGN < NB && BPW < NB && 8 % BGN == 0, "schedule");
            bf16x8 w[2][BS];
            f32x4 part[MT][2];
            bf16x8 pp, hq, hnew, hfull[MT];
            f32x4 bi;
            lds_frags<0, BS>(base, w[0]);
            pp = lds_read_b128(xr);           // partner's partial of MY tile, chunk c - 1: [mt][r] bf16
            bi = lds_f32x4(baddr);            // fc1 biases of my tile, chunk c: the accumulators of my tile start there
            hq = lds_read_b128(xr + XGELU);   // partner's GELU half of chunk c - 2
            static_for<0, NB>([&](auto b_) {
                constexpr int b = decltype(b_)::value;
                lds_wait();
#if TAIL_ABL & 16
                if constexpr (b == 0)
#endif
                if constexpr (b + 1 < NB && (b + 1) * BS < I::HFM)   // used fragments only, as in the proj loop
                    lds_frags<(b + 1) * BS, (I::HFM - (b + 1) * BS < BS ? I::HFM - (b + 1) * BS : BS)>(base, w[(b + 1) & 1]);
                if constexpr (b < PDMAB(NB)) issue_part(b_, std::integral_constant<int, PDMAB(NB)>{}, next_ck, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (b == 0) {   // fc2 operand of chunk c - 2: own half in k-slots 0..3, the partner's in 4..7
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            hfull[mt][r] = hown[4 * mt + r];
                            hfull[mt][4 + r] = hq[4 * mt + r];
                        }
                }
                if constexpr (b >= BG0 && b < BG0 + BGN) {   // GELU of chunk c - 1, my tile
#pragma unroll
                    for (int e = (8 / BGN) * (b - BG0); e < (8 / BGN) * (b - BG0 + 1); ++e) {
                        float x = own[e >> 2][e & 3] + (float)pp[e];
#if !(TAIL_ABL & 4)
                        x = gelu_t<true>(x);
#endif
                        asm volatile("" : "+v"(x));
                        hnew[e] = (bf16)x;
                    }
                }
                if constexpr (b == BG0 + BGN) lds_write_b128(xw + XGELU, hnew);
                if constexpr (b == BPW) {   // fc1 partials of chunk c are complete: the partner's tile goes out as bf16
                    bf16x8 pk;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pk[e] = (bf16)part[e >> 2][1][e & 3];
                    lds_write_b128(xw, pk);
                }
#pragma unroll
                for (int i = 0; i < BS; ++i) {
                    const int fl = b * BS + i;
                    if (fl < F1) {   // fc1 partial: fragment [ksl][tile], tile 0 = mine (seeded with the bias)
                        const int ksl = fl >> 1, t = fl & 1;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if (ksl == 0) part[mt][t] = t == 0 ? bi : f32x4{0.f, 0.f, 0.f, 0.f};
                            mma16(w[b & 1][i], xn[mt][ksl], part[mt][t]);
                        }
                    } else if (fl < I::HFM) {   // fc2 of chunk c - 2
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) mma16(w[b & 1][i], hfull[mt], acc[mt][fl - F1]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) own[mt] = part[mt][0];
            hown = hnew;
            buf ^= 1;
        }
        // ---- + fc2 bias, store; optional LayerNorm 1 of the next block -----------------------------------
#pragma unroll
        for (int nt = 0; nt < NTH; ++nt) {
            const f32x4 b = load4(vfc2 + cbase0 + 16 * nt);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] += b;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + mt * 16 + lr;
            if (m < M) {
#pragma unroll
                for (int nt = 0; nt < NTH; ++nt) {
#if PAIR_NT & 4
                    __builtin_nontemporal_store(acc[mt][nt], reinterpret_cast<f32x4*>(p.out + (size_t)m * d + cbase0 + 16 * nt));
#else
                    store4(p.out + (size_t)m * d + cbase0 + 16 * nt, acc[mt][nt]);
#endif
                }
            }
        }
        if (p.xn_next) {   // kernel-uniform
            bf16* __restrict__ xo = static_cast<bf16*>(p.xn_next);
            float mean[MT], rstd[MT];
            pair_stats(acc, mean, rstd);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = m0 + mt * 16 + lr;
                if constexpr (NTH % 2 == 1) {   // the odd column tile: 4 channels per lane
                    constexpr int nt = NTH - 1;
                    const int c0 = cbase0 + 16 * nt;
                    const f32x4 gm = load4(p.gamma_n + c0), bt = load4(p.beta_n + c0);
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (bf16)fmaf((acc[mt][nt][r] - mean[mt]) * rstd[mt], gm[r], bt[r]);
                    if (m < M) *reinterpret_cast<bf16x4*>(xo + (size_t)m * p.ldn + c0) = o;
                }
#pragma unroll
                for (int nt = 0; nt + 1 < NTH; nt += 2) {
                    f32x4 v[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int c0 = cbase0 + 16 * (nt + t);
                        const f32x4 gm = load4(p.gamma_n + c0), bt = load4(p.beta_n + c0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[t][r] = fmaf((acc[mt][nt + t][r] - mean[mt]) * rstd[mt], gm[r], bt[r]);
                    }
                    store_bf16_tile_pair(xo + (size_t)min(m, M - 1) * p.ldn + 16 * NTH * hh + 16 * nt, g, v[0], v[1], m < M);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifndef TAIL_448_PAIR
#define TAIL_448_PAIR 1
#endif
// d = 224 as wave pairs (tailpair_kernel<7, 14, NW>): built and parity-clean, but slower than the four-waves-per-SIMD kernel
// (stand-alone, 262144 tokens: 432 us with one 8-wave workgroup per CU, 432 us with two 4-wave workgroups, against 403 us):
// half the MFMAs per chunk barrier of the 448-wide case, 33 % more VALU instructions (partial exchange), 10 % more MFMA
// work (the half-empty k-step).  Kept selectable for measurement.
#ifndef TAIL_224_PAIR
#define TAIL_224_PAIR 0
#endif
#ifndef TAIL_224_NW
#define TAIL_224_NW 4
#endif
#ifndef TAIL_224_MT
#define TAIL_224_MT 1   // 16-token tiles per wave of the 224-wide kernel
#endif

// supported widths: exactly 224 (KS 7) and 448 (KS 14), the Hiera-B+ stage-2 / stage-3 widths
static int tail_variant(int d) {
    if (d == 224) return 0;
    if (d == 448) return 1;
    return -1;
}
bool tailblock_supported(int d) { return tail_variant(d) >= 0; }
int tailblock_chunks(int d) { return d / 8; }   // 4d hidden units in chunks of 32
size_t tailblock_image_bytes(int d) {
    const int nch = tailblock_chunks(d);
    switch (tail_variant(d)) {
#if TAIL_224_PAIR
        case 0: return (size_t)(PairImg<7, 14>::NPC + nch + 2) * PairImg<7, 14>::CHUNK_B;
#else
        case 0: return (size_t)(TailImg<7, 14>::NPC + nch + 1) * TailImg<7, 14>::CHUNK_B;
#endif
#if TAIL_448_PAIR
        case 1: return (size_t)(PairImg<14, 28>::NPC + nch + 2) * PairImg<14, 28>::CHUNK_B;
#else
        case 1: return (size_t)(TailImg<14, 28>::NPC + nch + 1) * TailImg<14, 28>::CHUNK_B;
#endif
    }
    return 0;
}
size_t tailblock_vec_floats(int d) {
    switch (tail_variant(d)) {
        case 0: return 4 * TailImg<7, 14>::VEC;
        case 1: return 4 * TailImg<14, 28>::VEC;
    }
    return 0;
}

int launch_pack_tailblock(const float* wproj, const float* bproj, const float* g2, const float* be2, const float* w1,
                          const float* b1, const float* w2, const float* b2, int d, void* img, float* vec,
                          hipStream_t s) {
    LSAM_REQUIRE(wproj && bproj && g2 && be2 && w1 && b1 && w2 && b2 && img && vec, LSAM_ERR_BAD_ARG,
                 "pack_tailblock: null argument");
    const int v = tail_variant(d), nch = tailblock_chunks(d);
    LSAM_REQUIRE(v >= 0, LSAM_ERR_UNSUPPORTED, "pack_tailblock: width %d", d);
    char* im = static_cast<char*>(img);
    if (v == 0)
#if TAIL_224_PAIR
        hipLaunchKernelGGL((pack_tailpair_kernel<7, 14>), dim3(256), dim3(256), 0, s, wproj, bproj, g2, be2, w1, b1, w2,
                           b2, d, nch, im, vec);
#else
        hipLaunchKernelGGL((pack_tailblock_kernel<7, 14>), dim3(256), dim3(256), 0, s, wproj, bproj, g2, be2, w1, b1, w2,
                           b2, d, nch, im, vec);
#endif
    else
#if TAIL_448_PAIR
        hipLaunchKernelGGL((pack_tailpair_kernel<14, 28>), dim3(256), dim3(256), 0, s, wproj, bproj, g2, be2, w1, b1, w2,
                           b2, d, nch, im, vec);
#else
        hipLaunchKernelGGL((pack_tailblock_kernel<14, 28>), dim3(256), dim3(256), 0, s, wproj, bproj, g2, be2, w1, b1, w2,
                           b2, d, nch, im, vec);
#endif
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

template <int KS, int NTO, int MT, int NW, int WPC> static int launch_tail(const TailBlockArgs& a, hipStream_t s) {
    constexpr int LDSB = 2 * TailImg<KS, NTO>::CHUNK_B;
    static_assert(WPC * LDSB <= 160 * 1024, "LDS budget");
    auto kern = tailblock_kernel<KS, NTO, MT, NW, WPC>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        attr_set = true;
    }
    const int ngroups = cdiv(a.M, 16 * MT * NW);
    const int grid = ngroups < 256 * WPC ? ngroups : 256 * WPC;   // persistent
    const int rec = prof_begin(s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), LDSB, s, a);
    LSAM_LAUNCH_CHECK();
    // algorithmic work: proj 2 M d^2 + fc1 8 M d^2 + fc2 8 M d^2; cfg code 9000 + d
    prof_end(rec, s, 18.0 * a.M * a.d * a.d, 9000 + a.d, a.M, a.d, a.d);
    return LSAM_OK;
}

template <int KS, int NTO, int NW> static int launch_tailpair(const TailBlockArgs& a, hipStream_t s) {
    using I = PairImg<KS, NTO>;
    constexpr int LDSB = I::lds_bytes(NW), WGS = 8 / NW;
    static_assert(WGS * LDSB <= 160 * 1024, "LDS budget");
    auto kern = tailpair_kernel<KS, NTO, NW>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        attr_set = true;
    }
    const int ngroups = cdiv(a.M, 16 * NW);
    const int grid = ngroups < 256 * WGS ? ngroups : 256 * WGS;   // persistent: 8 waves per CU in one or two workgroups
    const int rec = prof_begin(s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), LDSB, s, a);
    LSAM_LAUNCH_CHECK();
    prof_end(rec, s, 18.0 * a.M * a.d * a.d, 9000 + a.d, a.M, a.d, a.d);
    return LSAM_OK;
}

int launch_tailblock(const TailBlockArgs& a, hipStream_t s) {
    LSAM_REQUIRE(a.a && a.res && a.out && a.img && a.vec, LSAM_ERR_BAD_ARG, "tailblock: null argument");
    LSAM_REQUIRE(a.M > 0 && a.nch == tailblock_chunks(a.d) && a.lda >= a.d && a.lda % 8 == 0, LSAM_ERR_BAD_ARG,
                 "tailblock: bad shape");
    LSAM_REQUIRE(aligned16(a.a) && aligned16(a.res) && aligned16(a.out) && aligned16(a.img) && aligned16(a.vec),
                 LSAM_ERR_BAD_ARG, "tailblock: pointers must be 16-byte aligned");
    if (a.xn_next)
        LSAM_REQUIRE(a.gamma_n && a.beta_n && a.ldn >= a.d && a.ldn % 8 == 0 && aligned16(a.gamma_n) && aligned16(a.beta_n) &&
                         aligned16(a.xn_next),
                     LSAM_ERR_BAD_ARG, "tailblock: next-LayerNorm output needs gamma, beta, 16-byte alignment and ldn %% 8 == 0");
    switch (tail_variant(a.d)) {
#if TAIL_224_PAIR
        case 0: return launch_tailpair<7, 14, TAIL_224_NW>(a, s);   // wave pairs: a wave keeps 112 channels = 3 1/2 k-steps of fc1's K
#else
        case 0: return launch_tail<7, 14, TAIL_224_MT, 8, 2>(a, s);    // 8 waves x 16 tokens, two waves per SIMD (254 registers)
#endif
#if TAIL_448_PAIR
        case 1: return launch_tailpair<14, 28, 8>(a, s);     // 8 waves, wave pairs share 32 tokens: two waves per SIMD
#elif defined(TAIL_448_MT1)
        case 1: return launch_tail<14, 28, 1, 8, 1>(a, s);
#else
        case 1: return launch_tail<14, 28, 2, 4, 1>(a, s);   // 4 waves x 32 tokens: one wave per SIMD, 512 registers
#endif
    }
    set_error("tailblock: width %d not supported", a.d);
    return LSAM_ERR_UNSUPPORTED;
}

}  // namespace lsam
