// Attention half of a 4x4-window block (Hiera-B+ stage 2: dim 224, 4 heads of 56; MultiScaleAttention.forward,
// hieradet.py:56-81) as ONE kernel, qkv projection included:
//
//     q | k | v = norm1(x) . W_qkv^T + b_qkv       (the 16 tokens of a window)
//     out       = softmax(q k^T / sqrt(hd)) v      (per head)
//
// and, for the q-pool stage-entry block behind it (B+ block 5: 2x2 pooled queries from the qkv GEMM's pooling epilogue,
// hieradet.py:69-71), the k | v projection + attention of the window's 4 pooled queries.
//
// Un-fused, qkv of such a block (262144 x 672 bf16 = 352 MB at batch 16) is written by a projection that is bound by its
// stores and read back once by an attention kernel that is bound by those reads: 150 + 134 us around 83 GFLOP.  A window of
// 16 tokens is exactly one MFMA tile, so here a wave projects, scores and mixes a window without any of it leaving registers:
//
//   * the window's LN1 rows are read once as operand fragments (a lane's 16-byte load IS its k-contiguous fragment);
//   * Q^T = W_q . X^T and K^T = W_k . X^T with the ROWS of W_q / W_k (the head's channels) permuted at pack time, so that a
//     lane's accumulator values of two row tiles are exactly its 8 k-slots of the operands of S^T = K . Q^T (the head's
//     channels are contracted, so any order works as long as q and k use the same one);
//   * V = X . W_v^T in natural order: a lane's four accumulator values are key slots 8 g + 0..3 of the A operand of
//     O^T = V^T . P^T, and its four score registers are the matching slots of P^T (slots 4..7 are zero in both);
//   * the softmax over a query's 16 keys is 4 registers x the 4 lane groups.
//
// The weights of one head (84 KiB: 3 parts x 4 tiles x 7 k-steps of 1 KiB fragments; two heads of k | v for the pooled
// variant) stay in LDS for the lifetime of a workgroup; the workgroups of the different heads that walk the same rows are
// neighbours on one XCD, so the rows come from HBM once.  Persistent, one 8-wave workgroup per CU, no barrier after the fill.
#include <type_traits>

#include "kernels.hpp"

#ifndef QA_ABL
#define QA_ABL 0   // timing-only ablation builds (tools/build_abl.sh): 1 = no row loads, 2 = no LDS weight reads, 4 = no stores
#endif

namespace lsam {

namespace {
constexpr int QA_KS = 7;   // 32-wide k-steps over the input channels (dim <= 224)

__host__ __device__ inline int qa_parts(bool poolq) { return poolq ? 2 : 3; }
__host__ __device__ inline size_t qa_head_bytes(bool poolq) { return (size_t)qa_parts(poolq) * 4 * QA_KS * 1024; }
inline int qa_heads_per_wg(bool poolq, int nh) { return (poolq && nh % 2 == 0) ? 2 : 1; }   // <= 112 KiB of LDS
}  // namespace

bool qkvattn4_supported(int H, int W, int ws, int pool, int nh, int hd, int dim, int dout) {
    if (!(ws == 4 && H % 4 == 0 && W % 4 == 0 && nh >= 1 && nh <= 32 && hd % 8 == 0 && hd >= 8 && hd <= 64 && dim % 8 == 0 &&
          dim >= 8 && dim <= 32 * QA_KS && dout == nh * hd))
        return false;
    return pool == 1 ? dim == dout : pool == 2;
}

// image: [head][part: (q,) k = permuted rows, v = natural][tile 0..3][k-step] fragments of [64 lanes][8 bf16], then fp32
// [head][part][64] biases in the same row order (zero beyond hd).
size_t qkvattn4_image_bytes(int nh, bool poolq) { return (size_t)nh * (qa_head_bytes(poolq) + qa_parts(poolq) * 64 * sizeof(float)); }

__global__ void pack_qkvattn4_kernel(const float* __restrict__ wqkv, const float* __restrict__ bqkv, int dim, int nh, int hd,
                                     int poolq, bf16* __restrict__ img, float* __restrict__ bias) {
    const int q = nh * hd, np = qa_parts(poolq), p0 = poolq ? 1 : 0;
    const long total = (long)nh * np * 4 * QA_KS * 512;
    const long nb = (long)nh * np * 64;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total + nb; i += (long)gridDim.x * blockDim.x) {
        if (i < total) {
            const int f = (int)(i >> 9), lane = (int)((i & 511) >> 3), j = (int)(i & 7);
            const int ks = f % QA_KS, tile = (f / QA_KS) % 4, part = (f / (QA_KS * 4)) % np, h = f / (QA_KS * 4 * np);
            const int g = lane >> 4, lr = lane & 15;
            const int k = 32 * ks + 8 * g + j;
            // q, k: permuted row 16 tile + lr holds channel d = 32 (tile / 2) + 8 (lr / 4) + 4 (tile % 2) + lr % 4
            const bool perm = part < np - 1;
            const int d = perm ? 32 * (tile >> 1) + 8 * (lr >> 2) + 4 * (tile & 1) + (lr & 3) : 16 * tile + lr;
            float v = 0.f;
            if (d < hd && k < dim) v = wqkv[(size_t)((part + p0) * q + h * hd + d) * dim + k];
            img[i] = (bf16)v;
        } else {
            const int t = (int)(i - total), p = t % 64, part = (t / 64) % np, h = t / (64 * np);
            const int tile = p >> 4, lr = p & 15;
            const bool perm = part < np - 1;
            const int d = perm ? 32 * (tile >> 1) + 8 * (lr >> 2) + 4 * (tile & 1) + (lr & 3) : p;
            bias[t] = d < hd ? bqkv[(part + p0) * q + h * hd + d] : 0.f;
        }
    }
}

int launch_pack_qkvattn4(const float* wqkv, const float* bqkv, int dim, int nh, int hd, bool poolq, void* img, hipStream_t s) {
    LSAM_REQUIRE(wqkv && bqkv && img, LSAM_ERR_BAD_ARG, "qkvattn4_pack: null argument");
    LSAM_REQUIRE(qkvattn4_supported(4, 4, 4, 2, nh, hd, dim, nh * hd), LSAM_ERR_UNSUPPORTED,
                 "qkvattn4_pack: %d heads of %d over %d channels not supported", nh, hd, dim);
    char* base = static_cast<char*>(img);
    hipLaunchKernelGGL(pack_qkvattn4_kernel, dim3(256), dim3(256), 0, s, wqkv, bqkv, dim, nh, hd, poolq ? 1 : 0,
                       reinterpret_cast<bf16*>(base), reinterpret_cast<float*>(base + (size_t)nh * qa_head_bytes(poolq)));
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// TT windows per wave and pass (a weight fragment read from LDS feeds TT MFMAs); HPW heads per workgroup
template <int TT, bool POOLQ, int NW>
__global__ __launch_bounds__(64 * NW) void qkvattn4_kernel(const QkvAttn4Args p, const int n_win, const int n_hg, const int hpw,
                                                       const int nrow_wg) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NP = POOLQ ? 2 : 3, KS = QA_KS;
    constexpr int HB = NP * 4 * KS * 1024;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, lr = lane & 15;
    const int hd = p.hd, dim = p.dim;
    // workgroup -> (head group, row slot): the n_hg workgroups of a row slot are consecutive on ONE XCD
    const int xcd = blockIdx.x & 7, j_ = blockIdx.x >> 3;
    const int hg = j_ % n_hg, rslot = (j_ / n_hg) * 8 + xcd;
    if (rslot >= nrow_wg) return;   // workgroup-uniform, before the barrier
    const int h0 = hg * hpw;
    // ---- this head group's weight image and biases -> LDS, once per workgroup ---------------------------------------------
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.img + (size_t)h0 * HB);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        const int n16 = hpw * HB / 16;
        for (int i = threadIdx.x; i < n16; i += 64 * NW) dst[i] = src[i];
        const float* bsrc = reinterpret_cast<const float*>(p.img + (size_t)p.nh * HB) + (size_t)h0 * NP * 64;
        float* bdst = reinterpret_cast<float*>(lds + (size_t)hpw * HB);
        for (int i = threadIdx.x; i < hpw * NP * 64; i += 64 * NW) bdst[i] = bsrc[i];
    }
    __syncthreads();
    const float* bl = reinterpret_cast<const float*>(lds + (size_t)hpw * HB);
    const int wpr = p.W / 4, wpi = (p.H / 4) * wpr;
    const int Hq = p.H / 2, Wq = p.W / 2;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#if QA_ABL & 2
    const bf16x8 abl_w = __builtin_bit_cast(bf16x8, make_uint4(lane, lane + 1, lane + 2, lane + 3));
#define QA_W(ptr, idx) abl_w
#else
#define QA_W(ptr, idx) (ptr)[idx]
#endif
    const int n_items = (n_win + TT - 1) / TT;

    // this lane's token of window wi (clamped: a window beyond the last is computed and never stored)
    auto win_tok = [&](int wi, size_t& tok, size_t& qtok) {
        const int w = min(wi, n_win - 1);
        const int n = w / wpi, wy = (w % wpi) / wpr, wx = w % wpr;
        tok = ((size_t)n * p.H + wy * 4 + (lr >> 2)) * p.W + wx * 4 + (lr & 3);           // token lr: row lr / 4, column lr % 4
        qtok = ((size_t)n * Hq + wy * 2 + ((lr >> 1) & 1)) * Wq + wx * 2 + (lr & 1);      // POOLQ: pooled query lr % 4
    };
    // ---- the windows' LN1 rows as operand fragments ----------------------------------------------------------------------
    auto load_rows = [&](int item, bf16x8 (&xf)[TT][KS]) {
#if QA_ABL & 1
        for (int t = 0; t < TT; ++t)
            for (int ks = 0; ks < KS; ++ks) xf[t][ks] = __builtin_bit_cast(bf16x8, make_uint4(item, lane, t, ks));
        return;
#endif
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            size_t tok, qtok;
            win_tok(item * TT + t, tok, qtok);
            const bf16* row = p.xn + tok * p.ldx;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k0 = 32 * ks + 8 * g;
                uint4 v = *reinterpret_cast<const uint4*>(row + min(k0, dim - 8));   // clamped, branch-free
                if (k0 >= dim) v = make_uint4(0, 0, 0, 0);
                xf[t][ks] = __builtin_bit_cast(bf16x8, v);
            }
        }
    };
    auto process = [&](int item, const bf16x8 (&xf)[TT][KS]) {
        size_t tok[TT], qtok[TT];
        bool valid[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            valid[t] = item * TT + t < n_win;
            win_tok(item * TT + t, tok[t], qtok[t]);
        }
#pragma unroll 1
        for (int hh = 0; hh < hpw; ++hh) {
            const int h = h0 + hh;
            const bf16x8* wl = reinterpret_cast<const bf16x8*>(lds + (size_t)hh * HB) + lane;
            const float* bh = bl + hh * NP * 64;
            // ---- four weight tiles x TT windows per pass: 4 TT independent accumulators, a k-step's four fragments read one
            //      k-step ahead of their MFMAs.  SWAP: X is the A operand (rows = tokens), else the weight tile is ------------
            auto proj4 = [&](const bf16x8* const (&wt)[4], f32x4 (&acc)[4][TT], auto swap_) {
                constexpr bool SWAP = decltype(swap_)::value;
                bf16x8 wf[2][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[0][i] = QA_W(wt[i], 0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 1 < KS) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) wf[(ks + 1) & 1][i] = QA_W(wt[i], (ks + 1) * 64);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int t = 0; t < TT; ++t) {
                            if constexpr (SWAP) mma16(xf[t][ks], wf[ks & 1][i], acc[i][t]);
                            else mma16(wf[ks & 1][i], xf[t][ks], acc[i][t]);
                        }
                }
            };
            // permuted-row tiles (q, k): rows = channels, columns = tokens; tiles 2 s, 2 s + 1 are k-step s of operand slots
            auto tile_w = [&](int part, int tile) { return wl + (size_t)(part * 4 + tile) * KS * 64; };
            auto pack2 = [&](const f32x4& lo, const f32x4& hi) {
                bf16x8 r;
#pragma unroll
                for (int j = 0; j < 4; ++j) { r[j] = (bf16)lo[j]; r[4 + j] = (bf16)hi[j]; }
                return r;
            };
            bf16x8 qf[TT][2], kf[TT][2];
            if constexpr (POOLQ) {
                // pooled queries (fp32, from the GEMM's pooling epilogue) in natural channel order: column lr = query lr % 4;
                // requested in front of the K projection, converted behind it
                f32x4 qa[TT][2], qb[TT][2];
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int d0 = 32 * s + 8 * g;
                        qa[t][s] = zero4;
                        qb[t][s] = zero4;
                        if (d0 < hd) {
                            const float* qs = p.qp + qtok[t] * p.ldqp + h * hd + d0;
                            qa[t][s] = load4(qs);
                            qb[t][s] = load4(qs + 4);
                        }
                    }
                f32x4 acc[4][TT];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 b = load4(bh + 16 * i + 4 * g);
#pragma unroll
                    for (int t = 0; t < TT; ++t) acc[i][t] = b;
                }
                const bf16x8* const wt[4] = {tile_w(0, 0), tile_w(0, 1), tile_w(0, 2), tile_w(0, 3)};
                proj4(wt, acc, std::false_type{});
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        kf[t][s] = pack2(acc[2 * s][t], acc[2 * s + 1][t]);
                        qf[t][s] = pack2(qa[t][s], qb[t][s]);
                    }
            } else {
#pragma unroll
                for (int s = 0; s < 2; ++s) {   // q and k tiles 2 s, 2 s + 1 together
                    f32x4 acc[4][TT];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 b = load4(bh + (i >> 1) * 64 + 16 * (2 * s + (i & 1)) + 4 * g);
#pragma unroll
                        for (int t = 0; t < TT; ++t) acc[i][t] = b;
                    }
                    const bf16x8* const wt[4] = {tile_w(0, 2 * s), tile_w(0, 2 * s + 1), tile_w(1, 2 * s), tile_w(1, 2 * s + 1)};
                    proj4(wt, acc, std::false_type{});
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        qf[t][s] = pack2(acc[0][t], acc[1][t]);
                        kf[t][s] = pack2(acc[2][t], acc[3][t]);
                    }
                }
            }
            // ---- S^T = K . Q^T (lane: query lr, keys 4 g .. 4 g + 3), softmax over the window's 16 keys ----------------------
            bf16x8 pf[TT];
            float linv[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                f32x4 sc = zero4;
                mma16(kf[t][0], qf[t][0], sc);
                mma16(kf[t][1], qf[t][1], sc);
                float mx = -1.0e30f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sc[r] *= p.c_log2;
                    mx = fmaxf(mx, sc[r]);
                }
                mx = group4_max(mx);
                float l = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(sc[r] - mx);
                    l += e;
                    pf[t][r] = (bf16)e;
                    pf[t][4 + r] = (bf16)0.f;
                }
                linv[t] = 1.0f / group4_sum(l);
            }
            // ---- V = X . W_v^T + b_v (rows = tokens, columns = channels), all four channel tiles; O^T = V^T . P^T ------------
            {
                f32x4 acc[4][TT];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float b = bh[(NP - 1) * 64 + 16 * i + lr];
#pragma unroll
                    for (int t = 0; t < TT; ++t) acc[i][t] = f32x4{b, b, b, b};
                }
                const bf16x8* const wt[4] = {tile_w(NP - 1, 0), tile_w(NP - 1, 1), tile_w(NP - 1, 2), tile_w(NP - 1, 3)};
                proj4(wt, acc, std::true_type{});
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bf16x8 vf = pack2(acc[i][t], zero4);   // key slots 8 g + 0..3 = tokens 4 g + 0..3; 4..7 empty
                        f32x4 o = zero4;
                        mma16(vf, pf[t], o);
                        // lane holds O[query lr][channel 16 i + 4 g .. + 3]
                        const int d0 = 16 * i + 4 * g;
                        const bool st = valid[t] && d0 < hd && (!POOLQ || lr < 4);
                        bf16* orow = p.out + (POOLQ ? qtok[t] : tok[t]) * p.ldo + h * hd + d0;
                        if (st && (!(QA_ABL & 4) || o[0] == 123.456f)) store4(orow, o * linv[t]);
                    }
            }
        }
    };
    // (a second register set of row fragments, with the next item's rows in flight during this one, was measured: 120 us
    //  against 110 -- the other wave of the SIMD already covers the loads, and the extra registers cost LDS-read hoisting)
#pragma unroll 1
    for (int item = rslot * NW + wave; item < n_items; item += nrow_wg * NW) {
        bf16x8 xf[TT][KS];
        load_rows(item, xf);
        process(item, xf);
    }
}

template <int TT, bool POOLQ> static int launch_qa(const QkvAttn4Args& a, int n_win, hipStream_t s) {
    const int hpw = qa_heads_per_wg(POOLQ, a.nh), n_hg = a.nh / hpw;
    const size_t lds = (size_t)hpw * (qa_head_bytes(POOLQ) + qa_parts(POOLQ) * 64 * sizeof(float));
    constexpr int NW = QKVATTN4_NW;
    auto kern = qkvattn4_kernel<TT, POOLQ, NW>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    // one workgroup per CU (LDS); blockIdx % 8 picks the XCD, whose 32 CUs hold 32 / n_hg row slots (a workgroup beyond that
    // would run after the others)
    const int want = cdiv(cdiv(n_win, TT), NW);
    int nrow_wg = 8 * (32 / n_hg);
    if (nrow_wg < 1) nrow_wg = 1;
    if (nrow_wg > want) nrow_wg = want;
    const int grid = 8 * cdiv(nrow_wg, 8) * n_hg;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, s, a, n_win, n_hg, hpw, nrow_wg);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

int launch_qkvattn4(const QkvAttn4Args& a, hipStream_t s) {
    const bool poolq = a.qp != nullptr;
    LSAM_REQUIRE(a.xn && a.img && a.out && a.N > 0, LSAM_ERR_BAD_ARG, "qkvattn4: null/empty argument");
    LSAM_REQUIRE(qkvattn4_supported(a.H, a.W, 4, poolq ? 2 : 1, a.nh, a.hd, a.dim, poolq ? a.nh * a.hd : a.dim), LSAM_ERR_UNSUPPORTED,
                 "qkvattn4: grid %dx%d, %d heads of %d over %d channels not supported", a.H, a.W, a.nh, a.hd, a.dim);
    LSAM_REQUIRE(a.ldx >= a.dim && a.ldx % 8 == 0 && a.ldo >= a.nh * a.hd && a.ldo % 4 == 0 && aligned16(a.xn) && aligned16(a.img) &&
                     (reinterpret_cast<uintptr_t>(a.out) & 7) == 0 && (!poolq || (a.ldqp >= a.nh * a.hd && a.ldqp % 4 == 0 && aligned16(a.qp))),
                 LSAM_ERR_BAD_ARG, "qkvattn4: leading dimensions / alignment");
    LSAM_REQUIRE((double)a.N * a.H * a.W < 2.0e9, LSAM_ERR_UNSUPPORTED, "qkvattn4: too many tokens");
    const int n_win = a.N * (a.H / 4) * (a.W / 4);
    const double M = (double)a.N * a.H * a.W;
    // algorithmic FLOPs: the projections of every token + the (16 or 4) x 16 attention of every window and head
    const double nq = poolq ? 4.0 : 16.0;
    ProfScope ps(s, 2.0 * M * a.dim * qa_parts(poolq) * a.nh * a.hd + 4.0 * n_win * nq * 16.0 * a.nh * a.hd, PROF_QKVATTN4, (int)M,
                 qa_parts(poolq) * a.nh * a.hd, a.dim);
    return poolq ? launch_qa<QKVATTN4_TT, true>(a, n_win, s) : launch_qa<QKVATTN4_TT, false>(a, n_win, s);
}

}  // namespace lsam
