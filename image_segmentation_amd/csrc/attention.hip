// Windowed / global multi-head attention for Hiera, forward (MFMA, flash-style) and the
// LoRA-block backward.
//
// The kernels read the qkv GEMM output in its natural [N,H,W,3*nh*hd] layout and write the
// un-partitioned, cropped [N,H',W',nh*hd] result, so the reference's window_partition /
// window_unpartition / do_pool copies (backbones/utils.py:16-60, hieradet.py:25-36) never
// materialise.  Semantics that are easy to lose:
//  * zero-pad tokens are created AFTER LN1 and BEFORE qkv, so their q/k/v equal the qkv bias and
//    they take part as keys/values in the un-masked softmax (hieradet.py:136,146,70-74);
//  * q-pooling is a 2x2 max over q only, inside each window (hieradet.py:64-67);
//  * only real (un-cropped) query positions are computed.
//
// Forward dataflow per workgroup = one (window, head, query block):  K/V tiles of 32 keys are
// staged to LDS (K row-major XOR-swizzled; V transposed).  S^T = K.Q^T is formed with the key on
// the MFMA row index, so each lane owns ONE query (column l&15) and 8 keys per tile: softmax
// needs only two xor-shuffles per tile, and the probabilities are already the B-operand fragment
// of O^T = V^T.P^T (k-slot 8g+j  <->  key 4g+j for j<4, 16+4g+(j-4) otherwise).
#include "kernels.hpp"

namespace lsam {

constexpr int KT = 32;  // keys per tile

template <typename T> struct VtLayout;  // transposed V tile: [d][32 keys] + pad
template <> struct VtLayout<bf16> { static constexpr int RB = 72; };
template <> struct VtLayout<float> { static constexpr int RB = 144; };

template <typename T> __device__ __forceinline__ void load8f(const T* p, float* v);
template <> __device__ __forceinline__ void load8f<bf16>(const bf16* p, float* v) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
}
template <> __device__ __forceinline__ void load8f<float>(const float* p, float* v) {
    const f32x4 a = load4(p), b = load4(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}
template <typename T> __device__ __forceinline__ typename Tr<T>::frag frag_from_f(const float* v) {
    typename Tr<T>::frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = from_f<T>(v[j]);
    return f;
}

// The 2x2 max-pool of q (do_pool on the q part of qkv, hieradet.py:63-67) for 8 channels of one pooled query: the
// four source rows are requested TOGETHER (clamped addresses, out-of-grid positions replaced by the pad token's
// q = bias afterwards).  Written as a loop over the pool window, each load waited for its own round trip before
// the next was issued: four dependent HBM latencies per k-step in front of every small-window workgroup.
template <typename T, typename BiasLoad>
__device__ __forceinline__ void load_q_pool2(const T* __restrict__ qkv, size_t img_tok, int ldqkv, int H, int W, int y0, int x0,
                                             int col, BiasLoad&& bias8, float* v) {
    float t[4][8];
    bool in[4];
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
        const int y = y0 + (ab >> 1), x = x0 + (ab & 1);
        in[ab] = y < H && x < W;
        load8f<T>(qkv + (img_tok + (size_t)min(y, H - 1) * W + min(x, W - 1)) * ldqkv + col, t[ab]);
    }
    if (!(in[0] && in[1] && in[2] && in[3])) {   // rare: only windows that hang over the grid
        float b[8];
        bias8(b);
#pragma unroll
        for (int ab = 0; ab < 4; ++ab)
            if (!in[ab]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) t[ab][j] = b[j];
            }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaxf(t[0][j], t[1][j]), fmaxf(t[2][j], t[3][j]));
}

struct WinGeom {
    int wsh, wsw, nwy, nwx;   // window size (rows, cols) on the input grid, windows per image
    int qh, qw, Ho, Wo;       // query window size and output grid
};
__host__ __device__ inline WinGeom make_geom(int H, int W, int ws, int pool) {
    WinGeom g;
    g.wsh = ws > 0 ? ws : H;
    g.wsw = ws > 0 ? ws : W;
    g.nwy = (H + g.wsh - 1) / g.wsh;
    g.nwx = (W + g.wsw - 1) / g.wsw;
    g.qh = g.wsh / pool;
    g.qw = g.wsw / pool;
    g.Ho = H / pool;
    g.Wo = W / pool;
    return g;
}

// NWH = 1: the waves of a workgroup split the queries of one (window, head).
// NWH = 4: windows with at most 16 queries (Hiera stage 2 and its transitions): every wave takes its
//          own HEAD of the same window, with a private K/V tile in LDS -- four times fewer, four times
//          larger workgroups than one 64-thread workgroup per (window, head), and the four heads' q/k/v
//          slices of a token are adjacent in memory.
template <typename T, int QT, int HDP, int NWH>
__global__ void attn_fwd_kernel(const AttnArgs p, const float c_log2) {
    constexpr int EPC = Tr<T>::EPC, CPF = Tr<T>::CPF;
    constexpr int KSD = HDP / 32, DT = HDP / 16;
    constexpr int RB = HDP * (int)sizeof(T), CPR = RB / 16;  // K tile row bytes / chunks
    constexpr int VRB = VtLayout<T>::RB;
    using frag = typename Tr<T>::frag;

    __shared__ __attribute__((aligned(16))) char ks_all[NWH * KT * RB];
    __shared__ __attribute__((aligned(16))) char vt_all[NWH * HDP * VRB];

    const int lane = threadIdx.x & 63, wave_id = threadIdx.x >> 6;
    const int hw = NWH > 1 ? wave_id : 0;                      // head slot of this wave
    const int wave = NWH > 1 ? 0 : wave_id;                    // query-tile slot of this wave
    const int tid = NWH > 1 ? lane : (int)threadIdx.x, nthr = NWH > 1 ? 64 : (int)blockDim.x;   // staging team
    char* ks_ = ks_all + hw * KT * RB;
    char* vt_ = vt_all + hw * HDP * VRB;
    const int g = lane >> 4, lr = lane & 15;
    const int H = p.H, W = p.W, nh = p.nh, hd = p.hd, dout = nh * hd;
    const WinGeom G = make_geom(H, W, p.ws, p.pool);
    const bool hok = (int)blockIdx.y * NWH + hw < nh;          // idle head slots still join the barriers
    const int h = hok ? blockIdx.y * NWH + hw : nh - 1;
    const int win = blockIdx.x % (G.nwy * G.nwx), n = blockIdx.x / (G.nwy * G.nwx);
    const int wy = win / G.nwx, wx = win % G.nwx;
    const int rh = min(G.qh, G.Ho - wy * G.qh), rw = min(G.qw, G.Wo - wx * G.qw);
    const int nq = rh * rw, nk = G.wsh * G.wsw;
    const int qblock = (NWH > 1 ? 1 : (int)(blockDim.x >> 6)) * QT * 16;
    const int qbase = blockIdx.z * qblock;
    if (qbase >= nq) return;  // uniform for the workgroup, before any barrier

    const T* __restrict__ qkv = static_cast<const T*>(p.qkv);
    const float* __restrict__ bias = p.bias;
    const size_t img_tok = (size_t)n * H * W;

    // ---- Q fragments (kept in registers for the whole kernel) -------------------------------
    frag qf[QT][KSD];
    int qtok[QT];  // output token index or -1
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qi = qbase + (wave * QT + qt) * 16 + lr;
        const bool qv = qi < nq && hok;
        const int iy = qv ? qi / rw : 0, ix = qv ? qi % rw : 0;
        qtok[qt] = qv ? (int)(((size_t)n * G.Ho + (wy * G.qh + iy)) * G.Wo + (wx * G.qw + ix)) : -1;
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) {
            const int d0 = ks * 32 + g * 8;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if (qv && d0 < hd) {
                if (p.qpool) {   // pooled in the qkv GEMM's epilogue: max is monotonic under the bf16 rounding, same bits
                    const float* qs = p.qpool + (size_t)qtok[qt] * p.ldqp + h * hd + d0;
                    const f32x4 q0 = load4(qs), q1 = load4(qs + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = q0[j]; v[4 + j] = q1[j]; }
                } else if (p.pool == 1) {
                    const size_t tok = img_tok + (size_t)(wy * G.wsh + iy) * W + (wx * G.wsw + ix);
                    load8f<T>(qkv + tok * p.ldqkv + h * hd + d0, v);
                } else if (p.pool == 2) {
                    load_q_pool2<T>(qkv, img_tok, p.ldqkv, H, W, wy * G.wsh + iy * 2, wx * G.wsw + ix * 2, h * hd + d0,
                                    [&](float* b) {
#pragma unroll
                                        for (int j = 0; j < 8; ++j) b[j] = to_f(from_f<T>(bias[h * hd + d0 + j]));
                                    }, v);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = -3.0e38f;
                    for (int a = 0; a < p.pool; ++a)
                        for (int b = 0; b < p.pool; ++b) {
                            const int y = wy * G.wsh + iy * p.pool + a, x = wx * G.wsw + ix * p.pool + b;
                            float t[8];
                            if (y < H && x < W) {
                                load8f<T>(qkv + (img_tok + (size_t)y * W + x) * p.ldqkv + h * hd + d0, t);
                            } else {
#pragma unroll
                                for (int j = 0; j < 8; ++j) t[j] = to_f(from_f<T>(bias[h * hd + d0 + j]));
                            }
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], t[j]);
                        }
                }
            }
            qf[qt][ks] = frag_from_f<T>(v);
        }
    }

    f32x4 o[QT][DT];
    float mrun[QT], lrun[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        mrun[qt] = -1.0e30f;
        lrun[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int ntiles = (nk + KT - 1) / KT;
    for (int tile = 0; tile < ntiles; ++tile) {
        __syncthreads();
        // ---- stage K (row-major, swizzled) and V (transposed) --------------------------------
        if constexpr (NWH > 1 && sizeof(T) == 2) {
            // head-packed small windows: the wave stages its own tile in exactly KT CPR / 64 rounds; all rounds'
            // global loads are requested before the first LDS store (clamped addresses; pad tokens and the
            // slots past nk / hd are patched afterwards), one memory round trip per tile instead of one per round
            constexpr int NIT = KT * CPR / 64;
            uint4 kvr[NIT], vvr[NIT];
            bool use[NIT], real[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + 64 * it;
                const int key = i / CPR, ch = i % CPR, d0 = ch * EPC;
                const int kk = tile * KT + key;
                const int ky = kk / G.wsw, kx = kk % G.wsw;
                const int y = wy * G.wsh + ky, x = wx * G.wsw + kx;
                use[it] = kk < nk && d0 < hd;
                real[it] = y < H && x < W;
                const T* src = qkv + (img_tok + (size_t)min(y, H - 1) * W + min(x, W - 1)) * p.ldqkv + dout + h * hd + min(d0, hd - EPC);
                kvr[it] = *reinterpret_cast<const uint4*>(src);
                vvr[it] = *reinterpret_cast<const uint4*>(src + dout);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + 64 * it;
                const int key = i / CPR, ch = i % CPR, d0 = ch * EPC;
                uint4 kv = kvr[it], vv = vvr[it];
                if (!use[it]) {
                    kv = make_uint4(0, 0, 0, 0);
                    vv = kv;
                } else if (!real[it]) {   // zero-pad token: k, v = bias
                    T kt_[EPC], vt2[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        kt_[e] = from_f<T>(bias[dout + h * hd + d0 + e]);
                        vt2[e] = from_f<T>(bias[2 * dout + h * hd + d0 + e]);
                    }
                    kv = *reinterpret_cast<const uint4*>(kt_);
                    vv = *reinterpret_cast<const uint4*>(vt2);
                }
                *reinterpret_cast<uint4*>(ks_ + key * RB + (swz(key, ch) << 4)) = kv;
                const T* ve = reinterpret_cast<const T*>(&vv);
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    *reinterpret_cast<T*>(vt_ + (d0 + e) * VRB + key * (int)sizeof(T)) = ve[e];
            }
        } else
        for (int i = tid; i < KT * CPR; i += nthr) {
            const int key = i / CPR, ch = i % CPR, d0 = ch * EPC;
            const int kk = tile * KT + key;
            uint4 kv = make_uint4(0, 0, 0, 0), vv = kv;
            if (kk < nk && d0 < hd) {
                const int ky = kk / G.wsw, kx = kk % G.wsw;
                const int y = wy * G.wsh + ky, x = wx * G.wsw + kx;
                if (y < H && x < W) {
                    const T* src = qkv + (img_tok + (size_t)y * W + x) * p.ldqkv + dout + h * hd + d0;
                    kv = *reinterpret_cast<const uint4*>(src);
                    vv = *reinterpret_cast<const uint4*>(src + dout);
                } else {  // zero-pad token: k, v = bias
                    T kt_[EPC], vt2[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        kt_[e] = from_f<T>(bias[dout + h * hd + d0 + e]);
                        vt2[e] = from_f<T>(bias[2 * dout + h * hd + d0 + e]);
                    }
                    kv = *reinterpret_cast<const uint4*>(kt_);
                    vv = *reinterpret_cast<const uint4*>(vt2);
                }
            }
            *reinterpret_cast<uint4*>(ks_ + key * RB + (swz(key, ch) << 4)) = kv;
            const T* ve = reinterpret_cast<const T*>(&vv);
#pragma unroll
            for (int e = 0; e < EPC; ++e)
                *reinterpret_cast<T*>(vt_ + (d0 + e) * VRB + key * (int)sizeof(T)) = ve[e];
        }
        __syncthreads();

        // ---- S^T = K . Q^T ---------------------------------------------------------------------
        frag kf[2][KSD];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < KSD; ++ks) {
                const int row = t * 16 + lr, c0 = (ks * 4 + g) * CPF;
                const uint4 x0 = *reinterpret_cast<const uint4*>(ks_ + row * RB + (swz(row, c0) << 4));
                uint4 x1 = x0;
                if constexpr (CPF == 2)
                    x1 = *reinterpret_cast<const uint4*>(ks_ + row * RB + (swz(row, c0 + 1) << 4));
                kf[t][ks] = make_frag(x0, x1, T{});
            }
        frag pf[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KSD; ++ks) mma16(kf[t][ks], qf[qt][ks], s[t]);
            }
            float mx = -1.0e30f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kidx = tile * KT + t * 16 + g * 4 + r;
                    s[t][r] = kidx < nk ? s[t][r] * c_log2 : -1.0e30f;
                    mx = fmaxf(mx, s[t][r]);
                }
            mx = group4_max(mx);
            const float mnew = fmaxf(mrun[qt], mx);
            const float alpha = exp2f(mrun[qt] - mnew);
            mrun[qt] = mnew;
            float pv[8], psum = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(s[t][r] - mnew);
                    pv[t * 4 + r] = e;
                    psum += e;
                }
            lrun[qt] = lrun[qt] * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= alpha;
            pf[qt] = frag_from_f<T>(pv);
        }
        // ---- O^T += V^T . P^T --------------------------------------------------------------------
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const char* vrow = vt_ + (dt * 16 + lr) * VRB;
            frag vf;
            if constexpr (sizeof(T) == 2) {
                const uint2 a = *reinterpret_cast<const uint2*>(vrow + 8 * g);
                const uint2 b = *reinterpret_cast<const uint2*>(vrow + 32 + 8 * g);
                vf = __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
            } else {
                const uint4 a = *reinterpret_cast<const uint4*>(vrow + 16 * g);
                const uint4 b = *reinterpret_cast<const uint4*>(vrow + 64 + 16 * g);
                vf = make_frag(a, b, T{});
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) mma16(vf, pf[qt], o[qt][dt]);
        }
    }

    // ---- normalise and store: lane holds O[q = l&15][d = 16dt + 4g .. +3] ---------------------
    T* __restrict__ out = static_cast<T*>(p.out);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = lrun[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (qtok[qt] < 0) continue;
        const float inv = 1.0f / l;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = dt * 16 + g * 4;
            if (d0 < hd) store4(out + (size_t)qtok[qt] * p.ldo + h * hd + d0, o[qt][dt] * inv);
        }
        if (p.lse && g == 0)
            p.lse[(size_t)qtok[qt] * nh + h] = (mrun[qt] + log2f(l)) * 0.69314718055994531f;
    }
}

// ---------------------------------------------------------------------------------------------
// v2 (bf16, windows with >= 128 keys and global attention).  Same arithmetic as above, different
// data movement:
//  * K and V tiles of 64 keys go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round
//    trip), double-buffered: the DMA of tile t+1 is issued right after the barrier that publishes
//    tile t and lands while tile t is being multiplied (one barrier per tile).
//  * both tiles are row-major [key][64 d] with the chunk-XOR swizzle applied on the per-lane
//    SOURCE address; V^T fragments come straight out of the row-major V tile with the transposing
//    LDS read ds_read_b64_tr_b16, so no transposed copy of V is ever written.
//  * pad tokens (k, v = bias) are fetched from a bf16 copy of the bias, masked keys / the d >= hd
//    tail from a 16-byte zero word, through the same per-lane source pointer.
// One workgroup = 4 waves x QT query tiles of 16 = one (window, head, block of 64*QT queries).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_vp_t;
typedef const __attribute__((address_space(1))) void* gbl_vp_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_p;
__device__ uint4 g_attn_zero16 = {0u, 0u, 0u, 0u};
__device__ uint4 g_attn_one16 = {0x00003F80u, 0u, 0u, 0u};   // bf16 1.0 followed by seven zeros

// ONES (head_dim <= 56): the first padding channel of every valid key's V row is fed as 1.0 (its 16-byte chunk
// is DMA'd from g_attn_one16 instead of the zero word), so channel hd of the output accumulates the softmax
// denominator in the PV MFMAs: no per-element adds on the VALU, which is what bounds this kernel.
#ifndef ATTN2_QT
#define ATTN2_QT 4
#endif
// KSQ = 32-wide k-steps of Q K^T, DT = 16-wide output tiles: (2, 4) for head_dim <= 64 (128-byte tile rows);
// (3, 5) / (3, 6) for head_dim <= 80 / 96 (Hiera-L: 72, Hiera-T/S: 96) on 256-byte tile rows of 16 chunks.
template <int QT, bool ONES, int KSQ = 2, int DT = 4>
__global__ __launch_bounds__(256, (QT == 2 && KSQ == 2) ? 3 : 2) void attn2_fwd_kernel(const AttnArgs p, const float c_log2, const int nqb) {
    using T = bf16;
    static_assert((KSQ == 2 && DT == 4) || (KSQ == 3 && (DT == 5 || DT == 6)), "supported head_dim classes");
    constexpr int KT2 = 64, RB = KSQ == 2 ? 128 : 256, TILE_B = KT2 * RB, STAGE = 2 * TILE_B;
    constexpr int CPR = RB / 16, SWM = CPR - 1;      // chunks per row; the swizzle XORs the chunk index with row & SWM
    constexpr int RPI = 1024 / RB, NE = KT2 / RPI / 4;   // tile rows per DMA wave-instruction; instructions per wave and operand
    auto swzk = [](int row, int chunk) { return chunk ^ (row & SWM); };
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lr = lane & 15;
    const int H = p.H, W = p.W, nh = p.nh, hd = p.hd, dout = nh * hd;
    const WinGeom G = make_geom(H, W, p.ws, p.pool);
    // 1-D grid, XCD-aware (workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MiB L2): XCD x takes the
    // windows w = x (mod 8); within it the query blocks of one (window, head) are consecutive and the heads of a window
    // follow each other.  The workgroups an XCD runs at one time then stream the K / V of two or three (window, head)
    // pairs (0.9 MB each for a 4096-key global block) out of its L2, instead of all heads of two frames (15 MB): the old
    // (window, head, query block) grid order fetched 965 MB per launch for 176 MB of q / k / v.  The heads of a window
    // share the 128-byte lines their 112-byte slices straddle, so they stay on one XCD.
    const int nwin_all = p.N * G.nwy * G.nwx;
    const int j_ = blockIdx.x >> 3, z_ = j_ % nqb, h = (j_ / nqb) % nh, gwin = (j_ / (nqb * nh)) * 8 + (blockIdx.x & 7);
    if (gwin >= nwin_all) return;   // uniform for the workgroup, before any barrier
    const int win = gwin % (G.nwy * G.nwx), n = gwin / (G.nwy * G.nwx);
    const int wy = win / G.nwx, wx = win % G.nwx;
    const int rh = min(G.qh, G.Ho - wy * G.qh), rw = min(G.qw, G.Wo - wx * G.qw);
    const int nq = rh * rw, nk = G.wsh * G.wsw;
    const int qbase = z_ * (64 * QT);
    if (qbase >= nq) return;  // uniform for the workgroup, before any barrier

    const T* __restrict__ qkv = static_cast<const T*>(p.qkv);
    const T* __restrict__ bias_t = static_cast<const T*>(p.bias_t);
    const size_t img_tok = (size_t)n * H * W;

    // ---- Q fragments ----------------------------------------------------------------------------
    bf16x8 qf[QT][KSQ];
    int qtok[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qi = qbase + (wave * QT + qt) * 16 + lr;
        const bool qv = qi < nq;
        const int iy = qv ? qi / rw : 0, ix = qv ? qi % rw : 0;
        qtok[qt] = qv ? (int)(((size_t)n * G.Ho + (wy * G.qh + iy)) * G.Wo + (wx * G.qw + ix)) : -1;
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) {
            const int d0 = ks * 32 + g * 8;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if (qv && d0 < hd) {
                if (p.qpool) {   // pooled in the qkv GEMM's epilogue: max is monotonic under the bf16 rounding, same bits
                    const float* qs = p.qpool + (size_t)qtok[qt] * p.ldqp + h * hd + d0;
                    const f32x4 q0 = load4(qs), q1 = load4(qs + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = q0[j]; v[4 + j] = q1[j]; }
                } else if (p.pool == 1) {
                    const size_t tok = img_tok + (size_t)(wy * G.wsh + iy) * W + (wx * G.wsw + ix);
                    load8f<T>(qkv + tok * p.ldqkv + h * hd + d0, v);
                } else if (p.pool == 2) {
                    load_q_pool2<T>(qkv, img_tok, p.ldqkv, H, W, wy * G.wsh + iy * 2, wx * G.wsw + ix * 2, h * hd + d0,
                                    [&](float* b) { load8f<T>(bias_t + h * hd + d0, b); }, v);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = -3.0e38f;
                    for (int a = 0; a < p.pool; ++a)
                        for (int b = 0; b < p.pool; ++b) {
                            const int y = wy * G.wsh + iy * p.pool + a, x = wx * G.wsw + ix * p.pool + b;
                            float t[8];
                            if (y < H && x < W) load8f<T>(qkv + (img_tok + (size_t)y * W + x) * p.ldqkv + h * hd + d0, t);
                            else load8f<T>(bias_t + h * hd + d0, t);
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], t[j]);
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= c_log2;   // scores come out of the MFMAs in log2 units
            qf[qt][ks] = frag_from_f<T>(v);
        }
    }

    // ---- DMA lane mapping: instruction j = wave + 4 i covers rows 8(j&7).. of K (j<8) or V (j>=8);
    // a lane therefore touches two keys per tile (tile rows r0 and r0 + 32), whose window
    // coordinates are advanced incrementally from tile to tile (no per-tile division).
    // (tile row of instruction j = wave + 4 e: RPI j + lrow; its low swizzle bits do not depend on e)
    const int lrow = lane / CPR, dsrc = ((lane % CPR) ^ ((RPI * wave + lrow) & SWM)) * 8;   // source d offset of this lane's chunk
    const bool dok = dsrc < hd;
    const int kstep_y = KT2 / G.wsw, kstep_x = KT2 % G.wsw;
    int kky[NE], kkx[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int kk = RPI * wave + lrow + 4 * RPI * e;
        kky[e] = kk / G.wsw;
        kkx[e] = kk - kky[e] * G.wsw;
    }
    auto issue = [&](int buf) {   // stages the tile the (kky, kkx) cursors point at, then advances them
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int y = wy * G.wsh + kky[e], x = wx * G.wsw + kkx[e];
            const bool kin = kky[e] < G.wsh && dok;            // key index < nk
            const bool real = y < H && x < W;
            const size_t tokoff = (img_tok + (size_t)y * W + x) * p.ldqkv;
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int part = (v ? 2 * dout : dout) + h * hd + dsrc;
                const void* src = &g_attn_zero16;
                if (kin) src = real ? static_cast<const void*>(qkv + tokoff + part) : static_cast<const void*>(bias_t + part);
                if (ONES && v == 1 && dsrc == hd && kky[e] < G.wsh) src = &g_attn_one16;
                __builtin_amdgcn_global_load_lds((gbl_vp_t)src, (lds_vp_t)(smem + buf * STAGE + v * TILE_B + (wave + 4 * e) * 1024), 16, 0, 0);
            }
            kky[e] += kstep_y;
            kkx[e] += kstep_x;
            if (kkx[e] >= G.wsw) { kkx[e] -= G.wsw; kky[e] += 1; }
        }
    };

    // Running row maximum, kept NEGATED and replicated as the C operand of the score MFMAs: the scores come out of
    // the matrix pipe as s - m (log2 units, the scale is folded into Q), so the common case -- no row's maximum
    // moved -- is one v_exp_f32 per score with no subtract in front of it.  The first tile runs against m = 0 and
    // always takes the re-centring path.
    f32x4 o[QT][DT], negm[QT];
    float lrun[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        negm[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        lrun[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int ntiles = (nk + KT2 - 1) / KT2;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(smem);
    issue(0);
#pragma nounroll
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        lds_dma_wait_barrier();   // tile landed (vmcnt(0) + barrier); everyone is done with the other buffer
        if (tile + 1 < ntiles) issue(buf ^ 1);
        const char* kb = smem + buf * STAGE;
        const unsigned vb_lds = lds0 + buf * STAGE + TILE_B;
        const bool tail = __builtin_amdgcn_readfirstlane(tile * KT2 + KT2 > nk ? 1 : 0) != 0;

        bf16x8 kf[4][KSQ];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < KSQ; ++ks) {
                const int row = t * 16 + lr;
                kf[t][ks] = *reinterpret_cast<const bf16x8*>(kb + row * RB + (swzk(row, ks * 4 + g) << 4));
            }
        bf16x8 pf[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4 s[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][0], qf[qt][0], negm[qt], 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < KSQ; ++ks) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][ks], qf[qt][ks], s[t], 0, 0, 0);
            }
            // online softmax on s - m; O (and l) are rescaled only when some row's maximum actually moved
            // (exact, not a threshold)
            if (tail) {   // block-uniform: only the last tile of a window pays for the key mask
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (tile * KT2 + t * 16 + g * 4 + r >= nk) s[t][r] = -1.0e30f;
            }
            float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
            for (int t = 1; t < 4; ++t) mx = fmaxf(fmaxf(mx, s[t][0]), fmaxf(fmaxf(s[t][1], s[t][2]), s[t][3]));
            mx = group4_max(mx);
            if (tile == 0 || __any(mx > 0.f)) {
                const float d = tile == 0 ? mx : fmaxf(mx, 0.f);   // how far this row's maximum moved
                if (tile != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-d);
                    if constexpr (!ONES) lrun[qt] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= alpha;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) negm[qt][r] -= d;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[t][r] -= d;
            }
            float psum = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float pv[8];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(s[2 * u + t][r]);
                        pv[t * 4 + r] = e;
                        if constexpr (!ONES) psum += e;
                    }
                pf[qt][u] = frag_from_f<T>(pv);
            }
            if constexpr (!ONES) lrun[qt] += psum;
        }
        // ---- O^T += V^T . P^T, V^T fragments by transposing LDS reads --------------------------
        // The reads are inline asm: hipcc cannot tell the builtin transposing read apart from the LDS
        // write of the K/V DMA issued above and puts s_waitcnt vmcnt(0) in front of it, which makes
        // every tile wait for the NEXT tile's DMA half-way through.  Hand-ordered instead: the reads
        // of key half u + 1 are issued before the MFMAs of half u.
        uint2 v0[DT][2], v1[DT][2];
#define LSAM_VREAD(dst, u)                                                                              \
    _Pragma("unroll") for (int dt = 0; dt < DT; ++dt) {                                                 \
        const int row = 32 * (u) + 4 * g + (lr >> 2), ch = 2 * dt + ((lr & 3) >> 1);                    \
        const unsigned a0 = vb_lds + row * RB + (swzk(row, ch) << 4) + (lr & 1) * 8;                    \
        const int row1 = row + 16;                                                                      \
        const unsigned a1 = vb_lds + row1 * RB + (swzk(row1, ch) << 4) + (lr & 1) * 8;                  \
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst[dt][0]) : "v"(a0));                         \
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst[dt][1]) : "v"(a1));                         \
    }
#define LSAM_VMMA(src, u)                                                                               \
    _Pragma("unroll") for (int dt = 0; dt < DT; ++dt) {                                                 \
        const bf16x8 vf = __builtin_bit_cast(bf16x8, make_uint4(src[dt][0].x, src[dt][0].y, src[dt][1].x, src[dt][1].y)); \
        _Pragma("unroll") for (int qt = 0; qt < QT; ++qt) mma16(vf, pf[qt][u], o[qt][dt]);              \
    }
        LSAM_VREAD(v0, 0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        LSAM_VREAD(v1, 1)
        __builtin_amdgcn_sched_barrier(0);
        LSAM_VMMA(v0, 0)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        LSAM_VMMA(v1, 1)
        __builtin_amdgcn_sched_barrier(0);
#undef LSAM_VREAD
#undef LSAM_VMMA
    }

    T* __restrict__ out = static_cast<T*>(p.out);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l;
        if constexpr (ONES) {
            // channel hd = tile hd / 16, lane group (hd % 16) / 4, element hd % 4 (hd % 8 == 0): owned by one lane per query
            const int dt1 = hd >> 4, g1 = (hd & 15) >> 2;
            float lv = 0.f;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                if (dt == dt1) lv = o[qt][dt][0];
            l = __shfl(lv, 16 * g1 + lr, 64);
        } else {
            l = group4_sum(lrun[qt]);
        }
        if (qtok[qt] < 0) continue;
        const float inv = 1.0f / l;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = dt * 16 + g * 4;
            if (d0 < hd) store4(out + (size_t)qtok[qt] * p.ldo + h * hd + d0, o[qt][dt] * inv);
        }
        if (p.lse && g == 0)
            p.lse[(size_t)qtok[qt] * nh + h] = (log2f(l) - negm[qt][0]) * 0.69314718055994531f;
    }
}


// ---------------------------------------------------------------------------------------------
// v3 (bf16, windows with 65..208 keys: the 14x14 windows of Hiera stage 3 and the pooled 14x14 -> 7x7
// transition).  The whole window's K and V ([224 rows][64 d] each, 56 KiB together) are staged ONCE by
// LDS-DMA, so the softmax is a plain single pass (no running maximum, no rescaling of O) and there is
// one barrier per workgroup.  Work is cut to what the window needs: 13 key tiles of 16 (not 4 x 64)
// and ceil(nq / 16) query tiles (13 for a full window, not 16): 54 MFMAs per query tile.
// One workgroup = one (window, head); its 4 waves take query tiles round-robin; two workgroups per CU
// overlap one's staging with the other's MFMAs.
// ---------------------------------------------------------------------------------------------
template <int NKT>
#ifndef ATTNW_WPS
#define ATTNW_WPS 3
#endif
__global__ __launch_bounds__(256, ATTNW_WPS) void attnw_fwd_kernel(const AttnArgs p, const float c_log2) {
    using T = bf16;
    // 16 NKT rows per operand, V first: the PV MFMAs take keys 32 at a time, and for an odd NKT the upper half of
    // the last pair (rows 16 NKT .. 16 NKT + 15 of V) is then the first 16 rows of K -- finite values against
    // probabilities that are exactly zero -- instead of 4 KiB of padding.  52 KiB instead of 56 per workgroup is
    // what lets THREE workgroups share a CU's 160 KiB (with waves of <= 168 registers): the kernel is bound by
    // the latency of its one-shot K/V staging, so residency is what it needs.
    constexpr int U = (NKT + 1) / 2, NKR = 16 * NKT, RB = 128, TILE_B = NKR * RB, DT = 4;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* vb = smem;
    char* kb = smem + TILE_B;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lr = lane & 15;
    const int H = p.H, W = p.W, nh = p.nh, hd = p.hd, dout = nh * hd;
    const WinGeom G = make_geom(H, W, p.ws, p.pool);
    // workgroup id -> (window, head) so that the heads of one window run back to back on ONE XCD
    // (workgroups are dealt round-robin to the 8 XCDs): a head's 112-byte k/v slices share their
    // 128-byte lines with the neighbouring heads', which then hit in that XCD's L2.
    const int k_ = blockIdx.x >> 3;
    const int h = k_ % nh, gwin = (k_ / nh) * 8 + (blockIdx.x & 7);
    if (gwin >= p.N * G.nwy * G.nwx) return;   // uniform for the workgroup, before any barrier
    const int win = gwin % (G.nwy * G.nwx), n = gwin / (G.nwy * G.nwx);
    const int wy = win / G.nwx, wx = win % G.nwx;
    const int rh = min(G.qh, G.Ho - wy * G.qh), rw = min(G.qw, G.Wo - wx * G.qw);
    const int nq = rh * rw, nk = G.wsh * G.wsw;
    const T* __restrict__ qkv = static_cast<const T*>(p.qkv);
    const T* __restrict__ bias_t = static_cast<const T*>(p.bias_t);
    const size_t img_tok = (size_t)n * H * W;

    const bool ones = hd % 8 == 0 && hd <= 56;   // kernel-uniform
    // ---- stage K and V: DMA instruction j covers tile rows 8j .. 8j+7 of V (j < 2 NKT) or K -------
    {
        const int lrow = lane >> 3, dsrc = ((lane & 7) ^ lrow) * 8;   // swizzle applied on the source side
        const bool dok = dsrc < hd;
        constexpr int NI = NKR / 8;
#pragma unroll 1
        for (int j = wave; j < 2 * NI; j += 4) {
            const int v = j < NI ? 1 : 0, kk = 8 * (j - (1 - v) * NI) + lrow;
            const int ky = kk / G.wsw, kx = kk - ky * G.wsw;
            const int y = wy * G.wsh + ky, x = wx * G.wsw + kx;
            const void* src = &g_attn_zero16;
            if (kk < nk && dok) {
                const int part = (v ? 2 * dout : dout) + h * hd + dsrc;
                src = (y < H && x < W) ? static_cast<const void*>(qkv + (img_tok + (size_t)y * W + x) * p.ldqkv + part)
                                       : static_cast<const void*>(bias_t + part);
            }
            // softmax denominator through the PV MFMAs: V's first padding channel of every valid key is 1.0
            if (ones && v == 1 && kk < nk && dsrc == hd) src = &g_attn_one16;
            __builtin_amdgcn_global_load_lds((gbl_vp_t)src, (lds_vp_t)(smem + j * 1024), 16, 0, 0);
        }
    }
    // ---- Q fragments of ALL of this wave's query tiles, requested while the K/V DMA is in flight --------
    constexpr int MAXQT = (NKT + 3) / 4;   // query tiles per wave (nq <= nk <= 16 NKT)
    bf16x8 qf_all[MAXQT][2];
    int qtok_all[MAXQT];
#pragma unroll
    for (int i = 0; i < MAXQT; ++i) {
        const int qi = (wave + 4 * i) * 16 + lr;
        const bool qv = qi < nq;
        const int iy = qv ? qi / rw : 0, ix = qv ? qi % rw : 0;
        qtok_all[i] = qv ? (int)(((size_t)n * G.Ho + (wy * G.qh + iy)) * G.Wo + (wx * G.qw + ix)) : -1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int d0 = ks * 32 + g * 8;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if (qv && d0 < hd) {
                if (p.qpool) {   // pooled in the qkv GEMM's epilogue: max is monotonic under the bf16 rounding, same bits
                    const float* qs = p.qpool + (size_t)qtok_all[i] * p.ldqp + h * hd + d0;
                    const f32x4 q0 = load4(qs), q1 = load4(qs + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = q0[j]; v[4 + j] = q1[j]; }
                } else if (p.pool == 1) {
                    const size_t tok = img_tok + (size_t)(wy * G.wsh + iy) * W + (wx * G.wsw + ix);
                    load8f<T>(qkv + tok * p.ldqkv + h * hd + d0, v);
                } else if (p.pool == 2) {
                    load_q_pool2<T>(qkv, img_tok, p.ldqkv, H, W, wy * G.wsh + iy * 2, wx * G.wsw + ix * 2, h * hd + d0,
                                    [&](float* b) { load8f<T>(bias_t + h * hd + d0, b); }, v);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = -3.0e38f;
                    for (int a = 0; a < p.pool; ++a)
                        for (int b = 0; b < p.pool; ++b) {
                            const int y = wy * G.wsh + iy * p.pool + a, x = wx * G.wsw + ix * p.pool + b;
                            float t[8];
                            if (y < H && x < W) load8f<T>(qkv + (img_tok + (size_t)y * W + x) * p.ldqkv + h * hd + d0, t);
                            else load8f<T>(bias_t + h * hd + d0, t);
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], t[j]);
                        }
                }
            }
            qf_all[i][ks] = frag_from_f<T>(v);
        }
    }
    lds_dma_wait_barrier();

    T* __restrict__ out = static_cast<T*>(p.out);
    const int nqt = (nq + 15) >> 4;
#pragma unroll
    for (int i = 0; i < MAXQT; ++i) {
        if (wave + 4 * i >= nqt) break;   // wave-uniform
        const int qtok = qtok_all[i];
        const bf16x8 (&qf)[2] = qf_all[i];
        // ---- S^T = K . Q^T for all key tiles; lane = one query, 4 keys per tile --------------------
        f32x4 s[NKT];
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            const int row = t * 16 + lr;
            const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kb + row * RB + (swz(row, g) << 4));
            const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kb + row * RB + (swz(row, 4 + g) << 4));
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[1], s[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 4; t < NKT; ++t)   // keys past the window (alignment slots) never count; nk > 64
            if (t * 16 + 16 > nk) {     // workgroup-uniform: only the window's last key tile(s) pay for the mask
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (t * 16 + g * 4 + r >= nk) s[t][r] = -1.0e30f;
            }
        float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
        for (int t = 1; t < NKT; ++t) mx = fmaxf(fmaxf(mx, s[t][0]), fmaxf(fmaxf(s[t][1], s[t][2]), s[t][3]));
        mx = group4_max(mx);
        const float mneg = -mx * c_log2;
        float l = 0.f;
        bf16x8 pf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float pv[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float e = 0.f;
                    if (2 * u + t < NKT) {
                        e = __builtin_amdgcn_exp2f(fmaf(s[2 * u + t < NKT ? 2 * u + t : 0][r], c_log2, mneg));
                        if (!ones) l += e;
                    }
                    pv[t * 4 + r] = e;
                }
            pf[u] = frag_from_f<T>(pv);
        }
        // ---- O^T = V^T . P^T, V^T fragments by transposing LDS reads -------------------------------
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int row = 32 * u + 4 * g + (lr >> 2), ch = 2 * dt + ((lr & 3) >> 1);
                const char* a0 = vb + row * RB + (swz(row, ch) << 4) + (lr & 1) * 8;
                const int row1 = row + 16;
                const char* a1 = vb + row1 * RB + (swz(row1, ch) << 4) + (lr & 1) * 8;
                const s16x4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(a0));
                const s16x4_t x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(a1));
                const uint2 u0 = __builtin_bit_cast(uint2, x0), u1 = __builtin_bit_cast(uint2, x1);
                mma16(__builtin_bit_cast(bf16x8, make_uint4(u0.x, u0.y, u1.x, u1.y)), pf[u], o[dt]);
            }
        if (ones) {   // channel hd of O^T: tile hd / 16, lane group (hd % 16) / 4, element 0
            float lv = 0.f;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                if (dt == (hd >> 4)) lv = o[dt][0];
            l = __shfl(lv, 16 * ((hd & 15) >> 2) + lr, 64);
        } else {
            l = group4_sum(l);
        }
        if (qtok >= 0) {
            const float inv = 1.0f / l;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d0 = dt * 16 + g * 4;
                if (d0 < hd) store4(out + (size_t)qtok * p.ldo + h * hd + d0, o[dt] * inv);
            }
            if (p.lse && g == 0) p.lse[(size_t)qtok * nh + h] = (mx * c_log2 + log2f(l)) * 0.69314718055994531f;
        }
    }
}

static int attnw_launch(const AttnArgs& a, hipStream_t s) {
    constexpr int NKT = 13, LDSB = 2 * 16 * NKT * 128;
    const WinGeom G = make_geom(a.H, a.W, a.ws, a.pool);
    const float c = (1.0f / sqrtf((float)a.hd)) * 1.4426950408889634f;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attnw_fwd_kernel<NKT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        attr_set = true;
    }
    hipLaunchKernelGGL((attnw_fwd_kernel<NKT>), dim3(8 * cdiv(a.N * G.nwy * G.nwx, 8) * a.nh), dim3(256), LDSB, s, a, c);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// ---------------------------------------------------------------------------------------------
// 4 x 4 windows (16 keys; Hiera stage 2 and its exit block: hieradet.py:56-81 with window_spec[1] = 4): one WAVE per
// (window, head), no workgroup barrier, nothing staged but V.
//   S^T = K Q^T : K rows and Q rows are MFMA operands as they lie in memory (lane (g, lr): token lr, channels
//                 32 ks + 8 g .. + 7: one 16-byte global load per k-step and operand), so neither goes through LDS;
//   softmax over the 16 keys of a query = 4 accumulator registers x the 4 lane groups (v_permlane swaps);
//   O^T = V^T P^T: the accumulator lane (keys 4 g .. 4 g + 3 of query lr) IS the B operand if k-slot 8 g + j means key
//                 4 g + j (j < 4, the other slots zero); V^T comes out of a wave-private row-major V tile in LDS (3.3 KB)
//                 through the transposing read ds_read_b64_tr_b16, one read per 16 output channels.
// The wave walks a strided list of (window, head) items (consecutive waves take the heads of one window: their
// requests share cache lines) and asks for the next item's rows before it computes the current one.
// Bound: HBM (qkv read once, out written once); 64-byte segments per token row and instruction: 3.5 - 3.7 TB/s measured.
// POOLQ: the (2 x 2 max-pooled, fp32) queries come from `qpool` (4 per window).
// ---------------------------------------------------------------------------------------------
template <int KSD, int DT, bool POOLQ>
__global__ __launch_bounds__(256) void attn4_fwd_kernel(const AttnArgs p, const float c_log2, const int nitems) {
    constexpr int RBV = KSD * 64 + 16;                 // V tile row stride: conflict-free for the transposing read
    __shared__ __attribute__((aligned(16))) char vt_all[4 * 16 * RBV];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, lr = lane & 15;
    char* vt = vt_all + wave * 16 * RBV;
    const int H = p.H, W = p.W, nh = p.nh, hd = p.hd, dout = nh * hd;
    const int nwx = W >> 2, nwin = (H >> 2) * nwx;
    const int Ho = POOLQ ? H >> 1 : H, Wo = POOLQ ? W >> 1 : W, qw = POOLQ ? 2 : 4, nq = qw * qw;
    const bf16* __restrict__ qkv = static_cast<const bf16*>(p.qkv);
    bf16* __restrict__ out = static_cast<bf16*>(p.out);
    const int nwaves = gridDim.x * 4;

    struct Item { uint4 k[KSD], v[KSD], q[KSD]; f32x4 qf[POOLQ ? 2 * KSD : 1]; int qtok, h; };
    auto load_item = [&](int item, Item& it) {
        const int wg = item / nh, h = item - wg * nh;
        const int n = wg / nwin, win = wg - n * nwin, wy = win / nwx, wx = win - wy * nwx;
        const size_t ktok = ((size_t)n * H + (wy * 4 + (lr >> 2))) * W + (wx * 4 + (lr & 3));
        const bf16* kr = qkv + ktok * p.ldqkv + dout + h * hd;
        const int qi = lr < nq ? lr : 0;
        const size_t qtok = ((size_t)n * Ho + (wy * qw + qi / qw)) * Wo + (wx * qw + qi % qw);
        it.qtok = lr < nq ? (int)qtok : -1;
        it.h = h;
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) {
            const int d0 = 32 * ks + 8 * g;
            const bool ok = d0 < hd;
            const int dc = ok ? d0 : hd - 8;            // clamped: never read past the head's channels
            const uint4 kk = *reinterpret_cast<const uint4*>(kr + dc);
            it.k[ks] = ok ? kk : make_uint4(0, 0, 0, 0);
            it.v[ks] = *reinterpret_cast<const uint4*>(kr + dout + dc);   // channels >= hd: rows of O^T that are never stored
            if constexpr (POOLQ) {
                const float* qs = p.qpool + qtok * p.ldqp + h * hd + dc;
                it.qf[2 * ks] = load4(qs);
                it.qf[2 * ks + 1] = load4(qs + 4);
            } else {
                const uint4 qq = *reinterpret_cast<const uint4*>(qkv + qtok * p.ldqkv + h * hd + dc);
                it.q[ks] = ok ? qq : make_uint4(0, 0, 0, 0);
            }
        }
    };

    int item = blockIdx.x * 4 + wave;
    if (item >= nitems) return;                         // wave-uniform; the kernel has no workgroup barrier
    Item cur, nxt;
    load_item(item, cur);
#pragma unroll 1
    for (; item < nitems; item += nwaves) {
        const bool more = item + nwaves < nitems;       // wave-uniform
        if (more) load_item(item + nwaves, nxt);
        // ---- V tile -> LDS (row-major [key][channel]) --------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) *reinterpret_cast<uint4*>(vt + lr * RBV + ks * 64 + g * 16) = cur.v[ks];
        // ---- S^T = K Q^T -----------------------------------------------------------------------------------
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) {
            bf16x8 qf;
            if constexpr (POOLQ) {
                const bool ok = 32 * ks + 8 * g < hd;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    qf[j] = (bf16)(ok ? cur.qf[2 * ks][j] : 0.f);
                    qf[4 + j] = (bf16)(ok ? cur.qf[2 * ks + 1][j] : 0.f);
                }
            } else {
                qf = __builtin_bit_cast(bf16x8, cur.q[ks]);
            }
            mma16(__builtin_bit_cast(bf16x8, cur.k[ks]), qf, sc);
        }
        // ---- softmax over the 16 keys of query lr: registers r of the four lane groups -----------------------
        float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
        mx = group4_max(mx);
        float e[4], l = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            e[r] = exp2f((sc[r] - mx) * c_log2);
            l += e[r];
        }
        l = group4_sum(l);
        bf16x8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pf[r] = (bf16)e[r];
            pf[4 + r] = (bf16)0.f;
        }
        // ---- O^T = V^T P^T ---------------------------------------------------------------------------------
        const float inv = 1.0f / l;
        const bool qok = cur.qtok >= 0;
        bf16* orow = out + (size_t)(qok ? cur.qtok : 0) * p.ldo + cur.h * hd;
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const char* a0 = vt + (4 * g + (lr >> 2)) * RBV + 32 * dt + 8 * (lr & 3);
            const s16x4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(a0));
            const uint2 u0 = __builtin_bit_cast(uint2, x0);
            const bf16x8 vf = __builtin_bit_cast(bf16x8, make_uint4(u0.x, u0.y, 0u, 0u));
            o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma16(vf, pf, o[dt]);
            o[dt] *= inv;
        }
        // lane: channels 16 dt + 4 g + r of query lr; pairs of column tiles regrouped into 16-byte stores (hd % 8 == 0)
#pragma unroll
        for (int dt = 0; dt + 1 < DT; dt += 2)
            store_bf16_tile_pair(orow + 16 * dt, g, o[dt], o[dt + 1], qok && 16 * dt + 8 * g < hd);
        if constexpr (DT % 2 == 1) {
            if (qok && 16 * (DT - 1) + 4 * g < hd) store4(orow + 16 * (DT - 1) + 4 * g, o[DT - 1]);
        }
        if (p.lse && g == 0 && qok)
            p.lse[(size_t)cur.qtok * nh + cur.h] = (mx * c_log2 + log2f(l)) * 0.69314718055994531f;
        if (more) cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------
// 8 x 8 windows, head_dim 72 .. 96 (Hiera-L / S / T stage 1, whose widths the fused stage-1 kernel does not take): the same
// scheme on 64 keys and 64 queries -- one wave per (window, head), K and Q fragments straight from memory (4 token tiles
// x KSD k-steps each), V through a wave-private row-major LDS tile (64 x 208 B) and the transposing read.  All 16 score
// tiles are formed first (K and Q die), then per query tile: softmax over its 64 keys (16 registers x 4 lane groups),
// O^T = V^T P^T with k-slot 8 g + 4 t + r meaning key 32 u + 16 t + 4 g + r, store.  No prefetch across items: two waves
// per SIMD in different phases keep the loads flowing.  HBM-bound like the 4 x 4 kernel.
// ---------------------------------------------------------------------------------------------
#ifndef ATTN8_WGS
#define ATTN8_WGS 2   // workgroups per CU (52 KiB of LDS, 170 registers each; 3 is slower: 333 against 285 us)
#endif
template <int KSD, int DT>
__global__ __launch_bounds__(256, ATTN8_WGS) void attn8_fwd_kernel(const AttnArgs p, const float c_log2, const int nitems) {
    constexpr int RBV = KSD * 64 + 16;
    __shared__ __attribute__((aligned(16))) char vt_all[4 * 64 * RBV];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, lr = lane & 15;
    char* vt = vt_all + wave * 64 * RBV;
    const int H = p.H, W = p.W, nh = p.nh, hd = p.hd, dout = nh * hd;
    const int nwx = W >> 3, nwin = (H >> 3) * nwx;
    const bf16* __restrict__ qkv = static_cast<const bf16*>(p.qkv);
    bf16* __restrict__ out = static_cast<bf16*>(p.out);
    const int nwaves = gridDim.x * 4;
#pragma unroll 1
    for (int item = blockIdx.x * 4 + wave; item < nitems; item += nwaves) {
        const int wg = item / nh, h = item - wg * nh;
        const int n = wg / nwin, win = wg - n * nwin, wy = win / nwx, wx = win - wy * nwx;
        size_t tok[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {   // token 16 tt + lr of the window: row 2 tt + (lr >> 3), column lr & 7
            tok[tt] = ((size_t)n * H + (wy * 8 + 2 * tt + (lr >> 3))) * W + (wx * 8 + (lr & 7));
        }
        uint4 kf[4][KSD], qf[4][KSD];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int ks = 0; ks < KSD; ++ks) {
                const int d0 = 32 * ks + 8 * g;
                const bool ok = d0 < hd;
                const bf16* row = qkv + tok[tt] * p.ldqkv + h * hd + (ok ? d0 : hd - 8);   // clamped: never past the head's channels
                const uint4 vv = *reinterpret_cast<const uint4*>(row + 2 * dout);
                const uint4 kk = *reinterpret_cast<const uint4*>(row + dout);
                const uint4 qq = *reinterpret_cast<const uint4*>(row);
                *reinterpret_cast<uint4*>(vt + (16 * tt + lr) * RBV + ks * 64 + g * 16) = vv;   // channels >= hd: rows of O^T never stored
                kf[tt][ks] = ok ? kk : make_uint4(0, 0, 0, 0);
                qf[tt][ks] = ok ? qq : make_uint4(0, 0, 0, 0);
            }
        // ---- S^T[key tile][query tile] -----------------------------------------------------------------------
        f32x4 sc[4][4];   // [qt][kt]
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                sc[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KSD; ++ks)
                    mma16(__builtin_bit_cast(bf16x8, kf[kt][ks]), __builtin_bit_cast(bf16x8, qf[qt][ks]), sc[qt][kt]);
            }
        // ---- V^T fragments (shared by the four query tiles) ----------------------------------------------------
        bf16x8 vf[DT][2];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const char* a0 = vt + (32 * u + 4 * g + (lr >> 2)) * RBV + 32 * dt + 8 * (lr & 3);
                const s16x4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(a0));
                const s16x4_t x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(a0 + 16 * RBV));
                const uint2 u0 = __builtin_bit_cast(uint2, x0), u1 = __builtin_bit_cast(uint2, x1);
                vf[dt][u] = __builtin_bit_cast(bf16x8, make_uint4(u0.x, u0.y, u1.x, u1.y));
            }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            float mx = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[qt][kt][r]);
            mx = group4_max(mx);
            float l = 0.f;
            bf16x8 pf[2];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f((sc[qt][kt][r] - mx) * c_log2);
                    l += e;
                    pf[kt >> 1][4 * (kt & 1) + r] = (bf16)e;
                }
            l = group4_sum(l);
            const float inv = 1.0f / l;
            f32x4 o[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                mma16(vf[dt][0], pf[0], o[dt]);
                mma16(vf[dt][1], pf[1], o[dt]);
                o[dt] *= inv;
            }
            bf16* orow = out + tok[qt] * p.ldo + h * hd;
#pragma unroll
            for (int dt = 0; dt + 1 < DT; dt += 2) store_bf16_tile_pair(orow + 16 * dt, g, o[dt], o[dt + 1], 16 * dt + 8 * g < hd);
            if constexpr (DT % 2 == 1) {
                if (16 * (DT - 1) + 4 * g < hd) store4(orow + 16 * (DT - 1) + 4 * g, o[DT - 1]);
            }
            if (p.lse && g == 0) p.lse[tok[qt] * nh + h] = (mx * c_log2 + log2f(l)) * 0.69314718055994531f;
        }
    }
}
static bool attn8_supported(const AttnArgs& a) {
    return a.ws == 8 && a.pool == 1 && a.H % 8 == 0 && a.W % 8 == 0 && a.hd % 8 == 0 && a.hd > 64 && a.hd <= 96 && a.ldqkv % 8 == 0 &&
           a.ldo % 8 == 0;   // (the scores are formed from the unscaled bf16 q and k, as the backward recomputes them: the LSE is consistent)
}
static int attn8_launch(const AttnArgs& a, hipStream_t s) {
    const float c = (1.0f / sqrtf((float)a.hd)) * 1.4426950408889634f;
    const long nitems_l = (long)a.N * (a.H / 8) * (a.W / 8) * a.nh;
    LSAM_REQUIRE(nitems_l < (1L << 31), LSAM_ERR_UNSUPPORTED, "attention: too many windows");
    const int nitems = (int)nitems_l;
    const dim3 grid(min(cdiv(nitems, 4), 256 * ATTN8_WGS)), blk(256);
    if (a.hd <= 80) hipLaunchKernelGGL((attn8_fwd_kernel<3, 5>), grid, blk, 0, s, a, c, nitems);
    else hipLaunchKernelGGL((attn8_fwd_kernel<3, 6>), grid, blk, 0, s, a, c, nitems);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// (head_dim <= 64 with full windows of queries stays on the four-heads-per-workgroup kernel: 118 against 129 us on the B+
// stage-2 shape; head_dim 72 .. 96: 692 against 1156 us on the Hiera-L shape; pooled queries: 172 against 189 us)
static bool attn4_supported(const AttnArgs& a) {
    return a.ws == 4 && a.H % 4 == 0 && a.W % 4 == 0 && a.hd % 8 == 0 && a.hd <= 96 && a.ldqkv % 8 == 0 && a.ldo % 8 == 0 &&
           ((a.pool == 1 && a.hd > 64) || (a.pool == 2 && a.qpool));
}
static int attn4_launch(const AttnArgs& a, hipStream_t s) {
    const float c = (1.0f / sqrtf((float)a.hd)) * 1.4426950408889634f;
    const long nitems_l = (long)a.N * (a.H / 4) * (a.W / 4) * a.nh;
    LSAM_REQUIRE(nitems_l < (1L << 31), LSAM_ERR_UNSUPPORTED, "attention: too many windows");
    const int nitems = (int)nitems_l;
    const int nblk = min(cdiv(nitems, 4), 256 * 8);     // strided item lists: up to 8 workgroups per CU
    const dim3 grid(nblk), blk(256);
#define LSAM_A4(KSD, DT)                                                                          \
    do {                                                                                          \
        if (a.pool == 2) hipLaunchKernelGGL((attn4_fwd_kernel<KSD, DT, true>), grid, blk, 0, s, a, c, nitems);   \
        else hipLaunchKernelGGL((attn4_fwd_kernel<KSD, DT, false>), grid, blk, 0, s, a, c, nitems);              \
    } while (0)
    if (a.hd <= 64) LSAM_A4(2, 4);
    else if (a.hd <= 80) LSAM_A4(3, 5);
    else LSAM_A4(3, 6);
#undef LSAM_A4
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// head_dim 65..96 (Hiera-L 72, Hiera-T/S 96): the same streaming kernel on 256-byte tile rows, two query tiles per wave
static int attn2_wide_launch(const AttnArgs& a, int max_nq, hipStream_t s) {
    const WinGeom G = make_geom(a.H, a.W, a.ws, a.pool);
    const float c = (1.0f / sqrtf((float)a.hd)) * 1.4426950408889634f;
    const int nqb = cdiv(max_nq, 128);
    const dim3 grid(8 * cdiv(a.N * G.nwy * G.nwx, 8) * a.nh * nqb), blk(256);   // see the kernel: XCD-aware 1-D order
    if (a.hd <= 80) hipLaunchKernelGGL((attn2_fwd_kernel<2, false, 3, 5>), grid, blk, 0, s, a, c, nqb);
    else hipLaunchKernelGGL((attn2_fwd_kernel<2, false, 3, 6>), grid, blk, 0, s, a, c, nqb);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

template <int QT> static int attn2_launch(const AttnArgs& a, int nqb, hipStream_t s) {
    const WinGeom G = make_geom(a.H, a.W, a.ws, a.pool);
    const float c = (1.0f / sqrtf((float)a.hd)) * 1.4426950408889634f;
    const dim3 grid(8 * cdiv(a.N * G.nwy * G.nwx, 8) * a.nh * nqb), blk(256);   // see the kernel: XCD-aware 1-D order
    if (a.hd % 8 == 0 && a.hd <= 56) hipLaunchKernelGGL((attn2_fwd_kernel<QT, true>), grid, blk, 0, s, a, c, nqb);
    else hipLaunchKernelGGL((attn2_fwd_kernel<QT, false>), grid, blk, 0, s, a, c, nqb);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

// HDP = head_dim padded to the LDS row the kernel is built for: 64 (Hiera-T/S/B+: 56 ... 64) or 128
// (Hiera-L: 72).  Rows stay a power-of-two number of 16-byte chunks, which the XOR swizzle needs.
template <typename T, int QT>
static int attn_fwd_launch(const AttnArgs& a, int nwaves, int nqb, hipStream_t s) {
    const WinGeom G = make_geom(a.H, a.W, a.ws, a.pool);
    const float c = (1.0f / sqrtf((float)a.hd)) * 1.4426950408889634f;
    if (QT == 1 && nwaves == 1 && nqb == 1 && a.nh >= 2 && a.hd <= 64) {   // <= 16 queries: one wave per head
        const dim3 grid(a.N * G.nwy * G.nwx, cdiv(a.nh, 4), 1), blk(256);
        hipLaunchKernelGGL((attn_fwd_kernel<T, 1, 64, 4>), grid, blk, 0, s, a, c);
        LSAM_LAUNCH_CHECK();
        return LSAM_OK;
    }
    const dim3 grid(a.N * G.nwy * G.nwx, a.nh, nqb), blk(64 * nwaves);
    if (a.hd <= 64) hipLaunchKernelGGL((attn_fwd_kernel<T, QT, 64, 1>), grid, blk, 0, s, a, c);
    else hipLaunchKernelGGL((attn_fwd_kernel<T, QT, 128, 1>), grid, blk, 0, s, a, c);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

template <typename T> int launch_attn_fwd(const AttnArgs& a, hipStream_t s) {
    constexpr int EPC = Tr<T>::EPC;
    LSAM_REQUIRE(a.qkv && a.bias && a.out, LSAM_ERR_BAD_ARG, "attention: null pointer");
    LSAM_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.nh > 0 && a.hd > 0, LSAM_ERR_BAD_ARG, "attention: empty shape");
    LSAM_REQUIRE(a.hd % 8 == 0 && a.hd <= 128, LSAM_ERR_UNSUPPORTED, "attention: head_dim %d (need %%8==0, <=128)", a.hd);
    LSAM_REQUIRE(a.pool == 1 || a.pool == 2, LSAM_ERR_UNSUPPORTED, "attention: q-pool stride %d", a.pool);
    LSAM_REQUIRE(!a.qpool || (a.pool == 2 && a.ldqp >= a.nh * a.hd && a.ldqp % 4 == 0 && aligned16(a.qpool)),
                 LSAM_ERR_BAD_ARG, "attention: pooled q needs pool == 2, ld %% 4 == 0 and 16-byte alignment");
    LSAM_REQUIRE(a.ws >= 0 && (a.pool == 1 || (a.ws > 0 && a.ws % 2 == 0 && a.H % 2 == 0 && a.W % 2 == 0)),
                 LSAM_ERR_UNSUPPORTED, "attention: pooled blocks need an even window and grid (ws=%d)", a.ws);
    LSAM_REQUIRE(a.ldqkv >= 3 * a.nh * a.hd && a.ldqkv % EPC == 0 && a.ldo >= a.nh * a.hd && a.ldo % 4 == 0 &&
                     aligned16(a.qkv) && aligned16(a.out),
                 LSAM_ERR_BAD_ARG, "attention: leading dimensions / alignment");
    LSAM_REQUIRE(a.nh <= 65535, LSAM_ERR_UNSUPPORTED, "attention: too many heads");
    const WinGeom G = make_geom(a.H, a.W, a.ws, a.pool);
    const int max_nq = G.qh * G.qw;
    // algorithmic FLOPs: every real query against all keys of its window (pad-token keys included), QK^T + PV
    ProfScope ps(s, 4.0 * a.N * (a.H / a.pool) * (a.W / a.pool) * (double)(G.wsh * G.wsw) * a.nh * a.hd,
                 a.ws == 0 ? PROF_ATTN_GLOBAL : PROF_ATTN_WIN, a.N * (a.H / a.pool) * (a.W / a.pool), G.wsh * G.wsw, a.nh * a.hd);
    if constexpr (sizeof(T) == 2) {
        if (attn4_supported(a)) return attn4_launch(a, s);
        if (attn8_supported(a)) return attn8_launch(a, s);   // 8 x 8 windows, head_dim 72 .. 96: the same scheme on 64 keys   // 4 x 4 windows: one wave per (window, head), V alone through LDS
        if (a.bias_t && a.hd <= 64 && a.ws > 0 && G.wsh * G.wsw > 64 && G.wsh * G.wsw <= 208 && max_nq > 64 && a.ldqkv % 8 == 0)
            return attnw_launch(a, s);   // (pooled windows have few queries per staged K/V tile: the streaming kernel below)
        // (not for a forward that feeds the backward: this kernel rounds the PRE-SCALED queries to bf16, so its log-sum-exp
        // belongs to slightly different scores than the ones lsam_attention_bwd recomputes -- 13 % error in dA_q measured)
        if (a.bias_t && !a.lse && a.hd > 64 && a.hd <= 96 && G.wsh * G.wsw >= 64 && max_nq >= 32 && a.ldqkv % 8 == 0)
            return attn2_wide_launch(a, max_nq, s);
        if (a.bias_t && a.hd <= 64 && G.wsh * G.wsw >= 128 && max_nq >= 32 && a.ldqkv % 8 == 0) {
            if (max_nq > 64) return attn2_launch<ATTN2_QT>(a, cdiv(max_nq, 64 * ATTN2_QT), s);
            return attn2_launch<1>(a, 1, s);
        }
    }
    if (max_nq > 64) {
        const int nwaves = min(4, cdiv(max_nq, 32));
        return attn_fwd_launch<T, 2>(a, nwaves, cdiv(max_nq, nwaves * 32), s);
    }
    const int nwaves = cdiv(max_nq, 16);
    return attn_fwd_launch<T, 1>(a, nwaves, 1, s);
}
template int launch_attn_fwd<float>(const AttnArgs&, hipStream_t);
template int launch_attn_fwd<bf16>(const AttnArgs&, hipStream_t);

// ---------------------------------------------------------------------------------------------
// General backward for the LoRA blocks: windows with at most 64 queries and ANY number of keys (streamed in tiles
// of 64), head_dim <= 96, optional 2x2 q-pool (the stage-entry block a two-block last stage puts under
// lora_qkv.py:162's blocks[-2:]: 49 pooled queries against 196 keys).  One workgroup per (window, head), fp32 in LDS:
//   P = exp(S*scale - lse), dP = dO V^T, delta_i = sum_j P_ij dP_ij, dS = P (dP - delta) scale,
//   dQ = dS K, dK = dS^T Q, dV = P^T dO.
// Pass A streams the key tiles once for delta, pass B again for dS and the three products (dQ stays in registers
// across tiles).  Pooled queries: q_i = max over the 2x2 input tokens per channel (do_pool, hieradet.py:25-36,69-71);
// dq_i[c] goes to the FIRST maximal token (MaxPool2d's backward) and the other three get 0, so every q, k, v element
// of a real token is written exactly once and dqkv needs no clearing.  Pad-token keys are constants: no gradient.
// ---------------------------------------------------------------------------------------------
constexpr int BW_T = 64;          // queries per window (max) and keys per tile
constexpr int BW_MAX_HD = 96;
constexpr int BW_NB = BW_MAX_HD / 16;
static size_t attn_bwd_gen_lds(int hd) {
    return (size_t)(4 * BW_T * (hd + 1) + 2 * BW_T * (BW_T + 1) + 2 * BW_T) * sizeof(float) + 6 * BW_T * sizeof(int) +
           (size_t)BW_T * hd;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_gen_kernel(const AttnBwdArgs p, const float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hd = p.hd, LDH = hd + 1, LDP = BW_T + 1;
    float* Qs = sm;
    float* Ks = Qs + BW_T * LDH;
    float* Vs = Ks + BW_T * LDH;
    float* Ds = Vs + BW_T * LDH;   // dO
    float* Ps = Ds + BW_T * LDH;
    float* Ss = Ps + BW_T * LDP;   // dP, then dS
    float* lse_s = Ss + BW_T * LDP;
    float* del_s = lse_s + BW_T;
    int* otok_s = reinterpret_cast<int*>(del_s + BW_T);   // output-grid token of a query row (dO, lse), -1 = none
    int* qsrc_s = otok_s + BW_T;                          // 4 input-grid tokens a (pooled) query is the max of
    int* ktok_s = qsrc_s + 4 * BW_T;                      // input-grid token of a key of the current tile; -1 pad, -2 none
    unsigned char* amax_s = reinterpret_cast<unsigned char*>(ktok_s + BW_T);   // [query][d]: which of the 4 tokens

    const int tid = threadIdx.x;
    const int H = p.H, W = p.W, nh = p.nh, dout = nh * hd, ws = p.ws, pool = p.pool;
    const int nwy = (H + ws - 1) / ws, nwx = (W + ws - 1) / ws;
    const int h = blockIdx.y;
    const int win = blockIdx.x % (nwy * nwx), n = blockIdx.x / (nwy * nwx);
    const int wy = win / nwx, wx = win % nwx;
    const int qs = ws / pool, Ho = H / pool, Wo = W / pool;
    const int rh = min(qs, Ho - wy * qs), rw = min(qs, Wo - wx * qs);
    const int nq = rh > 0 && rw > 0 ? rh * rw : 0;
    const int nk = ws * ws;
    if (nq == 0) return;   // a window of pad tokens only (uniform for the workgroup)
    const T* __restrict__ qkv = static_cast<const T*>(p.qkv);
    const T* __restrict__ dO = static_cast<const T*>(p.dout);
    const size_t img_in = (size_t)n * H * W, img_out = (size_t)n * Ho * Wo;

    if (tid < BW_T) {
        int ot = -1;
        int src[4] = {-1, -1, -1, -1};
        if (tid < nq) {
            const int py = tid / rw, px = tid % rw;
            ot = (int)(img_out + (size_t)(wy * qs + py) * Wo + (wx * qs + px));
            const int y0 = wy * ws + pool * py, x0 = wx * ws + pool * px;
            src[0] = (int)(img_in + (size_t)y0 * W + x0);
            if (pool == 2) {
                src[1] = src[0] + 1;
                src[2] = src[0] + W;
                src[3] = src[0] + W + 1;
            }
        }
        otok_s[tid] = ot;
#pragma unroll
        for (int k = 0; k < 4; ++k) qsrc_s[4 * tid + k] = src[k];
        lse_s[tid] = ot >= 0 ? p.lse[(size_t)ot * nh + h] : 0.f;
        del_s[tid] = 0.f;
    }
    __syncthreads();
    for (int i = tid; i < BW_T * hd; i += 256) {
        const int r = i / hd, d = i % hd;
        const int ot = otok_s[r];
        float q = 0.f, dv = 0.f;
        int am = 0;
        if (ot >= 0) {
            q = to_f(qkv[(size_t)qsrc_s[4 * r] * p.ldqkv + h * hd + d]);
            if (pool == 2) {
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    const float v = to_f(qkv[(size_t)qsrc_s[4 * r + k] * p.ldqkv + h * hd + d]);
                    if (v > q) { q = v; am = k; }   // strict: the first maximum keeps the gradient
                }
            }
            dv = to_f(dO[(size_t)ot * p.lddo + h * hd + d]);
        }
        Qs[r * LDH + d] = q;
        Ds[r * LDH + d] = dv;
        amax_s[r * hd + d] = (unsigned char)am;
    }

    const int ti = tid >> 4, tj = tid & 15;
    const int nkt = (nk + BW_T - 1) / BW_T;
    float dq[4][BW_NB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < BW_NB; ++b) dq[a][b] = 0.f;

#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll 1
        for (int kt = 0; kt < nkt; ++kt) {
            __syncthreads();   // previous tile fully consumed (and Q / dO / del_s visible on the first trip)
            if (tid < BW_T) {
                const int j = kt * BW_T + tid;
                int tk = -2;
                if (j < nk) {
                    const int y = wy * ws + j / ws, x = wx * ws + j % ws;
                    tk = (y < H && x < W) ? (int)(img_in + (size_t)y * W + x) : -1;
                }
                ktok_s[tid] = tk;
            }
            __syncthreads();
            for (int i = tid; i < BW_T * hd; i += 256) {
                const int r = i / hd, d = i % hd;
                const int tk = ktok_s[r];
                float k = 0.f, v = 0.f;
                if (tk >= 0) {
                    k = to_f(qkv[(size_t)tk * p.ldqkv + dout + h * hd + d]);
                    v = to_f(qkv[(size_t)tk * p.ldqkv + 2 * dout + h * hd + d]);
                } else if (tk == -1) {
                    k = to_f(from_f<T>(p.bias[dout + h * hd + d]));
                    v = to_f(from_f<T>(p.bias[2 * dout + h * hd + d]));
                }
                Ks[r * LDH + d] = k;
                Vs[r * LDH + d] = v;
            }
            __syncthreads();
            {
                float s[4][4], dp[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) s[a][b] = dp[a][b] = 0.f;
                for (int d = 0; d < hd; ++d) {
                    float qa[4], da[4], kb[4], vb[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        qa[a] = Qs[(ti + 16 * a) * LDH + d];
                        da[a] = Ds[(ti + 16 * a) * LDH + d];
                        kb[a] = Ks[(tj + 16 * a) * LDH + d];
                        vb[a] = Vs[(tj + 16 * a) * LDH + d];
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            s[a][b] += qa[a] * kb[b];
                            dp[a][b] += da[a] * vb[b];
                        }
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int i = ti + 16 * a, j = tj + 16 * b;
                        const bool ok = i < nq && kt * BW_T + j < nk;
                        const float pe = ok ? __expf(s[a][b] * scale - lse_s[i]) : 0.f;
                        Ps[i * LDP + j] = pe;
                        Ss[i * LDP + j] = pass == 0 ? pe * dp[a][b] : pe * (dp[a][b] - del_s[i]) * scale;
                    }
            }
            __syncthreads();
            if (pass == 0) {
                if (tid < BW_T) {
                    float dl = 0.f;
                    for (int j = 0; j < BW_T; ++j) dl += Ss[tid * LDP + j];
                    del_s[tid] += dl;
                }
                continue;
            }
            // dQ += dS K (rows ti+16a, channels tj+16b);  dK = dS^T Q,  dV = P^T dO  (key rows ti+16a of this tile)
            float dk[4][BW_NB], dvv[4][BW_NB];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < BW_NB; ++b) dk[a][b] = dvv[a][b] = 0.f;
            for (int j = 0; j < BW_T; ++j) {
                float ds_r[4], ds_c[4], p_c[4], kd[BW_NB], qd[BW_NB], dd[BW_NB];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    ds_r[a] = Ss[(ti + 16 * a) * LDP + j];   // dS[i][j]   (dQ: sum over keys j)
                    ds_c[a] = Ss[j * LDP + (ti + 16 * a)];   // dS[j][i']  (dK: sum over queries j)
                    p_c[a] = Ps[j * LDP + (ti + 16 * a)];    // P[j][i']   (dV)
                }
#pragma unroll
                for (int b = 0; b < BW_NB; ++b) {
                    const int d = tj + 16 * b;
                    const bool d_ok = d < hd;
                    kd[b] = d_ok ? Ks[j * LDH + d] : 0.f;
                    qd[b] = d_ok ? Qs[j * LDH + d] : 0.f;
                    dd[b] = d_ok ? Ds[j * LDH + d] : 0.f;
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < BW_NB; ++b) {
                        dq[a][b] += ds_r[a] * kd[b];
                        dk[a][b] += ds_c[a] * qd[b];
                        dvv[a][b] += p_c[a] * dd[b];
                    }
            }
            T* __restrict__ dqkv = static_cast<T*>(p.dqkv);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int tk = ktok_s[ti + 16 * a];
                if (tk < 0) continue;
#pragma unroll
                for (int b = 0; b < BW_NB; ++b) {
                    const int d = tj + 16 * b;
                    if (d >= hd) continue;
                    dqkv[(size_t)tk * p.lddqkv + dout + h * hd + d] = from_f<T>(dk[a][b]);
                    dqkv[(size_t)tk * p.lddqkv + 2 * dout + h * hd + d] = from_f<T>(dvv[a][b]);
                }
            }
        }
    }
    T* __restrict__ dqkv = static_cast<T*>(p.dqkv);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int r = ti + 16 * a;
        if (otok_s[r] < 0) continue;
#pragma unroll
        for (int b = 0; b < BW_NB; ++b) {
            const int d = tj + 16 * b;
            if (d >= hd) continue;
            if (pool == 1) {
                dqkv[(size_t)qsrc_s[4 * r] * p.lddqkv + h * hd + d] = from_f<T>(dq[a][b]);
            } else {
                const int am = amax_s[r * hd + d];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    dqkv[(size_t)qsrc_s[4 * r + k] * p.lddqkv + h * hd + d] = from_f<T>(k == am ? dq[a][b] : 0.f);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 MFMA backward.  One WAVE per (window, head); a workgroup is 4 heads of one window and uses
// no workgroup barrier: every wave stages its own Q, K, V, dO tiles ([64 rows][64 d] bf16, 128-byte
// rows, chunk-XOR swizzle) in a private 32 KiB slice of LDS and reads them back as MFMA operands:
//   S^T = K Q^T, dP^T = V dO^T      (lane owns one query)  -> delta_i, dS^T -> dQ^T = K^T dS^T
//   S   = Q K^T, dP   = dO V^T      (lane owns one key)    -> P, dS        -> dV^T = dO^T P, dK^T = Q^T dS
// The scores are recomputed in both orientations (cheap: the kernel is bound by its ~200 MB of
// global traffic, not by its 224 MFMAs per head-window) so that P / dS always leave the MFMA in the
// register layout the next product wants as its B operand; the A operands K^T, dO^T, Q^T come out
// of the row-major tiles through the transposing LDS read ds_read_b64_tr_b16.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(const AttnBwdArgs p, const float scale, const float c_log2) {
    using T = bf16;
    constexpr int RB = 128, TILE_B = 64 * RB, WAVE_B = 4 * TILE_B + 512;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, lr = lane & 15;
    const int h = blockIdx.y * 4 + wave;
    if (h >= p.nh) return;   // wave-uniform; the kernel has no workgroup barrier

    char* qb = smem + wave * WAVE_B;
    char* kb = qb + TILE_B;
    char* vb = kb + TILE_B;
    char* ob = vb + TILE_B;
    float* lse_s = reinterpret_cast<float*>(ob + TILE_B);
    float* del_s = lse_s + 64;

    const int H = p.H, W = p.W, nh = p.nh, hd = p.hd, dout = nh * hd, ws = p.ws;
    const int nwy = (H + ws - 1) / ws, nwx = (W + ws - 1) / ws;
    const int win = blockIdx.x % (nwy * nwx), n = blockIdx.x / (nwy * nwx);
    const int wy = win / nwx, wx = win % nwx;
    const int rh = min(ws, H - wy * ws), rw = min(ws, W - wx * ws);
    const int nq = rh * rw, nk = ws * ws;
    const T* __restrict__ qkv = static_cast<const T*>(p.qkv);
    const T* __restrict__ dO = static_cast<const T*>(p.dout);
    const size_t img_tok = (size_t)n * H * W;

    // lane l describes row l of the tiles: query token, key token (-1 = pad key, -2 = no key)
    int qtok_l = -1, ktok_l = -2;
    if (lane < nq) qtok_l = (int)(img_tok + (size_t)(wy * ws + lane / rw) * W + (wx * ws + lane % rw));
    if (lane < nk) {
        const int y = wy * ws + lane / ws, x = wx * ws + lane % ws;
        ktok_l = (y < H && x < W) ? (int)(img_tok + (size_t)y * W + x) : -1;
    }
    lse_s[lane] = qtok_l >= 0 ? p.lse[(size_t)qtok_l * nh + h] * 1.4426950408889634f : 0.f;

    // ---- stage the four tiles --------------------------------------------------------------------
    {
        const int ch = lane & 7, d0 = ch * 8;
        auto pack_bias = [&](const float* b) -> uint4 {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = b[j];
            return __builtin_bit_cast(uint4, frag_from_f<T>(v));
        };
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int row = 8 * c + (lane >> 3);
            const int qt = __shfl(qtok_l, row, 64), kt = __shfl(ktok_l, row, 64);
            uint4 q = make_uint4(0u, 0u, 0u, 0u), dv = q, k = q, v = q;
            if (d0 < hd) {
                if (qt >= 0) {
                    q = *reinterpret_cast<const uint4*>(qkv + (size_t)qt * p.ldqkv + h * hd + d0);
                    dv = *reinterpret_cast<const uint4*>(dO + (size_t)qt * p.lddo + h * hd + d0);
                }
                if (kt >= 0) {
                    k = *reinterpret_cast<const uint4*>(qkv + (size_t)kt * p.ldqkv + dout + h * hd + d0);
                    v = *reinterpret_cast<const uint4*>(qkv + (size_t)kt * p.ldqkv + 2 * dout + h * hd + d0);
                } else if (kt == -1) {
                    k = pack_bias(p.bias + dout + h * hd + d0);
                    v = pack_bias(p.bias + 2 * dout + h * hd + d0);
                }
            }
            const int off = row * RB + (swz(row, ch) << 4);
            *reinterpret_cast<uint4*>(qb + off) = q;
            *reinterpret_cast<uint4*>(ob + off) = dv;
            *reinterpret_cast<uint4*>(kb + off) = k;
            *reinterpret_cast<uint4*>(vb + off) = v;
        }
    }

    auto row_frag = [&](const char* base, int t, int ks) -> bf16x8 {
        const int row = t * 16 + lr;
        return *reinterpret_cast<const bf16x8*>(base + row * RB + (swz(row, ks * 4 + g) << 4));
    };
    // A operand X^T[d = 16 dt + ..][k-slot]: slot 8g + 4t + r <-> tile row 32 u + 16 t + 4 g + r
    auto tr_frag = [&](const char* base, int u, int dt) -> bf16x8 {
        const int row = 32 * u + 4 * g + (lr >> 2), ch = 2 * dt + ((lr & 3) >> 1);
        const char* a0 = base + row * RB + (swz(row, ch) << 4) + (lr & 1) * 8;
        const int row1 = row + 16;
        const char* a1 = base + row1 * RB + (swz(row1, ch) << 4) + (lr & 1) * 8;
        const s16x4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(a0));
        const s16x4_t x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(a1));
        const uint2 u0 = __builtin_bit_cast(uint2, x0), u1 = __builtin_bit_cast(uint2, x1);
        return __builtin_bit_cast(bf16x8, make_uint4(u0.x, u0.y, u1.x, u1.y));
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    T* __restrict__ dqkv = static_cast<T*>(p.dqkv);

    // ---- pass 1: lane owns query i = 16 it + lr.  delta_i, dS^T, dQ --------------------------------
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
        if (it * 16 >= nq) {   // wave-uniform: no query in this tile
            if (g == 0) del_s[it * 16 + lr] = 0.f;
            continue;
        }
        const bf16x8 qf0 = row_frag(qb, it, 0), qf1 = row_frag(qb, it, 1);
        const bf16x8 of0 = row_frag(ob, it, 0), of1 = row_frag(ob, it, 1);
        const int i = it * 16 + lr;
        const float lse2 = lse_s[i];
        float pr[4][4], dpv[4][4];
        float delta = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(kb, jt, 0), qf0, zero4, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(kb, jt, 1), qf1, s, 0, 0, 0);
            f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(vb, jt, 0), of0, zero4, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(vb, jt, 1), of1, dp, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jt * 16 + 4 * g + r;
                const float e = (i < nq && j < nk) ? __builtin_amdgcn_exp2f(fmaf(s[r], c_log2, -lse2)) : 0.f;
                pr[jt][r] = e;
                dpv[jt][r] = dp[r];
                delta = fmaf(e, dp[r], delta);
            }
        }
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
        if (g == 0) del_s[i] = delta;
        bf16x8 dsf[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float v[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[t * 4 + r] = pr[2 * u + t][r] * (dpv[2 * u + t][r] - delta) * scale;
            dsf[u] = frag_from_f<T>(v);
        }
        const int tokq = __shfl(qtok_l, i, 64);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4 acc = zero4;
            mma16(tr_frag(kb, 0, dt), dsf[0], acc);
            mma16(tr_frag(kb, 1, dt), dsf[1], acc);
            const int d0 = dt * 16 + 4 * g;
            if (tokq >= 0 && d0 < hd) store4(dqkv + (size_t)tokq * p.lddqkv + h * hd + d0, acc);
        }
    }

    // ---- pass 2: lane owns key j = 16 jt + lr.  P, dS -> dV, dK ------------------------------------
#pragma unroll 1
    for (int jt = 0; jt < 4; ++jt) {
        if (jt * 16 >= nk) break;   // wave-uniform
        const bf16x8 kf0 = row_frag(kb, jt, 0), kf1 = row_frag(kb, jt, 1);
        const bf16x8 vf0 = row_frag(vb, jt, 0), vf1 = row_frag(vb, jt, 1);
        const int j = jt * 16 + lr;
        float pr[4][4], dsv[4][4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(qb, it, 0), kf0, zero4, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(qb, it, 1), kf1, s, 0, 0, 0);
            f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(ob, it, 0), vf0, zero4, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(ob, it, 1), vf1, dp, 0, 0, 0);
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + it * 16 + 4 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + it * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = it * 16 + 4 * g + r;
                const float e = (i < nq && j < nk) ? __builtin_amdgcn_exp2f(fmaf(s[r], c_log2, -l4[r])) : 0.f;
                pr[it][r] = e;
                dsv[it][r] = e * (dp[r] - d4[r]) * scale;
            }
        }
        bf16x8 pf[2], dsf[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float a[8], b[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a[t * 4 + r] = pr[2 * u + t][r];
                    b[t * 4 + r] = dsv[2 * u + t][r];
                }
            pf[u] = frag_from_f<T>(a);
            dsf[u] = frag_from_f<T>(b);
        }
        const int tokk = __shfl(ktok_l, j, 64);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4 av = zero4, ak = zero4;
            mma16(tr_frag(ob, 0, dt), pf[0], av);
            mma16(tr_frag(ob, 1, dt), pf[1], av);
            mma16(tr_frag(qb, 0, dt), dsf[0], ak);
            mma16(tr_frag(qb, 1, dt), dsf[1], ak);
            const int d0 = dt * 16 + 4 * g;
            if (tokk >= 0 && d0 < hd) {
                store4(dqkv + (size_t)tokk * p.lddqkv + dout + h * hd + d0, ak);
                store4(dqkv + (size_t)tokk * p.lddqkv + 2 * dout + h * hd + d0, av);
            }
        }
    }
}

template <typename T> int launch_attn_bwd(const AttnBwdArgs& a, hipStream_t s) {
    LSAM_REQUIRE(a.qkv && a.bias && a.dout && a.lse && a.dqkv, LSAM_ERR_BAD_ARG, "attention_bwd: null pointer");
    LSAM_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.nh > 0, LSAM_ERR_BAD_ARG, "attention_bwd: empty shape");
    LSAM_REQUIRE(a.pool == 1 || a.pool == 2, LSAM_ERR_UNSUPPORTED, "attention_bwd: q-pool stride %d", a.pool);
    LSAM_REQUIRE(a.ws > 0 && (a.ws / a.pool) * (a.ws / a.pool) <= BW_T, LSAM_ERR_UNSUPPORTED,
                 "attention_bwd: window %d (q-pool %d) has more than %d queries, or is global: not supported", a.ws, a.pool, BW_T);
    LSAM_REQUIRE(a.pool == 1 || (a.ws % 2 == 0 && a.H % 2 == 0 && a.W % 2 == 0), LSAM_ERR_UNSUPPORTED,
                 "attention_bwd: pooled blocks need an even window and grid (ws=%d)", a.ws);
    LSAM_REQUIRE(a.hd > 0 && a.hd <= BW_MAX_HD, LSAM_ERR_UNSUPPORTED, "attention_bwd: head_dim %d > %d", a.hd, BW_MAX_HD);
    LSAM_REQUIRE(a.ldqkv >= 3 * a.nh * a.hd && a.lddqkv >= 3 * a.nh * a.hd && a.lddo >= a.nh * a.hd,
                 LSAM_ERR_BAD_ARG, "attention_bwd: leading dimensions");
    const int nwy = cdiv(a.H, a.ws), nwx = cdiv(a.W, a.ws);
    ProfScope ps(s, 10.0 * a.N * (a.H / a.pool) * (a.W / a.pool) * (double)(a.ws * a.ws) * a.nh * a.hd, PROF_ATTN_BWD,
                 a.N * (a.H / a.pool) * (a.W / a.pool), a.ws * a.ws, a.nh * a.hd);   // 2.5 x the forward's 4 nq nk hd
    if constexpr (sizeof(T) == 2) {
        const bool al16 = ((reinterpret_cast<uintptr_t>(a.qkv) | reinterpret_cast<uintptr_t>(a.dout)) & 15) == 0;
        if (a.pool == 1 && a.ws * a.ws <= 64 && a.hd <= 64 && a.hd % 8 == 0 && a.ldqkv % 8 == 0 && a.lddo % 8 == 0 &&
            a.lddqkv % 4 == 0 && al16 && (reinterpret_cast<uintptr_t>(a.dqkv) & 7) == 0) {
            constexpr int LDSB = 4 * (4 * 64 * 128 + 512);
            static thread_local bool attr_set = false;
            if (!attr_set) {
                LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_mfma_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
                attr_set = true;
            }
            const float sc = 1.0f / sqrtf((float)a.hd);
            hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3(a.N * nwy * nwx, cdiv(a.nh, 4)), dim3(256), LDSB, s, a, sc,
                               sc * 1.4426950408889634f);
            LSAM_LAUNCH_CHECK();
            return LSAM_OK;
        }
    }
    const size_t lds = attn_bwd_gen_lds(a.hd);
    auto kern = attn_bwd_gen_kernel<T>;
    LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)attn_bwd_gen_lds(BW_MAX_HD)));
    hipLaunchKernelGGL(kern, dim3(a.N * nwy * nwx, a.nh), dim3(256), lds, s, a, 1.0f / sqrtf((float)a.hd));
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
template int launch_attn_bwd<float>(const AttnBwdArgs&, hipStream_t);
template int launch_attn_bwd<bf16>(const AttnBwdArgs&, hipStream_t);

}  // namespace lsam
