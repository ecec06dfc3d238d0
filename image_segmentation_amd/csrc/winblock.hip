// Fused attention half of a same-dim 8x8-window MultiScaleBlock (hieradet.py:134-163 for the stage-1
// blocks):   x <- x + proj(attn(qkv(LN1(x))))   in ONE kernel, one HBM read and one HBM write of x.
//
// Un-fused, a stage-1 block moves LN1 out, qkv, the attention output and the proj input through HBM
// (~3.8 GB per block at batch 16) for ~40 GFLOP of work, so every one of those kernels is HBM-bound.
// Here one WAVE owns one window (64 tokens) from the LayerNorm to the residual add and keeps every
// activation in registers, in MFMA operand layouts chosen so that no transpose through LDS is needed:
//   * xn is built directly as k-contiguous operand fragments (lane = token, 8 consecutive channels);
//   * q and k come out of a "swapped" product  D = W . xn^T   (lane = token, 4 consecutive channels),
//     which, with the channel order permuted consistently in both, IS the operand layout of
//     S^T = K . Q^T;
//   * v comes out of the un-swapped product  D = xn . W^T   (lane = channel, 4 consecutive tokens),
//     which is the A-operand layout of O^T = V^T . P^T;  P leaves the softmax in the B layout;
//   * O^T (lane = token, 4 consecutive channels) is the B operand of the proj product; the proj
//     weight's K order is permuted to match when it is packed.
// The block's weights (qkv with every head padded to 64 channels, proj) are packed once into an LDS
// image (128 KiB for two heads) that each persistent workgroup copies in at start and then only
// reads: the steady state has no workgroup barrier.
#include "kernels.hpp"

namespace lsam {

// LDS image: [part s][head h][k-subtile kk] sub-tiles of [64 rows][64 k] bf16 (128-byte rows, chunk-XOR
// swizzle), then proj sub-tiles [head h] of [64 NH rows][64 slots].
template <int NH> struct WinImg {
    static constexpr int SUB = 64 * 128;                      // bytes of one [64][64] sub-tile
    static constexpr int QKV_B = 3 * NH * NH * SUB;
    static constexpr int NP = 64 * NH;                        // proj rows (output channels, padded)
    static constexpr int PROJ_B = NH * NP * 128;
    static constexpr int BYTES = QKV_B + PROJ_B;
};

// slot s (0..63) of a head's padded channel block <-> channel: slot 32 ks + 8 g + 4 t + r holds
// channel 32 ks + 16 t + 4 g + r (what a lane of the D layout owns in tiles 2 ks and 2 ks + 1)
__host__ __device__ inline int slot_to_ch(int s) {
    const int ks = s >> 5, g = (s >> 3) & 3, t = (s >> 2) & 1, r = s & 3;
    return 32 * ks + 16 * t + 4 * g + r;
}

template <int NH>
__global__ void pack_winblock_kernel(const float* __restrict__ wqkv, const float* __restrict__ bqkv,
                                     const float* __restrict__ wproj, const float* __restrict__ bproj, int hd,
                                     char* __restrict__ img, float* __restrict__ bias_p) {
    using I = WinImg<NH>;
    const int D = NH * hd;
    const int total = I::BYTES / 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int byte = 2 * i;
        float v = 0.f;
        if (byte < I::QKV_B) {
            const int sub = byte / I::SUB, in = byte % I::SUB;
            const int kk = sub % NH, h = (sub / NH) % NH, s = sub / (NH * NH);
            const int row = in / 128, chunk = ((in % 128) >> 4) ^ (row & 7), k = 64 * kk + 8 * chunk + ((in & 15) >> 1);
            if (row < hd && k < D) v = wqkv[(size_t)(s * D + h * hd + row) * D + k];
        } else {
            const int b = byte - I::QKV_B;
            const int h = b / (I::NP * 128), in = b % (I::NP * 128);
            const int row = in / 128, chunk = ((in % 128) >> 4) ^ (row & 7), slot = 8 * chunk + ((in & 15) >> 1);
            const int ch = slot_to_ch(slot);
            if (row < D && ch < hd) v = wproj[(size_t)row * D + h * hd + ch];
        }
        reinterpret_cast<bf16*>(img)[i] = (bf16)v;
    }
    // padded biases: qkv [3][NH][64], then proj [64 NH]
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * NH * 64 + I::NP; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < 3 * NH * 64) {
            const int e = i % 64, h = (i / 64) % NH, s = i / (64 * NH);
            if (e < hd) v = bqkv[s * D + h * hd + e];
        } else {
            const int n = i - 3 * NH * 64;
            if (n < D) v = bproj[n];
        }
        bias_p[i] = v;
    }
}

#ifndef WINBLOCK_NW
#define WINBLOCK_NW 4   // waves per workgroup (one workgroup per CU: the weight image fills the LDS)
#endif
#ifndef WINBLOCK_NW_ATTN
#define WINBLOCK_NW_ATTN 4   // the same for the attention-only form
#endif
template <int NH, bool ATTN_ONLY, int NW>
__global__ __launch_bounds__(64 * NW, 1) void winblock_attn_kernel(const WinBlockArgs p) {
    using I = WinImg<NH>;
    constexpr int KS = 2 * NH;      // 32-wide k-steps over the model dim padded to 64 NH
    constexpr int NTO = 4 * NH;     // output column tiles of proj
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lr = lane & 15;
    const int hd = p.hd, D = NH * hd;

    // ---- weights: linear copy of the packed image, once per workgroup -----------------------------
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.img);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < I::BYTES / 16; i += 64 * NW) dst[i] = src[i];
    }
    __syncthreads();

    auto wfrag = [&](int sub_off, int row, int c) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(smem + sub_off + row * 128 + ((c ^ (row & 7)) << 4));
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int nwx = p.W / 8, nwy = p.H / 8;
    const int nwin = p.N * nwy * nwx;
    const float* __restrict__ bias_p = p.bias_p;

#pragma unroll 1
    for (int w = blockIdx.x * NW + wave; w < nwin; w += gridDim.x * NW) {
        const int n = w / (nwy * nwx), wi = w % (nwy * nwx);
        const int wy = wi / nwx, wx = wi % nwx;
        // token row of tile row t (0..63): (t / 8, t % 8) inside the window
        auto tok_row = [&](int t) -> size_t {
            return ((size_t)n * p.H + (wy * 8 + (t >> 3))) * p.W + (wx * 8 + (t & 7));
        };

        // ---- LayerNorm 1 straight into operand fragments -------------------------------------------
        bf16x8 xn[4][KS];
        {
            float xr[4][KS][8];
            float sum[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float* row = p.x + tok_row(mt * 16 + lr) * D;
                sum[mt] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int c0 = 32 * ks + 8 * g;
                    f32x4 a = zero4, b = zero4;
                    if (c0 < D) { a = load4(row + c0); b = load4(row + c0 + 4); }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { xr[mt][ks][j] = a[j]; xr[mt][ks][4 + j] = b[j]; }
                    sum[mt] += ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
                }
            }
            float mean[4], rstd[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                float s = sum[mt];
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                mean[mt] = s / (float)D;
                float q = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    if (32 * ks + 8 * g < D) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float d = xr[mt][ks][j] - mean[mt];
                            q += d * d;
                        }
                    }
                q += __shfl_xor(q, 16, 64);
                q += __shfl_xor(q, 32, 64);
                rstd[mt] = 1.0f / sqrtf(q / (float)D + p.eps);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c0 = 32 * ks + 8 * g;
                const bool ok = c0 < D;
                f32x4 g0 = zero4, g1 = zero4, b0 = zero4, b1 = zero4;
                if (ok) {
                    g0 = load4(p.gamma + c0); g1 = load4(p.gamma + c0 + 4);
                    b0 = load4(p.beta + c0);  b1 = load4(p.beta + c0 + 4);
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    bf16x8 f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f[j] = (bf16)(ok ? (xr[mt][ks][j] - mean[mt]) * rstd[mt] * g0[j] + b0[j] : 0.f);
                        f[4 + j] = (bf16)(ok ? (xr[mt][ks][4 + j] - mean[mt]) * rstd[mt] * g1[j] + b1[j] : 0.f);
                    }
                    xn[mt][ks] = f;
                }
            }
        }

        // ---- per head: q, k, v, attention; O kept as proj operand fragments ---------------------------
        // (sched_barriers fence the phases: left alone, the scheduler hoists the next phases' LDS reads
        // and global loads far enough up to spill hundreds of registers)
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 of[NH][4][2];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            bf16x8 qf[4][2], kf[4][2], vtf[2][4];
#pragma unroll
            for (int s = 0; s < 2; ++s) {   // q, k: swapped product, lane = token, 4 consecutive channels
                f32x4 acc[4][4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f32x4 b = load4(bias_p + (s * NH + h) * 64 + 16 * nt + 4 * g);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = b;
                }
                // weight fragments one k-step (4 row tiles) ahead of the MFMAs that use them: with one wave per
                // SIMD a read that is waited for right before its MFMAs exposes the whole LDS latency
                bf16x8 wb[2][4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wb[0][nt] = wfrag(((s * NH + h) * NH) * I::SUB, 16 * nt + lr, g);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 1 < KS) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            wb[(ks + 1) & 1][nt] = wfrag(((s * NH + h) * NH + ((ks + 1) >> 1)) * I::SUB, 16 * nt + lr, ((ks + 1) & 1) * 4 + g);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) mma16(wb[ks & 1][nt], xn[mt][ks], acc[mt][nt]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        bf16x8 f;
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) f[4 * t + r] = (bf16)acc[mt][2 * k2 + t][r];
                        if (s == 0) qf[mt][k2] = f; else kf[mt][k2] = f;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            {   // v: un-swapped product, lane = channel, 4 consecutive tokens
                f32x4 acc[4][4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float b = bias_p[(2 * NH + h) * 64 + 16 * nt + lr];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = f32x4{b, b, b, b};
                }
                bf16x8 wb[2][4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wb[0][nt] = wfrag(((2 * NH + h) * NH) * I::SUB, 16 * nt + lr, g);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 1 < KS) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            wb[(ks + 1) & 1][nt] = wfrag(((2 * NH + h) * NH + ((ks + 1) >> 1)) * I::SUB, 16 * nt + lr, ((ks + 1) & 1) * 4 + g);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) mma16(xn[mt][ks], wb[ks & 1][nt], acc[mt][nt]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        bf16x8 f;
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) f[4 * t + r] = (bf16)acc[2 * u + t][dt][r];
                        vtf[u][dt] = f;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // attention of this head: lane owns query 16 it + lr; all 64 keys are real (no padding)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                f32x4 s[4];
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    s[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt][0], qf[it][0], zero4, 0, 0, 0);
                    s[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt][1], qf[it][1], s[jt], 0, 0, 0);
                }
                float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
                for (int jt = 1; jt < 4; ++jt) mx = fmaxf(fmaxf(mx, s[jt][0]), fmaxf(fmaxf(s[jt][1], s[jt][2]), s[jt][3]));
                mx = group4_max(mx);
                const float mneg = -mx * p.c_log2;
                float l = 0.f;
                bf16x8 pf[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    bf16x8 f;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = __builtin_amdgcn_exp2f(fmaf(s[2 * u + t][r], p.c_log2, mneg));
                            l += e;
                            f[4 * t + r] = (bf16)e;
                        }
                    pf[u] = f;
                }
                l = group4_sum(l);
                const float inv = 1.0f / l;
                f32x4 o[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vtf[0][dt], pf[0], zero4, 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vtf[1][dt], pf[1], o[dt], 0, 0, 0);
                }
                if constexpr (ATTN_ONLY) {
                    // the attention output rows as they are (lane: query lr, channels 16 dt + 4 g + r); proj + residual are
                    // taken by the MLP kernel behind this one (mlpblock.hip, PROJ), which is VALU-bound and has the MFMA room
                    bf16* arow = static_cast<bf16*>(p.a_out) + tok_row(it * 16 + lr) * D + h * hd;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        if (16 * dt + 4 * g < hd) store4(arow + 16 * dt + 4 * g, o[dt] * inv);
                } else {
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        bf16x8 f;
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) f[4 * t + r] = (bf16)(o[2 * k2 + t][r] * inv);
                        of[h][it][k2] = f;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (ATTN_ONLY) continue;

        // ---- proj + residual: x1 = x + o . Wproj^T + b  (lane = token, 4 consecutive channels) -------
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // two halves of the column tiles bound the live registers
            constexpr int NTH = NTO / 2;
            f32x4 acc[4][NTH];
#pragma unroll
            for (int q = 0; q < NTH; ++q) {
                const int c0 = 16 * (half * NTH + q) + 4 * g;
                const bool ok = c0 < D;
                const f32x4 b = ok ? load4(bias_p + 3 * NH * 64 + c0) : zero4;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt][q] = ok ? load4(p.x + tok_row(mt * 16 + lr) * D + c0) + b : zero4;
            }
            {   // (head, k-half) steps of NTH fragments, read one step ahead (tiles past D hold zero weights)
                bf16x8 wb[2][NTH];
#pragma unroll
                for (int q = 0; q < NTH; ++q) wb[0][q] = wfrag(I::QKV_B, 16 * (half * NTH + q) + lr, g);
#pragma unroll
                for (int st = 0; st < 2 * NH; ++st) {
                    if (st + 1 < 2 * NH) {
#pragma unroll
                        for (int q = 0; q < NTH; ++q)
                            wb[(st + 1) & 1][q] = wfrag(I::QKV_B + ((st + 1) >> 1) * (I::NP * 128), 16 * (half * NTH + q) + lr, ((st + 1) & 1) * 4 + g);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < NTH; ++q)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) mma16(wb[st & 1][q], of[st >> 1][mt][st & 1], acc[mt][q]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int q = 0; q < NTH; ++q) {
                const int c0 = 16 * (half * NTH + q) + 4 * g;
                if (c0 < D) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) store4(p.y + tok_row(mt * 16 + lr) * D + c0, acc[mt][q]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

size_t winblock_image_bytes(int nh) {
    return nh == 1 ? (size_t)WinImg<1>::BYTES : (size_t)WinImg<2>::BYTES;
}
size_t winblock_bias_floats(int nh) { return (size_t)(3 * nh * 64 + 64 * nh); }

bool winblock_supported(int H, int W, int ws, int pool, int nh, int hd, int dim, int dout) {
    return ws == 8 && pool == 1 && dim == dout && (nh == 1 || nh == 2) && hd % 8 == 0 && hd >= 16 && hd <= 64 &&
           H % 8 == 0 && W % 8 == 0 && dim == nh * hd;
}

int launch_pack_winblock(const float* wqkv, const float* bqkv, const float* wproj, const float* bproj, int nh, int hd,
                         void* img, float* bias_p, hipStream_t s) {
    LSAM_REQUIRE(wqkv && bqkv && wproj && bproj && img && bias_p, LSAM_ERR_BAD_ARG, "pack_winblock: null argument");
    LSAM_REQUIRE((nh == 1 || nh == 2) && hd > 0 && hd <= 64, LSAM_ERR_UNSUPPORTED, "pack_winblock: heads %d x %d", nh, hd);
    if (nh == 1)
        hipLaunchKernelGGL(pack_winblock_kernel<1>, dim3(64), dim3(256), 0, s, wqkv, bqkv, wproj, bproj, hd,
                           static_cast<char*>(img), bias_p);
    else
        hipLaunchKernelGGL(pack_winblock_kernel<2>, dim3(64), dim3(256), 0, s, wqkv, bqkv, wproj, bproj, hd,
                           static_cast<char*>(img), bias_p);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

template <int NH, bool ATTN_ONLY> static int launch_wb(const WinBlockArgs& a, hipStream_t s) {
    constexpr int LDSB = WinImg<NH>::BYTES;
    constexpr int NW = ATTN_ONLY ? WINBLOCK_NW_ATTN : WINBLOCK_NW;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(winblock_attn_kernel<NH, ATTN_ONLY, NW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        attr_set = true;
    }
    const int nwin = a.N * (a.H / 8) * (a.W / 8);
    const int grid = nwin / NW < 256 ? (nwin + NW - 1) / NW : 256;
    hipLaunchKernelGGL((winblock_attn_kernel<NH, ATTN_ONLY, NW>), dim3(grid), dim3(64 * NW), LDSB, s, a);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

int launch_winblock_attn(const WinBlockArgs& a, int nh, hipStream_t s) {
    LSAM_REQUIRE(a.x && (a.y || a.a_out) && a.gamma && a.beta && a.img && a.bias_p, LSAM_ERR_BAD_ARG, "winblock: null argument");
    LSAM_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.H % 8 == 0 && a.W % 8 == 0, LSAM_ERR_UNSUPPORTED,
                 "winblock: grid %dx%d is not a multiple of the 8x8 window", a.H, a.W);
    LSAM_REQUIRE((nh == 1 || nh == 2) && a.hd % 8 == 0 && a.hd >= 16 && a.hd <= 64, LSAM_ERR_UNSUPPORTED,
                 "winblock: heads %d x %d", nh, a.hd);
    LSAM_REQUIRE(aligned16(a.x) && aligned16(a.y) && aligned16(a.a_out) && aligned16(a.img) && aligned16(a.gamma) && aligned16(a.beta) &&
                     aligned16(a.bias_p), LSAM_ERR_BAD_ARG, "winblock: pointers must be 16-byte aligned");
    ProfScope ps(s, (double)a.N * a.H * a.W * nh * a.hd * (a.a_out ? 6.0 : 8.0), PROF_WINBLOCK, a.N * a.H * a.W, nh * a.hd, 0);   // x in (fp32) + y out (fp32) or a out (bf16)
    if (a.a_out) return nh == 1 ? launch_wb<1, true>(a, s) : launch_wb<2, true>(a, s);
    return nh == 1 ? launch_wb<1, false>(a, s) : launch_wb<2, false>(a, s);
}

}  // namespace lsam
