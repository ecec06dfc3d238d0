// Attention half of a q-pool stage-entry block with 8x8 windows (Hiera-B+ block 2: dim 112 -> 224, 4 heads of 56;
// MultiScaleAttention.forward, hieradet.py:56-81, with q = do_pool(q) :69-71) as ONE kernel per window:
//
//     k | v = norm1(x) . W_kv^T + b_kv          (64 tokens of the window, all heads)
//     out   = softmax(q_pooled k^T / sqrt(hd)) v  (16 pooled queries per window, per head)
//
// Un-fused, k | v of this block are the largest tensor of the whole trunk (1 M tokens x 448 channels: 940 MB per
// 16-frame batch), written by a GEMM and read back once by the attention kernel -- 634 us of HBM traffic around 137
// GFLOP.  Here the window's 64 x dim LN1 rows are read once as MFMA operand fragments (a lane's 16-byte load IS its
// k-contiguous fragment), W_kv sits in LDS for the lifetime of the workgroup (fragment order, 128 KiB for 4 heads), and
// k, v, the scores and the probabilities never leave registers:
//
//   * K^T = W_k . X^T with the ROWS of W_k (the head's channels d) permuted at pack time so that a lane's accumulator
//     values of two row tiles are exactly its 8 k-slots of the A operand of S^T = K . Q^T (d is contracted, so any
//     order works as long as q is read in the same one -- and q is read in natural order: slot 32 s + 8 g + j).
//   * V = X . W_v^T in natural order: a lane's accumulator values of two TOKEN tiles are its 8 key slots of the A
//     operand of O^T = V^T . P^T, and the same two tiles' score registers are the matching slots of P^T.
//   * the pooled queries come from the qkv GEMM's pooling epilogue (fp32, lsam_linear_maxpool2x2), as before.
//
// One wave per window, 8 waves per workgroup (one workgroup per CU: the weight image fills the LDS), persistent.
#include "kernels.hpp"

namespace lsam {

constexpr int KVA_NH = 4;    // heads whose K / V weights share the LDS
constexpr int KVA_KS = 4;    // 32-wide k-steps over the input channels (dim <= 128)

// image: [head][part 0 = K (permuted rows), 1 = V][tile 0..3][k-step 0..3] fragments of [64 lanes][8 bf16], then
// fp32 [head][64] K bias in permuted-row order, then fp32 [head][64] V bias (zero beyond hd).
static constexpr size_t KVA_FRAGS = (size_t)KVA_NH * 2 * 4 * KVA_KS;
size_t kvattn_image_bytes() { return KVA_FRAGS * 1024 + 2 * KVA_NH * 64 * sizeof(float); }

bool kvattn_supported(int H, int W, int ws, int pool, int nh, int hd, int dim, int dout) {
    return ws == 8 && pool == 2 && H % 8 == 0 && W % 8 == 0 && nh == KVA_NH && hd % 8 == 0 && hd >= 8 && hd <= 64 &&
           dim % 8 == 0 && dim >= 8 && dim <= 32 * KVA_KS && dout == nh * hd;
}

__global__ void pack_kvattn_kernel(const float* __restrict__ wqkv, const float* __restrict__ bqkv, int dim, int hd,
                                   bf16* __restrict__ img, float* __restrict__ bias) {
    const int q = KVA_NH * hd;
    const long total = (long)KVA_FRAGS * 512;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total + 2 * KVA_NH * 64; i += (long)gridDim.x * blockDim.x) {
        if (i < total) {
            const int f = (int)(i >> 9), lane = (int)((i & 511) >> 3), j = (int)(i & 7);
            const int ks = f % KVA_KS, tile = (f / KVA_KS) % 4, part = (f / (KVA_KS * 4)) % 2, h = f / (KVA_KS * 8);
            const int g = lane >> 4, lr = lane & 15;
            const int k = 32 * ks + 8 * g + j;
            // K: permuted row 16 tile + lr holds channel d = 32 (tile / 2) + 8 (lr / 4) + 4 (tile % 2) + lr % 4
            const int d = part == 0 ? 32 * (tile >> 1) + 8 * (lr >> 2) + 4 * (tile & 1) + (lr & 3) : 16 * tile + lr;
            float v = 0.f;
            if (d < hd && k < dim) v = wqkv[(size_t)((part + 1) * q + h * hd + d) * dim + k];
            img[i] = (bf16)v;
        } else {
            const int t = (int)(i - total), part = t / (KVA_NH * 64), h = (t / 64) % KVA_NH, p = t % 64;
            const int tile = p >> 4, lr = p & 15;
            const int d = part == 0 ? 32 * (tile >> 1) + 8 * (lr >> 2) + 4 * (tile & 1) + (lr & 3) : p;
            bias[t] = d < hd ? bqkv[(part + 1) * q + h * hd + d] : 0.f;
        }
    }
}

int launch_pack_kvattn(const float* wqkv, const float* bqkv, int dim, int nh, int hd, void* img, hipStream_t s) {
    LSAM_REQUIRE(wqkv && bqkv && img, LSAM_ERR_BAD_ARG, "kvattn_pack: null argument");
    LSAM_REQUIRE(kvattn_supported(8, 8, 8, 2, nh, hd, dim, nh * hd), LSAM_ERR_UNSUPPORTED,
                 "kvattn_pack: %d heads of %d over %d channels not supported", nh, hd, dim);
    char* base = static_cast<char*>(img);
    hipLaunchKernelGGL(pack_kvattn_kernel, dim3(256), dim3(256), 0, s, wqkv, bqkv, dim, hd, reinterpret_cast<bf16*>(base),
                       reinterpret_cast<float*>(base + KVA_FRAGS * 1024));
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

__global__ __launch_bounds__(512) void kvattn_kernel(const KvAttnArgs p, const int n_win) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, lr = lane & 15;
    const int hd = p.hd, dim = p.dim;
    // ---- weight image -> LDS, once per workgroup ------------------------------------------------------------------
    {
        // (all 16 loads of a thread requested before its first LDS store: one memory round trip, not sixteen)
        const uint4* src = reinterpret_cast<const uint4*>(p.img) + threadIdx.x;
        uint4* dst = reinterpret_cast<uint4*>(lds) + threadIdx.x;
        if (threadIdx.x < 2 * KVA_NH * 64 / 4)   // the two bias tables (2 KiB) behind the image
            reinterpret_cast<uint4*>(lds + KVA_FRAGS * 1024)[threadIdx.x] =
                reinterpret_cast<const uint4*>(p.img + KVA_FRAGS * 1024)[threadIdx.x];
        constexpr int NLD = (int)(KVA_FRAGS * 64 / 512);
        uint4 tmp[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) tmp[i] = src[i * 512];
#pragma unroll
        for (int i = 0; i < NLD; ++i) dst[i * 512] = tmp[i];
    }
    __syncthreads();
    const float* bk = reinterpret_cast<const float*>(lds + KVA_FRAGS * 1024);   // LDS: a global read here is a memory
    const float* bv = bk + KVA_NH * 64;                                         // round trip in the middle of every head
    const bf16x8* wl = reinterpret_cast<const bf16x8*>(lds) + lane;
    const int wpr = p.W / 8, wpi = (p.H / 8) * wpr, Hq = p.H / 2, Wq = p.W / 2;
    const int q = KVA_NH * hd;

    for (int w = blockIdx.x * 8 + wave; w < n_win; w += gridDim.x * 8) {
        const int n = w / wpi, wy = (w % wpi) / wpr, wx = w % wpr;
        // ---- the window's LN1 rows as operand fragments: token tile t = window rows 2t, 2t+1 ---------------------
        bf16x8 xf[4][KVA_KS];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const size_t tok = ((size_t)n * p.H + wy * 8 + 2 * t + (lr >> 3)) * p.W + wx * 8 + (lr & 7);
            const bf16* row = p.xn + tok * p.ldx;
#pragma unroll
            for (int ks = 0; ks < KVA_KS; ++ks) {
                const int k0 = 32 * ks + 8 * g;
                uint4 v = *reinterpret_cast<const uint4*>(row + min(k0, dim - 8));   // clamped, branch-free
                if (k0 >= dim) v = make_uint4(0, 0, 0, 0);
                xf[t][ks] = __builtin_bit_cast(bf16x8, v);
            }
        }
        // pooled query lr of this window
        const size_t qtok = ((size_t)n * Hq + wy * 4 + (lr >> 2)) * Wq + wx * 4 + (lr & 3);
        f32x4 qa[2], qb[2];
        auto load_q = [&](int h) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int d0 = 32 * s + 8 * g;
                qa[s] = f32x4{0.f, 0.f, 0.f, 0.f};
                qb[s] = qa[s];
                if (d0 < hd) {
                    const float* qs = p.qp + qtok * p.ldqp + h * hd + d0;
                    qa[s] = load4(qs);
                    qb[s] = load4(qs + 4);
                }
            }
        };
        load_q(0);
#pragma unroll 1
        for (int h = 0; h < KVA_NH; ++h) {
            const bf16x8* wk = wl + (size_t)(h * 2 + 0) * 4 * KVA_KS * 64;
            const bf16x8* wv = wl + (size_t)(h * 2 + 1) * 4 * KVA_KS * 64;
            // ---- queries (fp32, pooled in the GEMM epilogue) -> B fragments; the next head's are requested now and
            //      converted at the end of this head's work ---------------------------------------------------------------
            bf16x8 qf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) { qf[s][j] = (bf16)qa[s][j]; qf[s][4 + j] = (bf16)qb[s][j]; }
            if (h + 1 < KVA_NH) load_q(h + 1);
            // ---- K^T = W_k(perm) . X^T + b_k: acc[rt][kt], rows = permuted channels, columns = tokens of tile kt -----
            bf16x8 kf[4][2];
            {
                f32x4 acc[4][4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const f32x4 b = load4(bk + h * 64 + 16 * rt + 4 * g);
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) acc[rt][kt] = b;
                }
                // a k-step's four weight fragments are read one k-step (16 MFMAs) ahead of their use
                bf16x8 wf[2][4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) wf[0][rt] = wk[(rt * KVA_KS + 0) * 64];
#pragma unroll
                for (int ks = 0; ks < KVA_KS; ++ks) {
                    if (ks + 1 < KVA_KS) {
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt) wf[(ks + 1) & 1][rt] = wk[(rt * KVA_KS + ks + 1) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise sinks every read to just in front of its MFMAs)
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                        for (int kt = 0; kt < 4; ++kt) mma16(wf[ks & 1][rt], xf[kt][ks], acc[rt][kt]);
                }
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            kf[kt][s][j] = (bf16)acc[2 * s][kt][j];
                            kf[kt][s][4 + j] = (bf16)acc[2 * s + 1][kt][j];
                        }
            }
            // ---- S^T = K . Q^T, softmax over the 64 keys of query lr ---------------------------------------------------
            f32x4 sc[4];
            float mx = -1.0e30f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                mma16(kf[kt][0], qf[0], sc[kt]);
                mma16(kf[kt][1], qf[1], sc[kt]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sc[kt][r] *= p.c_log2;
                    mx = fmaxf(mx, sc[kt][r]);
                }
            }
            mx = group4_max(mx);
            float l = 0.f;
            bf16x8 pf[2];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(sc[kt][r] - mx);
                    l += e;
                    pf[kt >> 1][4 * (kt & 1) + r] = (bf16)e;
                }
            l = group4_sum(l);
            // ---- V = X . W_v^T + b_v: acc[tt][dt], rows = tokens of tile tt, columns = channels of tile dt -------------
            f32x4 o[4];
            {
                f32x4 acc[4][4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const float b = bv[h * 64 + 16 * dt + lr];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[tt][dt] = f32x4{b, b, b, b};
                }
                bf16x8 wf[2][4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) wf[0][dt] = wv[(dt * KVA_KS + 0) * 64];
#pragma unroll
                for (int ks = 0; ks < KVA_KS; ++ks) {
                    if (ks + 1 < KVA_KS) {
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) wf[(ks + 1) & 1][dt] = wv[(dt * KVA_KS + ks + 1) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) mma16(xf[tt][ks], wf[ks & 1][dt], acc[tt][dt]);
                }
                // ---- O^T = V^T . P^T: key slot 8 g + j of k-step s = token 4 g + j % 4 of tile 2 s + j / 4 ---------------
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        bf16x8 vf;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            vf[j] = (bf16)acc[2 * s][dt][j];
                            vf[4 + j] = (bf16)acc[2 * s + 1][dt][j];
                        }
                        mma16(vf, pf[s], o[dt]);
                    }
                }
            }
            // ---- normalise and store: lane holds O[query lr][channel 16 dt + 4 g .. +3] ------------------------------
            const float inv = 1.0f / l;
            bf16* orow = p.out + qtok * p.ldo + h * hd;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int d0 = 16 * dt + 4 * g;
                if (d0 < hd) store4(orow + d0, o[dt] * inv);
            }
        }
    }
    (void)q;
}

int launch_kvattn(const KvAttnArgs& a, hipStream_t s) {
    LSAM_REQUIRE(a.xn && a.qp && a.img && a.out && a.N > 0, LSAM_ERR_BAD_ARG, "kvattn: null/empty argument");
    LSAM_REQUIRE(kvattn_supported(a.H, a.W, 8, 2, KVA_NH, a.hd, a.dim, KVA_NH * a.hd), LSAM_ERR_UNSUPPORTED,
                 "kvattn: grid %dx%d, head_dim %d, %d channels not supported", a.H, a.W, a.hd, a.dim);
    LSAM_REQUIRE(a.ldx >= a.dim && a.ldx % 8 == 0 && a.ldqp >= KVA_NH * a.hd && a.ldqp % 4 == 0 && a.ldo >= KVA_NH * a.hd &&
                     a.ldo % 4 == 0 && aligned16(a.xn) && aligned16(a.qp) && aligned16(a.img) && (reinterpret_cast<uintptr_t>(a.out) & 7) == 0,
                 LSAM_ERR_BAD_ARG, "kvattn: leading dimensions / alignment");
    LSAM_REQUIRE((double)a.N * a.H * a.W < 2.0e9, LSAM_ERR_UNSUPPORTED, "kvattn: too many tokens");
    const int n_win = a.N * (a.H / 8) * (a.W / 8);
    const size_t lds = KVA_FRAGS * 1024 + 2 * KVA_NH * 64 * sizeof(float);
    static thread_local bool attr_set = false;
    if (!attr_set) {
        LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kvattn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    // algorithmic FLOPs: the k | v projection of every token + the 16 x 64 attention of every window and head
    const double M = (double)a.N * a.H * a.W;
    ProfScope ps(s, 2.0 * M * a.dim * 2 * KVA_NH * a.hd + 4.0 * n_win * 16.0 * 64.0 * KVA_NH * a.hd, PROF_KVATTN, (int)M,
                 2 * KVA_NH * a.hd, a.dim);
    hipLaunchKernelGGL(kvattn_kernel, dim3(min(256, cdiv(n_win, 8))), dim3(512), lds, s, a, n_win);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

}  // namespace lsam
