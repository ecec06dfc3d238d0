// Whole-trunk plan: Hiera.forward / get_embedding (hieradet.py:284-334) with LoRA_qkv on the last
// blocks (lora_qkv.py:102-109,162) as ONE host call that enqueues ~8 kernels per block, and the
// LoRA-only backward as a second call.  The plan is host-side bookkeeping only: packed weights
// live in a caller-provided arena, activations in a caller-provided workspace.
//
// HBM layout: activations are token-major NHWC ([N*H*W][C]).  The residual stream and the stage
// outputs are fp32; every GEMM operand is T (bf16 or fp32).  Blocks of a stage update the stage's
// output buffer in place (x += proj(attn); x += mlp), so a non-differentiated block allocates no
// residual storage.  The LoRA blocks write their residual states to saved buffers instead, because
// the backward needs the LN inputs.
//
// LoRA is folded into the frozen-qkv MFMA main loop by K-augmentation: the LN1 output buffer
// carries 32 extra columns u = [xn.A_q^T (16) | xn.A_v^T (16)] and the packed qkv weight carries
// [B_q rows 0..d | B_v rows 2d..3d] in the matching 32 extra K columns, so
// qkv = [xn|u].[W|B]^T is the reference's two slice-adds (lora_qkv.py:104-107) with no extra pass
// over qkv.  The backward re-uses the same trick for dxn = dqkv.W + t_q.A_q + t_v.A_v.
#include <algorithm>
#include <new>
#include <vector>

#include "kernels.hpp"

namespace lsam {

constexpr int AUG = 32;       // extra K columns: [q part 16 | v part 16]
constexpr int PATCH_K = 160;  // 3*7*7 = 147 padded to a multiple of 16
#ifndef QPOOL_ANY_WIDTH
#define QPOOL_ANY_WIDTH 1     // pooled queries from the GEMM epilogue at any width the guarded tiles take (0: multiples of 112 only)
#endif
static inline bool kv_width_ok(long q) { return QPOOL_ANY_WIDTH ? (2 * q) % 8 == 0 && 2 * q >= 144 : (2 * q) % 224 == 0; }
#ifndef SPLIT_PROJ
#define SPLIT_PROJ 1          // stage-1 blocks: proj + residual in the MLP kernel instead of the attention kernel (0: as before)
#endif
#ifndef QKVATTN4
#define QKVATTN4 1            // 4x4-window blocks: qkv (or k | v) projection inside the attention kernel (0: projection GEMM + attention)
#endif
#ifndef STREAMLIN
#define STREAMLIN 1           // K <= 224 qkv projections on the LDS-resident-weight streaming kernel (0: the tiled GEMM)
#endif
#ifndef MERGE_POOLED
#define MERGE_POOLED 1        // a stage-entry block's shortcut and pooled queries from ONE pooled GEMM (0: two launches)
#endif

struct BlockGeom {
    int dim, dout, heads, ws, pool, gin, gout, stage;
    bool lora, need_dx;
    int kq;  // K of the qkv GEMM (dim or dim + AUG)
    bool fused_attn;  // bf16 only: LN1 + qkv + window attention + proj + residual in one kernel (winblock.hip)
    bool fused_mlp;   // bf16 only: LN2 + fc1 + GELU + fc2 + residual in one kernel (mlpblock.hip)
    bool fused_tail;  // bf16 only: proj + residual + LN2 + MLP + residual (+ next block's LN1) in one kernel (tailblock.hip)
    bool split_proj;  // fused_attn + fused_mlp: the attention kernel stops at its output rows, proj + residual open the MLP kernel
    bool fused_qkv4;  // bf16 only: qkv (k | v for a q-pool block) projection + 4x4-window attention in one kernel (qkvattn.hip)
    bool fused_kv;    // bf16 only: k | v projection + 8x8-window attention of a q-pool stage-entry block in one kernel (kvattn.hip)
};
struct BlockOff {
    size_t ln1w, ln1b, wqkv, bqkv, wproj, bproj, ln2w, ln2b, w1, b1, w2, b2, wsc, bsc;
    size_t wqkvT, wprojT, w1T, w2T, acat, bT;
    size_t bqkvT;  // qkv bias in T (pad-token k/v source of the LDS-DMA attention kernel)
    size_t wimg, wbias;  // fused window-block weight image + padded biases (winblock.hip), if g.fused_attn
    size_t mimg, mbias;  // fused MLP weight image + fc2 bias (mlpblock.hip), if g.fused_mlp
    size_t mpimg, mpbias; // the attention projection as a chunk of that kernel's weight stream + its bias, if g.split_proj
    size_t timg, tvec;   // fused tail weight stream + packed vectors (tailblock.hip), if g.fused_tail
    size_t kvimg;        // k | v weight image of the fused projection + attention kernel (kvattn.hip), if g.fused_kv
    size_t qaimg;        // weight image of the fused projection + 4x4-window attention kernel (qkvattn.hip), if g.fused_qkv4
    size_t simg;         // small-K qkv (or k | v) projection with LDS-resident weights (lateral.hip: streamlin), rows simg_n0 .. + simg_n
    int simg_n0, simg_n; // of the qkv weight; simg_n == 0: not built
};

}  // namespace lsam

using namespace lsam;

struct lsam_hiera {
    lsam_hiera_config_t cfg;
    std::vector<BlockGeom> blk;
    std::vector<BlockOff> off;
    int stage_ch[4], stage_grid[4];
    int grid0;
    size_t wpatch, bpatch, pos;
    size_t wpatch_img;   // bf16: fragment-order weight image of the direct PatchEmbed kernel (patchembed.hip)
    bool direct_patch;
    size_t arena_bytes;
    size_t esz;
    int first_lora;
};

namespace lsam {

struct Bump {
    size_t cur = 0;
    size_t take(size_t bytes) {
        const size_t o = cur;
        cur = align_up(cur + bytes, 256);
        return o;
    }
};

struct FwdWs {
    size_t im2col, xn, qkv, scpre, a, hbuf;
    // per LoRA block (index g = block - first_lora)
    struct Saved { size_t st1, xn, qkv, lse, y, st2, pre, xout; } sv[8];
    // backward scratch
    size_t gT, dh, dyn, dy, dyT, da, dqkv, dxn, dxprev, partial;
    size_t total;
};

static FwdWs plan_ws(const lsam_hiera* h, int N, bool with_grad) {
    FwdWs w{};
    Bump b;
    const size_t e = h->esz;
    const size_t M0 = (size_t)N * h->grid0 * h->grid0;
    w.im2col = b.take(h->direct_patch ? 0 : M0 * PATCH_K * e);
    size_t xn = 0, qkv = 0, sc = 0, a = 0, hb = 0;
    for (const BlockGeom& g : h->blk) {
        const size_t Mi = (size_t)N * g.gin * g.gin, Mo = (size_t)N * g.gout * g.gout;
        xn = std::max(xn, std::max(Mi * g.kq, Mo * g.dout) * e);
        qkv = std::max(qkv, Mi * 3 * g.dout * e);
        if (g.dim != g.dout && g.pool == 2) sc = std::max(sc, Mi * g.dout * sizeof(float));
        a = std::max(a, Mo * g.dout * e);
        hb = std::max(hb, Mo * 4 * g.dout * e);
    }
    w.xn = b.take(xn);
    w.qkv = b.take(qkv);
    w.scpre = b.take(sc);
    w.a = b.take(a);
    w.hbuf = b.take(hb);
    if (with_grad && h->cfg.n_lora_blocks > 0) {
        // per LoRA block: what its backward needs; the first one may be a stage-entry block (dim != dim_out, q-pool)
        size_t moq = 0, miq = 0, mid = 0, part = 0;
        for (int i = 0; i < h->cfg.n_lora_blocks; ++i) {
            const BlockGeom& g = h->blk[h->first_lora + i];
            const size_t Mi = (size_t)N * g.gin * g.gin, Mo = (size_t)N * g.gout * g.gout, d = g.dim, q = g.dout;
            auto& s = w.sv[i];
            s.st1 = b.take(Mi * 2 * sizeof(float));
            s.xn = b.take(Mi * (d + AUG) * e);
            s.qkv = b.take(Mi * 3 * q * e);
            s.lse = b.take(Mo * g.heads * sizeof(float));
            s.y = b.take(Mo * q * sizeof(float));
            s.st2 = b.take(Mo * 2 * sizeof(float));
            s.pre = b.take(Mo * 4 * q * e);
            s.xout = b.take(Mo * q * sizeof(float));
            moq = std::max(moq, Mo * q);
            miq = std::max(miq, Mi * (3 * q + AUG));
            mid = std::max(mid, Mi * d);
            part = std::max(part, outer_reduce_ws((int)Mi, (int)d));
        }
        w.gT = b.take(moq * e);
        w.dh = b.take(moq * 4 * e);
        w.dyn = b.take(moq * sizeof(float));
        w.dy = b.take(moq * sizeof(float));
        w.dyT = b.take(moq * e);
        w.da = b.take(moq * e);
        w.dqkv = b.take(miq * e);
        w.dxn = b.take(mid * sizeof(float));
        w.dxprev = b.take(mid * sizeof(float));
        w.partial = b.take(4 * part);
    }
    w.total = b.cur;
    return w;
}

// ---------------------------------------------------------------------------------------------
// LoRA packing: rewrite the LoRA-dependent slices of the packed operands from the fp32 masters.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_lora_kernel(const float* __restrict__ aq, const float* __restrict__ bq,
                                 const float* __restrict__ av, const float* __restrict__ bv, int d, int q, int r,
                                 T* __restrict__ wqkv, T* __restrict__ acat, T* __restrict__ bT,
                                 T* __restrict__ wqkvT) {
    // d = in_features of the wrapped qkv (LoRA_qkv.d_model, lora_qkv.py:99), q = dim_out: the adds go to the first and
    // the last d of the 3q output columns (lora_qkv.py:106-107), which are the q and v thirds only when d == q
    const int kq = d + AUG, n3 = 3 * q;
    const long n_w = (long)n3 * AUG, n_a = (long)AUG * d, n_b = (long)AUG * n3, n_t = wqkvT ? (long)d * AUG : 0;
    const long total = n_w + n_a + n_b + n_t;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (i < n_w) {  // wqkv[n][d + j]
            const int n = (int)(i / AUG), j = (int)(i % AUG);
            float v = 0.f;
            if (j < 16) { if (j < r && n < d) v = bq[(size_t)n * r + j]; }
            else { const int k = j - 16; if (k < r && n >= n3 - d) v = bv[(size_t)(n - (n3 - d)) * r + k]; }
            wqkv[(size_t)n * kq + d + j] = from_f<T>(v);
        } else if (i < n_w + n_a) {  // acat[j][c]
            const long t = i - n_w;
            const int j = (int)(t / d), c = (int)(t % d);
            float v = 0.f;
            if (j < 16) { if (j < r) v = aq[(size_t)j * d + c]; }
            else if (j - 16 < r) v = av[(size_t)(j - 16) * d + c];
            acat[t] = from_f<T>(v);
        } else if (i < n_w + n_a + n_b) {  // bT[j][n]
            const long t = i - n_w - n_a;
            const int j = (int)(t / n3), n = (int)(t % n3);
            float v = 0.f;
            if (j < 16) { if (j < r && n < d) v = bq[(size_t)n * r + j]; }
            else { const int k = j - 16; if (k < r && n >= n3 - d) v = bv[(size_t)(n - (n3 - d)) * r + k]; }
            bT[t] = from_f<T>(v);
        } else {  // wqkvT[c][3d + j] = A[j][c]
            const long t = i - n_w - n_a - n_b;
            const int c = (int)(t / AUG), j = (int)(t % AUG);
            float v = 0.f;
            if (j < 16) { if (j < r) v = aq[(size_t)j * d + c]; }
            else if (j - 16 < r) v = av[(size_t)(j - 16) * d + c];
            wqkvT[(size_t)c * (n3 + AUG) + n3 + j] = from_f<T>(v);
        }
    }
}

template <typename T> static T* at(void* base, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(base) + off); }

template <typename T>
static int pack_t(lsam_hiera* h, const lsam_block_params_t* bp, const float* patch_w, const float* patch_b,
                  const float* pos, void* arena, hipStream_t s) {
    const int e0 = h->cfg.embed_dim;
    LSAM_TRY(launch_pack_matrix<T>(patch_w, 147, 1, e0, 147, at<T>(arena, h->wpatch), PATCH_K, s));
    if (h->direct_patch) LSAM_TRY(launch_pack_patch_embed(patch_w, e0, at<char>(arena, h->wpatch_img), s));
    LSAM_TRY(launch_pack_matrix<float>(patch_b, 0, 1, 1, e0, at<float>(arena, h->bpatch), e0, s));
    LSAM_TRY(launch_pack_matrix<float>(pos, 0, 1, 1, h->grid0 * h->grid0 * e0, at<float>(arena, h->pos),
                                       h->grid0 * h->grid0 * e0, s));
    for (size_t i = 0; i < h->blk.size(); ++i) {
        const BlockGeom& g = h->blk[i];
        const BlockOff& o = h->off[i];
        const lsam_block_params_t& p = bp[i];
        const int d = g.dim, q = g.dout;
        LSAM_REQUIRE(p.norm1_w && p.norm1_b && p.qkv_w && p.qkv_b && p.attn_proj_w && p.attn_proj_b && p.norm2_w &&
                         p.norm2_b && p.fc1_w && p.fc1_b && p.fc2_w && p.fc2_b && (d == q || (p.proj_w && p.proj_b)),
                     LSAM_ERR_BAD_ARG, "hiera_pack: block %zu has a null parameter", i);
        auto vec = [&](const float* src, size_t off_, int n) {
            return launch_pack_matrix<float>(src, 0, 1, 1, n, at<float>(arena, off_), n, s);
        };
        LSAM_TRY(vec(p.norm1_w, o.ln1w, d));
        LSAM_TRY(vec(p.norm1_b, o.ln1b, d));
        LSAM_TRY(launch_pack_matrix<T>(p.qkv_w, d, 1, 3 * q, d, at<T>(arena, o.wqkv), g.kq, s));
        LSAM_TRY(vec(p.qkv_b, o.bqkv, 3 * q));
        LSAM_TRY(launch_pack_matrix<T>(p.qkv_b, 0, 1, 1, 3 * q, at<T>(arena, o.bqkvT), 3 * q, s));
        LSAM_TRY(launch_pack_matrix<T>(p.attn_proj_w, q, 1, q, q, at<T>(arena, o.wproj), q, s));
        LSAM_TRY(vec(p.attn_proj_b, o.bproj, q));
        LSAM_TRY(vec(p.norm2_w, o.ln2w, q));
        LSAM_TRY(vec(p.norm2_b, o.ln2b, q));
        LSAM_TRY(launch_pack_matrix<T>(p.fc1_w, q, 1, 4 * q, q, at<T>(arena, o.w1), q, s));
        LSAM_TRY(vec(p.fc1_b, o.b1, 4 * q));
        LSAM_TRY(launch_pack_matrix<T>(p.fc2_w, 4 * q, 1, q, 4 * q, at<T>(arena, o.w2), 4 * q, s));
        LSAM_TRY(vec(p.fc2_b, o.b2, q));
        if (d != q) {
            LSAM_TRY(launch_pack_matrix<T>(p.proj_w, d, 1, q, d, at<T>(arena, o.wsc), d, s));
            LSAM_TRY(vec(p.proj_b, o.bsc, q));
            if (!g.lora) {   // [W_proj ; W_q] and [b_proj ; b_q]: one pooled GEMM gives the shortcut and the pooled queries
                LSAM_TRY(launch_pack_matrix<T>(p.qkv_w, d, 1, q, d, at<T>(arena, o.wsc) + (size_t)q * d, d, s));
                LSAM_TRY(vec(p.qkv_b, o.bsc + (size_t)q * 4, q));
            }
        }
        if (g.fused_attn)
            LSAM_TRY(launch_pack_winblock(p.qkv_w, p.qkv_b, p.attn_proj_w, p.attn_proj_b, g.heads, q / g.heads,
                                          at<char>(arena, o.wimg), at<float>(arena, o.wbias), s));
        if (g.fused_mlp)
            LSAM_TRY(launch_pack_mlpblock(p.fc1_w, p.fc1_b, p.fc2_w, p.fc2_b, q, at<char>(arena, o.mimg),
                                          at<float>(arena, o.mbias), s));
        if (g.split_proj)
            LSAM_TRY(launch_pack_mlpblock_proj(p.attn_proj_w, p.attn_proj_b, q, at<char>(arena, o.mpimg), at<float>(arena, o.mpbias), s));
        if (g.fused_tail)
            LSAM_TRY(launch_pack_tailblock(p.attn_proj_w, p.attn_proj_b, p.norm2_w, p.norm2_b, p.fc1_w, p.fc1_b, p.fc2_w,
                                           p.fc2_b, q, at<char>(arena, o.timg), at<float>(arena, o.tvec), s));
        if (g.fused_kv)
            LSAM_TRY(launch_pack_kvattn(p.qkv_w, p.qkv_b, d, g.heads, q / g.heads, at<char>(arena, o.kvimg), s));
        if (g.fused_qkv4)
            LSAM_TRY(launch_pack_qkvattn4(p.qkv_w, p.qkv_b, d, g.heads, q / g.heads, g.pool == 2, at<char>(arena, o.qaimg), s));
        if (o.simg_n > 0)
            LSAM_TRY(launch_pack_streamlin(static_cast<const float*>(p.qkv_w) + (size_t)o.simg_n0 * d, d, o.simg_n, at<char>(arena, o.simg), s));
        if (g.lora) {  // transposed copies for dX = dY . W   (frozen: no dW GEMMs exist)
            LSAM_TRY(launch_pack_matrix<T>(p.fc2_w, 1, 4 * q, 4 * q, q, at<T>(arena, o.w2T), q, s));
            LSAM_TRY(launch_pack_matrix<T>(p.fc1_w, 1, q, q, 4 * q, at<T>(arena, o.w1T), 4 * q, s));
            LSAM_TRY(launch_pack_matrix<T>(p.attn_proj_w, 1, q, q, q, at<T>(arena, o.wprojT), q, s));
            if (g.need_dx)
                LSAM_TRY(launch_pack_matrix<T>(p.qkv_w, 1, d, d, 3 * q, at<T>(arena, o.wqkvT), 3 * q + AUG, s));
        }
    }
    return LSAM_OK;
}

template <typename T>
static int pack_lora_t(lsam_hiera* h, const lsam_lora_params_t* lp, void* arena, hipStream_t s) {
    const int r = h->cfg.lora_rank;
    for (int i = 0; i < h->cfg.n_lora_blocks; ++i) {
        const int bi = h->first_lora + i;
        const BlockGeom& g = h->blk[bi];
        const BlockOff& o = h->off[bi];
        LSAM_REQUIRE(lp[i].a_q && lp[i].b_q && lp[i].a_v && lp[i].b_v, LSAM_ERR_BAD_ARG, "pack_lora: null LoRA factor");
        hipLaunchKernelGGL(pack_lora_kernel<T>, dim3(256), dim3(256), 0, s, lp[i].a_q, lp[i].b_q, lp[i].a_v,
                           lp[i].b_v, g.dim, g.dout, r, at<T>(arena, o.wqkv), at<T>(arena, o.acat), at<T>(arena, o.bT),
                           g.need_dx ? at<T>(arena, o.wqkvT) : (T*)nullptr);
        LSAM_LAUNCH_CHECK();
    }
    return LSAM_OK;
}

template <typename T>
static int linear(const T* A, int lda, const T* W, int ldw, const float* bias, const float* res, int ldres,
                  int res_mod, const T* aux, int ldaux, void* out, int ldo, T* out2, int ldo2, int flags, int M,
                  int N, int K, hipStream_t s) {
    GemmArgs a{A, lda, W, ldw, bias, res, ldres, res_mod, aux, ldaux, out, ldo, out2, ldo2, flags, M, N, K};
    return launch_gemm<T>(a, s);
}

template <typename T>
static int forward_t(lsam_hiera* h, const float* img, int N, float* const* stage_out, void* arena, void* ws,
                     const FwdWs& w, bool with_grad, hipStream_t s) {
    const int e0 = h->cfg.embed_dim, g0 = h->grid0;
    const int M0 = N * g0 * g0;
    // PatchEmbed (backbones/utils.py:85-93) + pos embed (hieradet.py:307-312): bf16 as one direct convolution
    // kernel, fp32 as im2col + fused GEMM
    if (h->direct_patch) {
        LSAM_TRY(launch_patch_embed(img, at<char>(arena, h->wpatch_img), at<float>(arena, h->bpatch), at<float>(arena, h->pos),
                                    stage_out[0], N, h->cfg.img_size, e0, s));
    } else {
        LSAM_TRY(launch_im2col_patch<T>(img, at<T>(ws, w.im2col), PATCH_K, N, h->cfg.img_size, s));
        LSAM_TRY(linear<T>(at<T>(ws, w.im2col), PATCH_K, at<T>(arena, h->wpatch), PATCH_K, at<float>(arena, h->bpatch),
                           at<float>(arena, h->pos), e0, g0 * g0, nullptr, 0, stage_out[0], e0, nullptr, 0,
                           LSAM_EPI_OUT_F32, M0, e0, PATCH_K, s));
    }
    float* x = stage_out[0];
    bool xn_ready = false;   // the previous block's fused tail already wrote this block's LN1 output
    for (size_t i = 0; i < h->blk.size(); ++i) {
        const BlockGeom& g = h->blk[i];
        const BlockOff& o = h->off[i];
        const int d = g.dim, q = g.dout;
        const int Mi = N * g.gin * g.gin, Mo = N * g.gout * g.gout;
        const bool grad = with_grad && g.lora;
        const int gi = (int)i - h->first_lora;
        const bool last = i + 1 == h->blk.size();
        float* y;
        if (g.fused_attn) {   // LN1 -> qkv -> window attention -> proj -> + x, in place, one kernel
            WinBlockArgs wa{x, x, at<float>(arena, o.ln1w), at<float>(arena, o.ln1b), at<char>(arena, o.wimg),
                            at<float>(arena, o.wbias), N, g.gin, g.gin, q / g.heads, 1e-6f,
                            (1.0f / sqrtf((float)(q / g.heads))) * 1.4426950408889634f};
            if constexpr (sizeof(T) == 2) {
                if (g.split_proj) wa.a_out = at<T>(ws, w.a);   // attention rows only; proj + residual open the MLP kernel below
            }
            LSAM_TRY(launch_winblock_attn(wa, g.heads, s));
            y = x;
        } else {
        // ---- norm1 (+ LoRA u columns) -----------------------------------------------------------
        T* xn = grad ? at<T>(ws, w.sv[gi].xn) : at<T>(ws, w.xn);
        if (!xn_ready)
            LSAM_TRY(launch_ln_fwd<T>(x, d, at<float>(arena, o.ln1w), at<float>(arena, o.ln1b), xn, g.kq,
                                      (grad && g.need_dx) ? at<float>(ws, w.sv[gi].st1) : nullptr, Mi, d, 1e-6f, s));
        xn_ready = false;
        if (g.lora)
            LSAM_TRY(linear<T>(xn, g.kq, at<T>(arena, o.acat), d, nullptr, nullptr, 0, 0, nullptr, 0, xn + d, g.kq,
                               nullptr, 0, 0, Mi, AUG, d, s));
        // ---- residual target ----------------------------------------------------------------------
        if (grad) y = at<float>(ws, w.sv[gi].y);
        else if (d != q) y = stage_out[g.stage];
        else y = x;
        const float* res = x;
        // q-pool blocks: the q third of the projection goes through the GEMM's pooling epilogue (only the pooled
        // queries, a quarter of the rows, ever exist: do_pool(q), hieradet.py:69-71) and only k | v are written at
        // full resolution.  The pooled shortcut leaves its un-pooled scratch free for them.
        const bool qpooled = g.pool == 2 && d != q && sizeof(T) == 2 && !g.lora && !grad && (q / g.heads) % 8 == 0 &&
                             gemm_pool_supported(g.gin, g.gin, q) && kv_width_ok(q) && Mi >= 1024;
        float* qp = qpooled ? at<float>(ws, w.scpre) : nullptr;
        bool q_done = false;
        if (d != q) {  // shortcut = pool(proj(norm1(x)))  (hieradet.py:139-140)
            if (g.pool == 2 && sizeof(T) == 2 && gemm_pool_supported(g.gin, g.gin, q)) {
                // the 2x2 max-pool is taken in the GEMM's epilogue: the full-resolution fp32 product never exists.  With
                // pooled queries the same launch produces them too (N = 2 q over [W_proj ; W_q], columns >= q to qp): both
                // are pooled products of the same LN1 rows, which are then read once
                const bool both = qpooled && MERGE_POOLED && g.kq == d && q % 224 == 0;
                GemmArgs ga{xn, g.kq, at<T>(arena, o.wsc), d, at<float>(arena, o.bsc), nullptr, 0, 0, nullptr, 0, y, q,
                            both ? static_cast<void*>(qp) : nullptr, both ? q : 0, LSAM_EPI_OUT_F32, Mi, both ? 2 * q : q, d};
                ga.pool_h = g.gin;
                ga.pool_w = g.gin;
                ga.split_n = both ? q : 0;
                LSAM_TRY(launch_gemm<T>(ga, s));
                q_done = both;
            } else {
                float* sc = g.pool == 2 ? at<float>(ws, w.scpre) : y;
                LSAM_TRY(linear<T>(xn, g.kq, at<T>(arena, o.wsc), d, at<float>(arena, o.bsc), nullptr, 0, 0, nullptr, 0, sc,
                                   q, nullptr, 0, LSAM_EPI_OUT_F32, Mi, q, d, s));
                if (g.pool == 2) LSAM_TRY(launch_maxpool2x2(sc, y, N, g.gin, g.gin, q, s));
            }
            res = y;
        }
        // ---- qkv (+LoRA through the augmented K columns) ------------------------------------------
        T* qkv = grad ? at<T>(ws, w.sv[gi].qkv) : at<T>(ws, w.qkv);
        // 4x4 windows: a window is one MFMA tile and a wave projects, scores and mixes it in registers (no qkv in memory)
        const bool use_qa4 = g.fused_qkv4 && (g.pool == 1 || qpooled);
        if (qpooled) {
            if (!q_done) {
                GemmArgs gq{xn, g.kq, at<T>(arena, o.wqkv), g.kq, at<float>(arena, o.bqkv), nullptr, 0, 0, nullptr, 0, qp, q,
                            nullptr, 0, LSAM_EPI_OUT_F32, Mi, q, g.kq};
                gq.pool_h = g.gin;
                gq.pool_w = g.gin;
                LSAM_TRY(launch_gemm<T>(gq, s));
            }
            if (!g.fused_kv && !use_qa4) {
                if (o.simg_n == 2 * q && o.simg_n0 == q)
                    LSAM_TRY(launch_streamlin(xn, g.kq, at<char>(arena, o.simg), at<float>(arena, o.bqkv) + q, qkv + q, 3 * q, Mi, d, 2 * q, s));
                else
                    LSAM_TRY(linear<T>(xn, g.kq, at<T>(arena, o.wqkv) + (size_t)q * g.kq, g.kq, at<float>(arena, o.bqkv) + q, nullptr, 0,
                                       0, nullptr, 0, qkv + q, 3 * q, nullptr, 0, 0, Mi, 2 * q, g.kq, s));
            }
        } else if (use_qa4) {
        } else if (o.simg_n == 3 * q && o.simg_n0 == 0) {
            // K <= 224: all epilogue in the tiled GEMM; here the weights stay in LDS and the waves stream rows independently
            LSAM_TRY(launch_streamlin(xn, g.kq, at<char>(arena, o.simg), at<float>(arena, o.bqkv), qkv, 3 * q, Mi, d, 3 * q, s));
        } else {
            LSAM_TRY(linear<T>(xn, g.kq, at<T>(arena, o.wqkv), g.kq, at<float>(arena, o.bqkv), nullptr, 0, 0, nullptr, 0,
                               qkv, 3 * q, nullptr, 0, 0, Mi, 3 * q, g.kq, s));
        }
        // ---- attention ------------------------------------------------------------------------------
        T* a = at<T>(ws, w.a);
        if (use_qa4) {
            if constexpr (sizeof(T) == 2) {
                QkvAttn4Args qa{xn, g.kq, qp, q, at<char>(arena, o.qaimg), a, q, N, g.gin, g.gin, d, g.heads, q / g.heads,
                                (1.0f / sqrtf((float)(q / g.heads))) * 1.4426950408889634f};
                LSAM_TRY(launch_qkvattn4(qa, s));
            }
        } else if (qpooled && g.fused_kv) {
            // k | v of the window's 64 tokens are projected inside the attention kernel: they never cross HBM
            if constexpr (sizeof(T) == 2) {
                KvAttnArgs ka{xn, g.kq, qp, q, at<char>(arena, o.kvimg), a, q, N, g.gin, g.gin, d, q / g.heads,
                              (1.0f / sqrtf((float)(q / g.heads))) * 1.4426950408889634f};
                LSAM_TRY(launch_kvattn(ka, s));
            }
        } else {
        AttnArgs aa{qkv, 3 * q, at<float>(arena, o.bqkv), a, q, grad ? at<float>(ws, w.sv[gi].lse) : nullptr,
                    N, g.gin, g.gin, g.ws, g.pool, g.heads, q / g.heads, at<T>(arena, o.bqkvT)};
        aa.qpool = qp;
        aa.ldqp = q;
        LSAM_TRY(launch_attn_fwd<T>(aa, s));
        }
        if (g.fused_tail) {   // proj + residual + norm2 + MLP + residual (+ the next block's norm1), one kernel
            const bool next_ln = !last && !h->blk[i + 1].fused_attn && !h->blk[i + 1].lora && h->blk[i + 1].dim == q;
            TailBlockArgs ta{a, q, res, y, at<char>(arena, o.timg), at<float>(arena, o.tvec),
                             next_ln ? static_cast<void*>(at<T>(ws, w.xn)) : nullptr, h->blk[next_ln ? i + 1 : i].kq,
                             next_ln ? at<float>(arena, h->off[i + 1].ln1w) : nullptr,
                             next_ln ? at<float>(arena, h->off[i + 1].ln1b) : nullptr, Mo, q, tailblock_chunks(q), 1e-6f};
            LSAM_TRY(launch_tailblock(ta, s));
            xn_ready = next_ln;
            x = y;
            if ((last || h->blk[i + 1].stage != g.stage) && x != stage_out[g.stage]) {
                LSAM_HIP(hipMemcpyAsync(stage_out[g.stage], x, (size_t)Mo * q * sizeof(float), hipMemcpyDeviceToDevice, s));
                x = stage_out[g.stage];
            }
            continue;
        }
        // ---- proj + residual  (hieradet.py:79,163) -------------------------------------------------
        LSAM_TRY(linear<T>(a, q, at<T>(arena, o.wproj), q, at<float>(arena, o.bproj), res, q, 0, nullptr, 0, y, q,
                           nullptr, 0, LSAM_EPI_OUT_F32, Mo, q, q, s));
        }
        // ---- norm2 + MLP + residual  (hieradet.py:165) -----------------------------------------------
        if (g.fused_mlp) {   // one kernel, in place on the residual stream
            MlpBlockArgs ma{y, y, at<float>(arena, o.ln2w), at<float>(arena, o.ln2b), at<char>(arena, o.mimg),
                            at<float>(arena, o.mbias), Mo, q, mlpblock_chunks(q), 1e-6f};
            // the next block's norm1 from the same registers when that block starts with a plain LayerNorm
            const bool next_ln = !last && !h->blk[i + 1].fused_attn && !h->blk[i + 1].lora && h->blk[i + 1].dim == q;
            if (g.split_proj) {
                ma.a = at<T>(ws, w.a);
                ma.lda = q;
                ma.pimg = at<char>(arena, o.mpimg);
                ma.pbias = at<float>(arena, o.mpbias);
            }
            if (next_ln) {
                ma.xn_next = at<T>(ws, w.xn);
                ma.ldn = h->blk[i + 1].kq;
                ma.gamma_n = at<float>(arena, h->off[i + 1].ln1w);
                ma.beta_n = at<float>(arena, h->off[i + 1].ln1b);
            }
            LSAM_TRY(launch_mlpblock(ma, s));
            xn_ready = next_ln;
            x = y;
        } else {
        T* yn = at<T>(ws, w.xn);
        LSAM_TRY(launch_ln_fwd<T>(y, q, at<float>(arena, o.ln2w), at<float>(arena, o.ln2b), yn, q,
                                  grad ? at<float>(ws, w.sv[gi].st2) : nullptr, Mo, q, 1e-6f, s));
        T* hb = at<T>(ws, w.hbuf);
        LSAM_TRY(linear<T>(yn, q, at<T>(arena, o.w1), q, at<float>(arena, o.b1), nullptr, 0, 0, nullptr, 0, hb, 4 * q,
                           grad ? at<T>(ws, w.sv[gi].pre) : nullptr, 4 * q,
                           LSAM_EPI_GELU | (grad ? LSAM_EPI_SAVE_PRE : 0), Mo, 4 * q, q, s));
        float* out;
        if (grad) out = last ? stage_out[g.stage] : at<float>(ws, w.sv[gi].xout);
        else out = y;
        LSAM_TRY(linear<T>(hb, 4 * q, at<T>(arena, o.w2), 4 * q, at<float>(arena, o.b2), y, q, 0, nullptr, 0, out, q,
                           nullptr, 0, LSAM_EPI_OUT_F32, Mo, q, 4 * q, s));
        x = out;
        }
        // a stage's last block must leave its result in the caller's stage buffer
        if ((last || h->blk[i + 1].stage != g.stage) && x != stage_out[g.stage]) {
            LSAM_HIP(hipMemcpyAsync(stage_out[g.stage], x, (size_t)Mo * q * sizeof(float), hipMemcpyDeviceToDevice, s));
            x = stage_out[g.stage];
        }
    }
    return LSAM_OK;
}

template <typename T>
static int backward_t(lsam_hiera* h, const float* grad_out, int N, float* const* lora_grads, void* arena, void* ws,
                      const FwdWs& w, int gi_hi, int gi_lo, hipStream_t s) {
    const int r = h->cfg.lora_rank;
    // the last LoRA block differentiates the caller's gradient; every other block continues from the
    // dx its successor left in the workspace (also across separate lsam_hiera_backward_block calls)
    const float* dout = gi_hi == h->cfg.n_lora_blocks - 1 ? grad_out : at<float>(ws, w.dxprev);
    for (int gi = gi_hi; gi >= gi_lo; --gi) {
        const int bi = h->first_lora + gi;
        const BlockGeom& g = h->blk[bi];
        const BlockOff& o = h->off[bi];
        const auto& sv = w.sv[gi];
        // d = block input width (= LoRA width), q = output width; Mi / Mo input / output tokens.  They differ only for a
        // stage-entry block (dim != dim_out, q-pool), which can only be the FIRST LoRA block: its shortcut
        // pool(proj(norm1(x))) (hieradet.py:139-140) has no trainable parameter and x carries no gradient, so the
        // gradient flows through the attention branch alone and stops at the LoRA factors.
        const int d = g.dim, q = g.dout, M = N * g.gin * g.gin, Mo = N * g.gout * g.gout, ldq = 3 * q + AUG;
        T* gT = at<T>(ws, w.gT);
        T* dh = at<T>(ws, w.dh);
        float* dyn = at<float>(ws, w.dyn);
        float* dy = at<float>(ws, w.dy);
        T* dyT = at<T>(ws, w.dyT);
        T* da = at<T>(ws, w.da);
        T* dqkv = at<T>(ws, w.dqkv);
        T* xn = at<T>(ws, sv.xn);
        // out = y + fc2(gelu(fc1(norm2(y))))
        LSAM_TRY(launch_cast<T>(dout, q, gT, q, Mo, q, s));
        LSAM_TRY(linear<T>(gT, q, at<T>(arena, o.w2T), q, nullptr, nullptr, 0, 0, at<T>(ws, sv.pre), 4 * q, dh, 4 * q,
                           nullptr, 0, LSAM_EPI_GELU_BWD, Mo, 4 * q, q, s));
        LSAM_TRY(linear<T>(dh, 4 * q, at<T>(arena, o.w1T), 4 * q, nullptr, nullptr, 0, 0, nullptr, 0, dyn, q, nullptr, 0,
                           LSAM_EPI_OUT_F32, Mo, q, 4 * q, s));
        LSAM_TRY(launch_ln_bwd<T>(dyn, q, at<float>(ws, sv.y), q, at<float>(ws, sv.st2), at<float>(arena, o.ln2w), dout,
                                  q, dy, q, dyT, q, Mo, q, s));
        // y = shortcut + proj(attn(qkv))
        LSAM_TRY(linear<T>(dyT, q, at<T>(arena, o.wprojT), q, nullptr, nullptr, 0, 0, nullptr, 0, da, q, nullptr, 0, 0, Mo,
                           q, q, s));
        AttnBwdArgs ab{at<T>(ws, sv.qkv), 3 * q, at<float>(arena, o.bqkv), da, q, at<float>(ws, sv.lse), dqkv, ldq,
                       N, g.gin, g.gin, g.ws, g.heads, q / g.heads};
        ab.pool = g.pool;
        LSAM_TRY(launch_attn_bwd<T>(ab, s));
        // t = [dq[:, :d].B_q | dv[:, -d:].B_v] into the 32 augmented columns of dqkv
        LSAM_TRY(linear<T>(dqkv, ldq, at<T>(arena, o.bT), 3 * q, nullptr, nullptr, 0, 0, nullptr, 0, dqkv + 3 * q, ldq,
                           nullptr, 0, 0, M, AUG, 3 * q, s));
        // LoRA gradients (token reductions): dB = dqkv[:, first / last d]^T u, dA = t^T xn
        void* part = at<char>(ws, w.partial);
        float* const* G = lora_grads + 4 * gi;
        const OuterJobs oj{{xn, dqkv, xn, dqkv + 3 * q - d}, {g.kq, ldq, g.kq, ldq},
                           {dqkv + 3 * q, xn + d, dqkv + 3 * q + 16, xn + d + 16}, {ldq, g.kq, ldq, g.kq},
                           {G[0], G[1], G[2], G[3]}, {1, r, 1, r}, {d, 1, d, 1}};   // dA_q, dB_q, dA_v, dB_v
        LSAM_TRY(launch_outer_reduce4<T>(oj, M, d, r, part, s));
        if (g.need_dx) {
            float* dxn = at<float>(ws, w.dxn);
            float* dxp = at<float>(ws, w.dxprev);
            LSAM_TRY(linear<T>(dqkv, ldq, at<T>(arena, o.wqkvT), ldq, nullptr, nullptr, 0, 0, nullptr, 0, dxn, d, nullptr,
                               0, LSAM_EPI_OUT_F32, M, d, ldq, s));
            // x_in of this block is the previous LoRA block's saved output
            const float* xin = at<float>(ws, w.sv[gi - 1].xout);
            LSAM_TRY(launch_ln_bwd<T>(dxn, d, xin, d, at<float>(ws, sv.st1), at<float>(arena, o.ln1w), dy, d, dxp, d,
                                      (T*)nullptr, 0, M, d, s));
            dout = dxp;
        }
    }
    return LSAM_OK;
}

}  // namespace lsam

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int lsam_hiera_create(const lsam_hiera_config_t* c, lsam_hiera_t** out) {
    LSAM_REQUIRE(c && out, LSAM_ERR_BAD_ARG, "hiera_create: null argument");
    LSAM_REQUIRE(c->dtype == LSAM_F32 || c->dtype == LSAM_BF16, LSAM_ERR_BAD_ARG, "hiera_create: dtype");
    LSAM_REQUIRE(c->num_stages >= 1 && c->num_stages <= 4, LSAM_ERR_UNSUPPORTED, "hiera_create: 1..4 stages");
    LSAM_REQUIRE(c->img_size > 0 && c->img_size % 4 == 0, LSAM_ERR_UNSUPPORTED, "hiera_create: img_size %% 4");
    LSAM_REQUIRE(c->embed_dim > 0 && c->embed_dim % 8 == 0 && c->num_heads > 0, LSAM_ERR_UNSUPPORTED,
                 "hiera_create: embed_dim must be a multiple of 8");
    LSAM_REQUIRE(c->n_global >= 0 && c->n_global <= 8, LSAM_ERR_BAD_ARG, "hiera_create: n_global");
    LSAM_REQUIRE(c->lora_rank >= 0 && c->lora_rank <= 16, LSAM_ERR_UNSUPPORTED, "hiera_create: lora_rank <= 16");
    LSAM_REQUIRE(c->n_lora_blocks >= 0 && c->n_lora_blocks <= 8 && (c->lora_rank > 0 || c->n_lora_blocks == 0),
                 LSAM_ERR_BAD_ARG, "hiera_create: n_lora_blocks");
    lsam_hiera* h = new (std::nothrow) lsam_hiera();
    LSAM_REQUIRE(h, LSAM_ERR_BAD_ARG, "hiera_create: out of host memory");
    h->cfg = *c;
    h->esz = c->dtype == LSAM_BF16 ? 2 : 4;
    h->grid0 = c->img_size / 4;
    // geometry, as Hiera.__init__ derives it (hieradet.py:203-260)
    int depth = 0, stage_end[4];
    for (int s = 0; s < c->num_stages; ++s) {
        depth += c->stages[s];
        stage_end[s] = depth - 1;
    }
    int dim = c->embed_dim, heads = c->num_heads, cur_stage = 1, grid = h->grid0;
    for (int i = 0; i < depth; ++i) {
        BlockGeom g{};
        g.dim = dim;
        g.dout = dim;
        int ws = c->window_spec[cur_stage - 1];  // lags by one block (hieradet.py:237-240)
        for (int k = 0; k < c->n_global; ++k)
            if (c->global_att_blocks[k] == i) ws = 0;
        bool first_of_stage = false, pooled = false;
        for (int s = 0; s + 1 < c->num_stages; ++s)
            if (i - 1 == stage_end[s]) {
                first_of_stage = true;
                pooled = s < c->q_pool;
            }
        if (first_of_stage) {
            g.dout = dim * 2;
            heads *= 2;
            cur_stage += 1;
        }
        g.heads = heads;
        g.ws = ws;
        g.pool = pooled ? 2 : 1;
        g.gin = grid;
        g.gout = pooled ? grid / 2 : grid;
        g.stage = cur_stage - 1;
        g.lora = c->n_lora_blocks > 0 && i >= depth - c->n_lora_blocks;
        g.need_dx = g.lora && i > depth - c->n_lora_blocks;
        g.kq = g.lora ? g.dim + AUG : g.dim;
        g.fused_attn = c->dtype == LSAM_BF16 && !g.lora && g.dout % g.heads == 0 &&
                       winblock_supported(g.gin, g.gin, g.ws, g.pool, g.heads, g.dout / g.heads, g.dim, g.dout);
        g.fused_tail = c->dtype == LSAM_BF16 && !g.lora && !g.fused_attn && tailblock_supported(g.dout);
        g.fused_mlp = c->dtype == LSAM_BF16 && !g.lora && !g.fused_tail && mlpblock_supported(g.dout);
        g.split_proj = SPLIT_PROJ && g.fused_attn && g.fused_mlp && mlpblock_proj_supported(g.dout);
        // (the conditions under which forward_t takes the pooled-query path, plus the kernel's own)
        g.fused_kv = c->dtype == LSAM_BF16 && !g.lora && g.dim != g.dout && g.dout % g.heads == 0 &&
                     kvattn_supported(g.gin, g.gin, g.ws, g.pool, g.heads, g.dout / g.heads, g.dim, g.dout) &&
                     gemm_pool_supported(g.gin, g.gin, g.dout) && (2 * g.dout) % 224 == 0;
        g.fused_qkv4 = QKVATTN4 && c->dtype == LSAM_BF16 && !g.lora && !g.fused_attn && !g.fused_kv && g.dout % g.heads == 0 &&
                       qkvattn4_supported(g.gin, g.gin, g.ws, g.pool, g.heads, g.dout / g.heads, g.dim, g.dout) &&
                       (g.pool == 1 || (g.dim != g.dout && gemm_pool_supported(g.gin, g.gin, g.dout) && (2 * g.dout) % 224 == 0));
        h->blk.push_back(g);
        dim = g.dout;
        grid = g.gout;
    }
    h->first_lora = depth - c->n_lora_blocks;
    int rc = LSAM_OK;
    for (int i = 0; i < depth && rc == LSAM_OK; ++i) {
        const BlockGeom& g = h->blk[i];
        if (g.dout % g.heads != 0 || (g.dout / g.heads) % 8 != 0 || g.dout / g.heads > 128) {
            set_error("hiera_create: block %d head_dim %d unsupported (need %%8==0, <=128)", i, g.dout / g.heads);
            rc = LSAM_ERR_UNSUPPORTED;
        } else if (g.pool == 2 && (g.ws <= 0 || g.ws % 2 || g.gin % 2)) {
            set_error("hiera_create: q-pool block %d needs an even window and grid", i);
            rc = LSAM_ERR_UNSUPPORTED;
        } else if (g.lora && (g.ws <= 0 || (g.ws / g.pool) * (g.ws / g.pool) > 64)) {
            set_error("hiera_create: LoRA block %d must be a windowed block with <= 64 queries per window", i);
            rc = LSAM_ERR_UNSUPPORTED;
        } else if (g.lora && g.need_dx && (g.dim != g.dout || g.pool != 1)) {
            set_error("hiera_create: a stage-entry block (dim != dim_out / q-pool) can only be the FIRST LoRA block "
                      "(block %d): no gradient path through its pooled shortcut is built", i);
            rc = LSAM_ERR_UNSUPPORTED;
        }
    }
    if (rc != LSAM_OK) {
        delete h;
        return rc;
    }
    for (int s = 0; s < c->num_stages; ++s) {
        h->stage_ch[s] = h->blk[stage_end[s]].dout;
        h->stage_grid[s] = h->blk[stage_end[s]].gout;
    }
    // arena layout
    Bump b;
    const size_t e = h->esz;
    h->wpatch = b.take((size_t)c->embed_dim * PATCH_K * e);
    h->direct_patch = c->dtype == LSAM_BF16 && patch_embed_supported(c->embed_dim);
    h->wpatch_img = h->direct_patch ? b.take(patch_embed_image_bytes(c->embed_dim)) : 0;
    h->bpatch = b.take((size_t)c->embed_dim * 4);
    h->pos = b.take((size_t)h->grid0 * h->grid0 * c->embed_dim * 4);
    for (int i = 0; i < depth; ++i) {
        const BlockGeom& g = h->blk[i];
        const size_t d = g.dim, q = g.dout;
        BlockOff o{};
        o.ln1w = b.take(d * 4);
        o.ln1b = b.take(d * 4);
        o.wqkv = b.take(3 * q * g.kq * e);
        o.bqkv = b.take(3 * q * 4);
        o.bqkvT = b.take(3 * q * e);
        o.wproj = b.take(q * q * e);
        o.bproj = b.take(q * 4);
        o.ln2w = b.take(q * 4);
        o.ln2b = b.take(q * 4);
        o.w1 = b.take(4 * q * q * e);
        o.b1 = b.take(4 * q * 4);
        o.w2 = b.take(4 * q * q * e);
        o.b2 = b.take(q * 4);
        if (d != q) {   // non-LoRA blocks: the q rows of the qkv projection follow the shortcut projection's (see pack_t)
            o.wsc = b.take((g.lora ? 1 : 2) * q * d * e);
            o.bsc = b.take((g.lora ? 1 : 2) * q * 4);
        }
        if (g.fused_attn) {
            o.wimg = b.take(winblock_image_bytes(g.heads));
            o.wbias = b.take(winblock_bias_floats(g.heads) * 4);
        }
        if (g.fused_mlp) {
            o.mimg = b.take(mlpblock_image_bytes(g.dout));
            o.mbias = b.take(mlpblock_bias_floats(g.dout) * 4);
            if (g.split_proj) {
                o.mpimg = b.take(mlpblock_proj_image_bytes(g.dout));
                o.mpbias = b.take(16 * 14 * 4);
            }
        }
        if (g.fused_tail) {
            o.timg = b.take(tailblock_image_bytes(g.dout));
            o.tvec = b.take(tailblock_vec_floats(g.dout) * 4);
        }
        if (g.fused_kv) o.kvimg = b.take(kvattn_image_bytes());
        if (g.fused_qkv4) o.qaimg = b.take(qkvattn4_image_bytes(g.heads, g.pool == 2));
        if (c->dtype == LSAM_BF16 && STREAMLIN && !g.lora && !g.fused_attn && !g.fused_qkv4 && g.kq == g.dim) {
            // which rows of the qkv weight the block's projection GEMM multiplies: all of them, or k | v only when the pooled
            // queries come from the pooling epilogue (the plan-time part of forward_t's `qpooled`) and no fused k | v kernel runs
            const bool kv_only = g.pool == 2 && d != q && (q / g.heads) % 8 == 0 && gemm_pool_supported(g.gin, g.gin, (int)q) && kv_width_ok((long)q);
            const int n0 = kv_only ? (int)q : 0, n = (int)(3 * q) - n0;
            if (!(kv_only && g.fused_kv) && streamlin_supported((int)d, n)) {
                o.simg = b.take(streamlin_image_bytes((int)d, n));
                o.simg_n0 = n0;
                o.simg_n = n;
            }
        }
        if (g.lora) {
            o.w2T = b.take(4 * q * q * e);
            o.w1T = b.take(4 * q * q * e);
            o.wprojT = b.take(q * q * e);
            o.wqkvT = b.take(d * (3 * q + AUG) * e);
            o.acat = b.take((size_t)AUG * d * e);
            o.bT = b.take((size_t)AUG * 3 * q * e);
        }
        h->off.push_back(o);
    }
    h->arena_bytes = b.cur;
    *out = h;
    return LSAM_OK;
}

void lsam_hiera_destroy(lsam_hiera_t* h) { delete h; }

int lsam_hiera_num_blocks(const lsam_hiera_t* h) { return h ? (int)h->blk.size() : 0; }

int lsam_hiera_block_info(const lsam_hiera_t* h, int i, int* dim, int* dim_out, int* heads, int* window, int* pool,
                          int* grid) {
    LSAM_REQUIRE(h && i >= 0 && i < (int)h->blk.size(), LSAM_ERR_BAD_ARG, "block_info: bad index");
    const BlockGeom& g = h->blk[i];
    if (dim) *dim = g.dim;
    if (dim_out) *dim_out = g.dout;
    if (heads) *heads = g.heads;
    if (window) *window = g.ws;
    if (pool) *pool = g.pool == 2;
    if (grid) *grid = g.gin;
    return LSAM_OK;
}

int lsam_hiera_stage_info(const lsam_hiera_t* h, int s, int* channels, int* grid) {
    LSAM_REQUIRE(h && s >= 0 && s < h->cfg.num_stages, LSAM_ERR_BAD_ARG, "stage_info: bad index");
    if (channels) *channels = h->stage_ch[s];
    if (grid) *grid = h->stage_grid[s];
    return LSAM_OK;
}

size_t lsam_hiera_arena_bytes(const lsam_hiera_t* h) { return h ? h->arena_bytes : 0; }

size_t lsam_hiera_workspace_bytes(const lsam_hiera_t* h, int batch, int with_grad) {
    if (!h || batch <= 0) return 0;
    return plan_ws(h, batch, with_grad != 0).total;
}

int lsam_hiera_pack(lsam_hiera_t* h, const lsam_block_params_t* host_blocks, const float* patch_w,
                    const float* patch_b, const float* pos, void* arena, size_t arena_bytes, lsam_stream_t stream) {
    LSAM_REQUIRE(h && host_blocks && patch_w && patch_b && pos && arena, LSAM_ERR_BAD_ARG, "hiera_pack: null argument");
    LSAM_REQUIRE(arena_bytes >= h->arena_bytes, LSAM_ERR_WORKSPACE, "hiera_pack: arena %zu < %zu bytes", arena_bytes,
                 h->arena_bytes);
    LSAM_REQUIRE((reinterpret_cast<uintptr_t>(arena) & 255) == 0, LSAM_ERR_BAD_ARG, "hiera_pack: arena must be 256-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h->cfg.dtype == LSAM_BF16) return pack_t<bf16>(h, host_blocks, patch_w, patch_b, pos, arena, s);
    return pack_t<float>(h, host_blocks, patch_w, patch_b, pos, arena, s);
}

int lsam_hiera_pack_lora(lsam_hiera_t* h, const lsam_lora_params_t* host_lora, void* arena, lsam_stream_t stream) {
    LSAM_REQUIRE(h && arena, LSAM_ERR_BAD_ARG, "pack_lora: null argument");
    if (h->cfg.n_lora_blocks == 0) return LSAM_OK;
    LSAM_REQUIRE(host_lora, LSAM_ERR_BAD_ARG, "pack_lora: null LoRA parameter table");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h->cfg.dtype == LSAM_BF16) return pack_lora_t<bf16>(h, host_lora, arena, s);
    return pack_lora_t<float>(h, host_lora, arena, s);
}

int lsam_hiera_forward(lsam_hiera_t* h, const float* img, int batch, float* const* host_stage_out, void* arena,
                       void* workspace, size_t workspace_bytes, int with_grad, lsam_stream_t stream) {
    LSAM_REQUIRE(h && img && host_stage_out && arena && workspace, LSAM_ERR_BAD_ARG, "hiera_forward: null argument");
    LSAM_REQUIRE(batch > 0, LSAM_ERR_BAD_ARG, "hiera_forward: batch must be positive");
    for (int s = 0; s < h->cfg.num_stages; ++s)
        LSAM_REQUIRE(host_stage_out[s] && aligned16(host_stage_out[s]), LSAM_ERR_BAD_ARG,
                     "hiera_forward: stage output %d null or misaligned", s);
    LSAM_REQUIRE(aligned16(img) && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0, LSAM_ERR_BAD_ARG,
                 "hiera_forward: img must be 16-byte and workspace 256-byte aligned");
    const bool wg = with_grad != 0 && h->cfg.n_lora_blocks > 0;
    const FwdWs w = plan_ws(h, batch, wg);
    LSAM_REQUIRE(workspace_bytes >= w.total, LSAM_ERR_WORKSPACE, "hiera_forward: workspace %zu < %zu bytes",
                 workspace_bytes, w.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h->cfg.dtype == LSAM_BF16) return forward_t<bf16>(h, img, batch, host_stage_out, arena, workspace, w, wg, s);
    return forward_t<float>(h, img, batch, host_stage_out, arena, workspace, w, wg, s);
}

int lsam_hiera_backward(lsam_hiera_t* h, const float* grad_out, int batch, float* const* host_lora_grads, void* arena,
                        void* workspace, size_t workspace_bytes, lsam_stream_t stream) {
    LSAM_REQUIRE(h && grad_out && host_lora_grads && arena && workspace, LSAM_ERR_BAD_ARG, "hiera_backward: null argument");
    LSAM_REQUIRE(h->cfg.n_lora_blocks > 0, LSAM_ERR_BAD_ARG, "hiera_backward: plan has no LoRA blocks");
    LSAM_REQUIRE(batch > 0 && aligned16(grad_out), LSAM_ERR_BAD_ARG, "hiera_backward: batch / grad alignment");
    for (int i = 0; i < 4 * h->cfg.n_lora_blocks; ++i)
        LSAM_REQUIRE(host_lora_grads[i], LSAM_ERR_BAD_ARG, "hiera_backward: null gradient buffer %d", i);
    const FwdWs w = plan_ws(h, batch, true);
    LSAM_REQUIRE(workspace_bytes >= w.total, LSAM_ERR_WORKSPACE, "hiera_backward: workspace %zu < %zu bytes",
                 workspace_bytes, w.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int hi = h->cfg.n_lora_blocks - 1;
    if (h->cfg.dtype == LSAM_BF16) return backward_t<bf16>(h, grad_out, batch, host_lora_grads, arena, workspace, w, hi, 0, s);
    return backward_t<float>(h, grad_out, batch, host_lora_grads, arena, workspace, w, hi, 0, s);
}

int lsam_hiera_backward_block(lsam_hiera_t* h, const float* grad_out, int batch, int lora_block,
                              float* const* host_lora_grads, void* arena, void* workspace, size_t workspace_bytes,
                              lsam_stream_t stream) {
    LSAM_REQUIRE(h && host_lora_grads && arena && workspace, LSAM_ERR_BAD_ARG, "hiera_backward_block: null argument");
    LSAM_REQUIRE(lora_block >= 0 && lora_block < h->cfg.n_lora_blocks, LSAM_ERR_BAD_ARG,
                 "hiera_backward_block: block %d of %d", lora_block, h->cfg.n_lora_blocks);
    const bool top = lora_block == h->cfg.n_lora_blocks - 1;
    LSAM_REQUIRE(!top || (grad_out && aligned16(grad_out)), LSAM_ERR_BAD_ARG,
                 "hiera_backward_block: the last LoRA block needs the stage-output gradient");
    LSAM_REQUIRE(batch > 0, LSAM_ERR_BAD_ARG, "hiera_backward_block: batch");
    for (int i = 0; i < 4; ++i)
        LSAM_REQUIRE(host_lora_grads[4 * lora_block + i], LSAM_ERR_BAD_ARG, "hiera_backward_block: null gradient buffer %d", i);
    const FwdWs w = plan_ws(h, batch, true);
    LSAM_REQUIRE(workspace_bytes >= w.total, LSAM_ERR_WORKSPACE, "hiera_backward_block: workspace %zu < %zu bytes",
                 workspace_bytes, w.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h->cfg.dtype == LSAM_BF16)
        return backward_t<bf16>(h, grad_out, batch, host_lora_grads, arena, workspace, w, lora_block, lora_block, s);
    return backward_t<float>(h, grad_out, batch, host_lora_grads, arena, workspace, w, lora_block, lora_block, s);
}

}  // extern "C"
