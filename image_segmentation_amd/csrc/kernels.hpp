// Internal typed launchers shared by the C ABI (capi.hip) and the trunk plan (hiera.hip).
#pragma once
#include "common.hpp"

namespace lsam {

struct GemmArgs {
    const void* A; int lda;
    const void* W; int ldw;
    const float* bias;
    const float* res; int ldres; int res_row_mod;
    const void* aux; int ldaux;
    void* out; int ldo;
    void* out2; int ldo2;
    int flags;
    int M, N, K;
    // optional 2x2 max-pool of the OUTPUT rows (rows = tokens of an [pool_n][pool_h][pool_w] grid, M = their
    // product): out then has M / 4 rows.  bf16 operands, fp32 out, N % 224 == 0, bias only (see gemm_pool_supported).
    int pool_h = 0, pool_w = 0;
    // pooled epilogue only: output columns >= split_n go to out2 (fp32, leading dimension ldo2) at column - split_n, so ONE
    // pooled GEMM over concatenated weights serves two consumers (a stage-entry block's shortcut and its pooled queries,
    // hieradet.py:139-140 and :64-67, both products of the same LN1 rows).  split_n % 224 == 0.
    int split_n = 0;
};
bool gemm_pool_supported(int H, int W, int N);
template <typename T> int launch_gemm(const GemmArgs& a, hipStream_t s);
int prof_arm(int capacity);
int prof_read(int max, float* ms, double* flops, int* cfg, int* mnk);
int prof_begin(hipStream_t s);   // live HIP-event timing of one launch while armed (bench.py's roofline)
void prof_end(int rec, hipStream_t s, double flops, int cfg, int m, int n, int k);
// RAII form: brackets everything launched on `s` within the scope.  `work` = algorithmic FLOPs for MFMA-bound kernels
// (cfg >= 1000) and algorithmic HBM BYTES for the HBM-bound ones (cfg < 1000).  Scopes must not nest.
enum {
    PROF_LN_FWD = 100, PROF_LN_BWD = 101, PROF_CAST = 102, PROF_LATERAL = 103, PROF_PATCH_EMBED = 104, PROF_WINBLOCK = 105,
    PROF_MLPBLOCK = 106, PROF_MAXPOOL = 107, PROF_SPATIAL_MEAN = 108, PROF_OUTER = 109, PROF_NECK = 110,
    PROF_ATTN_SMALL = 111, PROF_KVATTN = 5003, PROF_QKVATTN4 = 5004,                                      /* windows with <= 16 queries per wave: bytes */
    PROF_ATTN_GLOBAL = 5000, PROF_ATTN_WIN = 5001, PROF_ATTN_BWD = 5002   /* FLOPs */
};
struct ProfScope {
    int rec; hipStream_t s; double work; int cfg, m, n, k;
    ProfScope(hipStream_t s_, double work_, int cfg_, int m_, int n_, int k_)
        : rec(prof_begin(s_)), s(s_), work(work_), cfg(cfg_), m(m_), n(n_), k(k_) {}
    ~ProfScope() { prof_end(rec, s, work, cfg, m, n, k); }
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;
};

template <typename T>
int launch_pack_matrix(const float* src, long src_rs, long src_cs, int rows, int cols, T* dst, int ldd,
                       hipStream_t s);
template <typename T>
int launch_ln_fwd(const float* x, int ldx, const float* gamma, const float* beta, T* y, int ldy,
                  float* stats, int M, int C, float eps, hipStream_t s);
template <typename T>
int launch_ln_bwd(const float* dy, int lddy, const float* x, int ldx, const float* stats,
                  const float* gamma, const float* dres, int lddres, float* dx, int lddx, T* dx_t,
                  int lddxt, int M, int C, hipStream_t s);

struct AttnArgs {
    const void* qkv; int ldqkv;
    const float* bias;
    void* out; int ldo;
    float* lse;
    int N, H, W, ws, pool, nh, hd;
    const void* bias_t;  // optional T-typed copy of bias (enables the LDS-DMA kernel)
    // optional (pool == 2): q already max-pooled, fp32 [N*(H/2)*(W/2)][ldqp] in output-token order (the q columns
    // of the qkv GEMM taken through the pooling epilogue); the q columns of qkv are then never read
    const float* qpool = nullptr; int ldqp = 0;
};
template <typename T> int launch_attn_fwd(const AttnArgs& a, hipStream_t s);

struct AttnBwdArgs {
    const void* qkv; int ldqkv;
    const float* bias;
    const void* dout; int lddo;
    const float* lse;
    void* dqkv; int lddqkv;
    int N, H, W, ws, nh, hd;
    int pool = 1;   // 2: q was max-pooled 2x2 inside each window; dout / lse are then indexed by the pooled (output) tokens
};
template <typename T> int launch_attn_bwd(const AttnBwdArgs& a, hipStream_t s);

int launch_maxpool2x2(const float* x, float* out, int N, int H, int W, int C, hipStream_t s);
int launch_spatial_mean(const float* x, float* out, int N, int HW, int C, hipStream_t s);
template <typename T>
int launch_cast(const float* src, int lds, T* dst, int ldd, int rows, int cols, hipStream_t s);
template <typename T>
int launch_im2col_patch(const float* img, T* cols, int ldc, int N, int S, hipStream_t s);

// direct PatchEmbed convolution + bias + position embedding (patchembed.hip), bf16 MFMA, fp32 out
bool patch_embed_supported(int e0);
size_t patch_embed_image_bytes(int e0);
int launch_pack_patch_embed(const float* w, int e0, void* img, hipStream_t s);
int launch_patch_embed(const float* frames, const void* wimg, const float* bias, const float* pos, float* out, int N, int S,
                       int e0, hipStream_t s);

// FPN lateral 1x1 convolution from the fp32 stage output, bf16 out (lateral.hip)
bool lateral_supported(int K, int N);
size_t lateral_image_bytes(int K, int N);
int launch_pack_lateral(const float* w, int K, int N, void* img, hipStream_t s);
int launch_lateral(const float* x, const void* img, const float* bias, void* out, int M, int K, int N, hipStream_t s);

// small-K projection with LDS-resident weights, bf16 in / bf16 out + bias (lateral.hip): the stage-2 qkv / k | v Linears
bool streamlin_supported(int K, int N);
size_t streamlin_image_bytes(int K, int N);
int launch_pack_streamlin(const float* w, int K, int N, void* img, hipStream_t s);   // w fp32 [N][K] row-major
int launch_streamlin(const void* A, int lda, const void* img, const float* bias, void* out, int ldo, int M, int K, int N, hipStream_t s);

// FPN neck level: lateral 1x1 conv (+ nearest-x2 top-down add) from the fp32 stage output, T out (neck.hip)
bool neck_level_supported(int K, int N);
size_t neck_image_bytes(int K, int N, size_t esz);
template <typename T> int launch_pack_neck(const float* w, int K, int N, void* img, hipStream_t s);
template <typename T>
int launch_neck_level(const float* x, const void* img, const float* bias, const void* top, void* out, int Nimg, int H, int W,
                      int K, int N, hipStream_t s);
// the same with two-term bf16 operands (fp32-accurate product), fp32 top-down addend / second output
size_t neck_split_image_bytes(int K, int N);
int launch_pack_neck_split(const float* w, int K, int N, void* img, hipStream_t s);
int launch_neck_level_split(const float* x, const void* img, const float* bias, const float* top, void* out, float* out32,
                            int Nimg, int H, int W, int K, int N, hipStream_t s);
template <typename T>
int launch_neck_topdown_bwd(const void* gf, const void* gc, void* out, int Nimg, int Hc, int Wc, int C, hipStream_t s);

// fused k | v projection + window attention of a q-pool stage-entry block with 8x8 windows (kvattn.hip)
struct KvAttnArgs {
    const bf16* xn; int ldx;      // LN1 output rows [N*H*W][dim]
    const float* qp; int ldqp;    // pooled queries, fp32 [N*(H/2)*(W/2)][nh*hd] (lsam_linear_maxpool2x2)
    const char* img;              // launch_pack_kvattn
    bf16* out; int ldo;           // attention output [N*(H/2)*(W/2)][nh*hd]
    int N, H, W, dim, hd;
    float c_log2;                 // log2(e) / sqrt(hd)
};
size_t kvattn_image_bytes();
bool kvattn_supported(int H, int W, int ws, int pool, int nh, int hd, int dim, int dout);
int launch_pack_kvattn(const float* wqkv, const float* bqkv, int dim, int nh, int hd, void* img, hipStream_t s);
int launch_kvattn(const KvAttnArgs& a, hipStream_t s);

// fused qkv (or k | v) projection + 4x4-window attention (qkvattn.hip): stage-2 blocks and the q-pool block behind them
#ifndef QKVATTN4_TT
#define QKVATTN4_TT 2   // windows per wave and pass
#endif
#ifndef QKVATTN4_NW
#define QKVATTN4_NW 8   // waves per workgroup
#endif
struct QkvAttn4Args {
    const bf16* xn; int ldx;      // LN1 output rows [N*H*W][dim]
    const float* qp; int ldqp;    // nullptr: q projected here; else pooled queries, fp32 [N*(H/2)*(W/2)][nh*hd]
    const char* img;              // launch_pack_qkvattn4 (packed for the same variant)
    bf16* out; int ldo;           // attention output [N*H*W][nh*hd], or [N*(H/2)*(W/2)][nh*hd] with pooled queries
    int N, H, W, dim, nh, hd;
    float c_log2;                 // log2(e) / sqrt(hd)
};
size_t qkvattn4_image_bytes(int nh, bool poolq);
bool qkvattn4_supported(int H, int W, int ws, int pool, int nh, int hd, int dim, int dout);
int launch_pack_qkvattn4(const float* wqkv, const float* bqkv, int dim, int nh, int hd, bool poolq, void* img, hipStream_t s);
int launch_qkvattn4(const QkvAttn4Args& a, hipStream_t s);

// fused attention half of a same-dim 8x8-window block (winblock.hip)
struct WinBlockArgs {
    const float* x;       // [N*H*W][D] fp32 residual stream (D = nh * hd)
    float* y;             // x + proj(attn(qkv(LN1(x)))); may alias x
    const float* gamma; const float* beta;   // LayerNorm 1
    const void* img;      // packed weight image (launch_pack_winblock)
    const float* bias_p;  // padded qkv + proj biases
    int N, H, W, hd;
    float eps, c_log2;    // LN epsilon; softmax scale * log2(e)
    void* a_out = nullptr;   // attention-only form: the attention output rows, bf16 [N*H*W][D], instead of y (no proj, no residual)
};
size_t winblock_image_bytes(int nh);
size_t winblock_bias_floats(int nh);
bool winblock_supported(int H, int W, int ws, int pool, int nh, int hd, int dim, int dout);
int launch_pack_winblock(const float* wqkv, const float* bqkv, const float* wproj, const float* bproj, int nh, int hd,
                         void* img, float* bias_p, hipStream_t s);
int launch_winblock_attn(const WinBlockArgs& a, int nh, hipStream_t s);

// fused MLP half of a block: y = x + fc2(GELU(fc1(LN2(x))))  (mlpblock.hip)
struct MlpBlockArgs {
    const float* x;       // [M][d] fp32 residual stream
    float* y;             // may alias x
    const float* gamma; const float* beta;   // LayerNorm 2
    const void* img;      // packed, chunked weight image (launch_pack_mlpblock)
    const float* bias_p;  // fc1 bias padded to 64 * nch, then fc2 bias
    int M, d, nch;
    float eps;
    // optional: LayerNorm of the block's OUTPUT with the next block's norm1 parameters, T = bf16 [M][ldn]
    void* xn_next = nullptr; int ldn = 0; const float* gamma_n = nullptr; const float* beta_n = nullptr;
    // optional: the attention projection in front (hieradet.py:79,163): x + proj(a) takes the place of x.  a bf16 [M][lda] (the
    // attention output of launch_winblock_attn's attention-only form), pimg / pbias from launch_pack_mlpblock_proj
    const void* a = nullptr; int lda = 0; const void* pimg = nullptr; const float* pbias = nullptr;
};
bool mlpblock_supported(int d);
bool mlpblock_proj_supported(int d);
size_t mlpblock_proj_image_bytes(int d);
int launch_pack_mlpblock_proj(const float* wproj, const float* bproj, int d, void* pimg, float* pbias, hipStream_t s);
int mlpblock_chunks(int d);
size_t mlpblock_image_bytes(int d);
size_t mlpblock_bias_floats(int d);
int launch_pack_mlpblock(const float* w1, const float* b1, const float* w2, const float* b2, int d, void* img,
                         float* bias_p, hipStream_t s);
int launch_mlpblock(const MlpBlockArgs& a, hipStream_t s);

// fused tail of a block: out = y + fc2(GELU(fc1(LN2(y)))), y = res + proj(a); optional next LN1  (tailblock.hip)
struct TailBlockArgs {
    const void* a; int lda;    // bf16 [M][lda] attention output
    const float* res;          // fp32 [M][d] shortcut
    float* out;                // fp32 [M][d]; may alias res
    const void* img;           // packed chunk stream (launch_pack_tailblock)
    const float* vec;          // packed fp32 vectors: proj bias, LN2 gamma, LN2 beta, fc2 bias
    void* xn_next; int ldn;    // optional bf16 [M][ldn]: LayerNorm(out) with gamma_n / beta_n
    const float* gamma_n; const float* beta_n;
    int M, d, nch;
    float eps;
};
bool tailblock_supported(int d);
int tailblock_chunks(int d);
size_t tailblock_image_bytes(int d);
size_t tailblock_vec_floats(int d);
int launch_pack_tailblock(const float* wproj, const float* bproj, const float* g2, const float* be2, const float* w1,
                          const float* b1, const float* w2, const float* b2, int d, void* img, float* vec,
                          hipStream_t s);
int launch_tailblock(const TailBlockArgs& a, hipStream_t s);

size_t outer_reduce_ws(int M, int C);
struct OuterJobs {   // four independent X^T Y token reductions with common M, C, R (one LoRA block's gradients)
    const void* X[4]; int ldx[4];
    const void* Y[4]; int ldy[4];
    float* out[4]; long cs[4], ks[4];
};
template <typename T> int launch_outer_reduce4(const OuterJobs& j, int M, int C, int R, void* partial, hipStream_t s);   // partial: 4 x outer_reduce_ws
template <typename T>
int launch_outer_reduce(const T* X, int ldx, const T* Y, int ldy, int M, int C, int R, float* out,
                        long out_cs, long out_ks, void* partial, hipStream_t s);
// the same with the partial sums in a library-owned static device array (no caller workspace, no allocation)
size_t outer_reduce_static_max_bytes();
int outer_reduce_static_init();
template <typename T>
int launch_outer_reduce_static(const T* X, int ldx, const T* Y, int ldy, int M, int C, int R, float* out,
                               long out_cs, long out_ks, hipStream_t s);

}  // namespace lsam
