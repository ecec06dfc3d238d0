// extern "C" entry points for the unit-level operators (see include/lsam.h).
#include <stdarg.h>

#include "kernels.hpp"

namespace lsam {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace lsam

using namespace lsam;

#define DISPATCH(dtype, CALL_F32, CALL_BF16)                              \
    do {                                                                  \
        if ((dtype) == LSAM_F32) return (CALL_F32);                       \
        if ((dtype) == LSAM_BF16) return (CALL_BF16);                     \
        set_error("unknown dtype %d", (int)(dtype));                      \
        return LSAM_ERR_BAD_ARG;                                          \
    } while (0)

extern "C" {

const char* lsam_last_error(void) { return g_err; }
int lsam_version(void) { return 1; }

int lsam_linear(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, const float* res,
                int ldres, int res_row_mod, const void* aux, int ldaux, void* out, int ldo, void* out2, int ldo2,
                int flags, int M, int N, int K, lsam_stream_t stream) {
    GemmArgs a{A, lda, W, ldw, bias, res, ldres, res_row_mod, aux, ldaux, out, ldo, out2, ldo2, flags, M, N, K};
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype, launch_gemm<float>(a, s), launch_gemm<bf16>(a, s));
}

int lsam_linear_maxpool2x2(const void* A, int lda, const void* W, int ldw, const float* bias, float* out, int ldo,
                           int N_img, int H, int Wd, int N, int K, lsam_stream_t stream) {
    LSAM_REQUIRE(N_img > 0 && gemm_pool_supported(H, Wd, N), LSAM_ERR_UNSUPPORTED,
                 "linear_maxpool2x2: grid %dx%d, N %d not supported (need W %% 32 == 0, whole 256-token patches, N %% 224 == 0)",
                 H, Wd, N);
    GemmArgs a{A, lda, W, ldw, bias, nullptr, 0, 0, nullptr, 0, out, ldo, nullptr, 0, LSAM_EPI_OUT_F32, N_img * H * Wd, N, K};
    a.pool_h = H;
    a.pool_w = Wd;
    return launch_gemm<bf16>(a, static_cast<hipStream_t>(stream));
}

int lsam_pack_matrix(int dtype, const float* src, long src_rs, long src_cs, int rows, int cols, void* dst, int ldd,
                     lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype, launch_pack_matrix<float>(src, src_rs, src_cs, rows, cols, static_cast<float*>(dst), ldd, s),
             launch_pack_matrix<bf16>(src, src_rs, src_cs, rows, cols, static_cast<bf16*>(dst), ldd, s));
}

int lsam_layernorm_fwd(int dtype, const float* x, int ldx, const float* gamma, const float* beta, void* y, int ldy,
                       float* stats, int M, int C, float eps, lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype, launch_ln_fwd<float>(x, ldx, gamma, beta, static_cast<float*>(y), ldy, stats, M, C, eps, s),
             launch_ln_fwd<bf16>(x, ldx, gamma, beta, static_cast<bf16*>(y), ldy, stats, M, C, eps, s));
}

int lsam_layernorm_bwd(int dtype, const float* dy, int lddy, const float* x, int ldx, const float* stats,
                       const float* gamma, const float* dres, int lddres, float* dx, int lddx, void* dx_t, int lddxt,
                       int M, int C, lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype,
             launch_ln_bwd<float>(dy, lddy, x, ldx, stats, gamma, dres, lddres, dx, lddx, static_cast<float*>(dx_t),
                                  lddxt, M, C, s),
             launch_ln_bwd<bf16>(dy, lddy, x, ldx, stats, gamma, dres, lddres, dx, lddx, static_cast<bf16*>(dx_t),
                                 lddxt, M, C, s));
}

int lsam_attention_fwd(int dtype, const void* qkv, int ldqkv, const float* bias, const void* bias_t, void* out, int ldo,
                       float* lse, int N, int H, int W, int ws, int pool, int nh, int hd, lsam_stream_t stream) {
    AttnArgs a{qkv, ldqkv, bias, out, ldo, lse, N, H, W, ws, pool, nh, hd, bias_t};
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype, launch_attn_fwd<float>(a, s), launch_attn_fwd<bf16>(a, s));
}

int lsam_attention_fwd_pooled_q(int dtype, const void* qkv, int ldqkv, const float* bias, const void* bias_t,
                                const float* q_pooled, int ld_qp, void* out, int ldo, float* lse, int N, int H, int W,
                                int ws, int nh, int hd, lsam_stream_t stream) {
    LSAM_REQUIRE(q_pooled, LSAM_ERR_BAD_ARG, "attention_fwd_pooled_q: q_pooled is null");
    AttnArgs a{qkv, ldqkv, bias, out, ldo, lse, N, H, W, ws, 2, nh, hd, bias_t};
    a.qpool = q_pooled;
    a.ldqp = ld_qp;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype, launch_attn_fwd<float>(a, s), launch_attn_fwd<bf16>(a, s));
}

int lsam_attention_bwd(int dtype, const void* qkv, int ldqkv, const float* bias, const void* dout, int lddo,
                       const float* lse, void* dqkv, int lddqkv, int N, int H, int W, int ws, int nh, int hd,
                       lsam_stream_t stream) {
    AttnBwdArgs a{qkv, ldqkv, bias, dout, lddo, lse, dqkv, lddqkv, N, H, W, ws, nh, hd};
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype, launch_attn_bwd<float>(a, s), launch_attn_bwd<bf16>(a, s));
}

int lsam_attention_bwd_qpool(int dtype, const void* qkv, int ldqkv, const float* bias, const void* dout, int lddo,
                             const float* lse, void* dqkv, int lddqkv, int N, int H, int W, int ws, int pool, int nh, int hd,
                             lsam_stream_t stream) {
    AttnBwdArgs a{qkv, ldqkv, bias, dout, lddo, lse, dqkv, lddqkv, N, H, W, ws, nh, hd};
    a.pool = pool;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype, launch_attn_bwd<float>(a, s), launch_attn_bwd<bf16>(a, s));
}

int lsam_spatial_mean(const float* x, float* out, int N, int HW, int C, lsam_stream_t stream) {
    return launch_spatial_mean(x, out, N, HW, C, static_cast<hipStream_t>(stream));
}
int lsam_maxpool2x2(const float* x, float* out, int N, int H, int W, int C, lsam_stream_t stream) {
    return launch_maxpool2x2(x, out, N, H, W, C, static_cast<hipStream_t>(stream));
}

int lsam_cast(int dtype, const float* src, int lds, void* dst, int ldd, int rows, int cols, lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype, launch_cast<float>(src, lds, static_cast<float*>(dst), ldd, rows, cols, s),
             launch_cast<bf16>(src, lds, static_cast<bf16*>(dst), ldd, rows, cols, s));
}

int lsam_im2col_patch(int dtype, const float* img, void* cols, int ldc, int N, int S, lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype, launch_im2col_patch<float>(img, static_cast<float*>(cols), ldc, N, S, s),
             launch_im2col_patch<bf16>(img, static_cast<bf16*>(cols), ldc, N, S, s));
}

size_t lsam_patch_embed_image_bytes(int embed_dim) { return patch_embed_supported(embed_dim) ? patch_embed_image_bytes(embed_dim) : 0; }
int lsam_patch_embed_pack(const float* proj_w, int embed_dim, void* image, lsam_stream_t stream) {
    LSAM_REQUIRE(patch_embed_supported(embed_dim), LSAM_ERR_UNSUPPORTED, "patch_embed: embed_dim %d not supported", embed_dim);
    return launch_pack_patch_embed(proj_w, embed_dim, image, static_cast<hipStream_t>(stream));
}
int lsam_patch_embed_fwd(const float* frames, const void* image, const float* proj_b, const float* pos, float* out, int N,
                         int S, int embed_dim, lsam_stream_t stream) {
    LSAM_REQUIRE(patch_embed_supported(embed_dim), LSAM_ERR_UNSUPPORTED, "patch_embed: embed_dim %d not supported", embed_dim);
    return launch_patch_embed(frames, image, proj_b, pos, out, N, S, embed_dim, static_cast<hipStream_t>(stream));
}

size_t lsam_lateral_conv_image_bytes(int in_channels, int out_channels) {
    return lateral_supported(in_channels, out_channels) ? lateral_image_bytes(in_channels, out_channels) : 0;
}
int lsam_lateral_conv_pack(const float* weight, int in_channels, int out_channels, void* image, lsam_stream_t stream) {
    return launch_pack_lateral(weight, in_channels, out_channels, image, static_cast<hipStream_t>(stream));
}
int lsam_lateral_conv_fwd(const float* x, const void* image, const float* bias, void* out, int M, int in_channels,
                          int out_channels, lsam_stream_t stream) {
    return launch_lateral(x, image, bias, out, M, in_channels, out_channels, static_cast<hipStream_t>(stream));
}

size_t lsam_stream_linear_image_bytes(int in_features, int out_features) { return streamlin_image_bytes(in_features, out_features); }
int lsam_stream_linear_pack(const float* weight, int in_features, int out_features, void* image, lsam_stream_t stream) {
    return launch_pack_streamlin(weight, in_features, out_features, image, static_cast<hipStream_t>(stream));
}
int lsam_stream_linear_fwd(const void* A, int lda, const void* image, const float* bias, void* out, int ldo, int M,
                           int in_features, int out_features, lsam_stream_t stream) {
    return launch_streamlin(A, lda, image, bias, out, ldo, M, in_features, out_features, static_cast<hipStream_t>(stream));
}

size_t lsam_neck_conv_image_bytes(int dtype, int in_channels, int out_channels) {
    return neck_image_bytes(in_channels, out_channels, dtype == LSAM_BF16 ? 2 : 4);
}
int lsam_neck_conv_pack(int dtype, const float* weight, int in_channels, int out_channels, void* image, lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype, launch_pack_neck<float>(weight, in_channels, out_channels, image, s),
             launch_pack_neck<bf16>(weight, in_channels, out_channels, image, s));
}
int lsam_neck_level_fwd(int dtype, const float* x, const void* image, const float* bias, const void* top, void* out, int N,
                        int H, int W, int in_channels, int out_channels, lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype, launch_neck_level<float>(x, image, bias, top, out, N, H, W, in_channels, out_channels, s),
             launch_neck_level<bf16>(x, image, bias, top, out, N, H, W, in_channels, out_channels, s));
}
int lsam_neck_topdown_bwd(int dtype, const void* g_fine, const void* g_coarse, void* out, int N, int Hc, int Wc, int C,
                          lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype, launch_neck_topdown_bwd<float>(g_fine, g_coarse, out, N, Hc, Wc, C, s),
             launch_neck_topdown_bwd<bf16>(g_fine, g_coarse, out, N, Hc, Wc, C, s));
}
size_t lsam_neck_conv_split_image_bytes(int in_channels, int out_channels) { return neck_split_image_bytes(in_channels, out_channels); }
int lsam_neck_conv_pack_split(const float* weight, int in_channels, int out_channels, void* image, lsam_stream_t stream) {
    return launch_pack_neck_split(weight, in_channels, out_channels, image, static_cast<hipStream_t>(stream));
}
int lsam_neck_level_fwd_split(const float* x, const void* image, const float* bias, const float* top32, void* out, float* out32,
                              int N, int H, int W, int in_channels, int out_channels, lsam_stream_t stream) {
    return launch_neck_level_split(x, image, bias, top32, out, out32, N, H, W, in_channels, out_channels,
                                   static_cast<hipStream_t>(stream));
}

size_t lsam_kv_window_attention_image_bytes(int dim, int nh, int hd) {
    return kvattn_supported(8, 8, 8, 2, nh, hd, dim, nh * hd) ? kvattn_image_bytes() : 0;
}
int lsam_kv_window_attention_pack(const float* qkv_w, const float* qkv_b, int dim, int nh, int hd, void* image,
                                  lsam_stream_t stream) {
    return launch_pack_kvattn(qkv_w, qkv_b, dim, nh, hd, image, static_cast<hipStream_t>(stream));
}
int lsam_kv_window_attention_fwd(const void* xn, int ldx, const float* q_pooled, int ldq, const void* image, void* out, int ldo,
                                 int N, int H, int W, int dim, int nh, int hd, lsam_stream_t stream) {
    if (!kvattn_supported(H, W, 8, 2, nh, hd, dim, nh * hd)) {
        set_error("kv_window_attention: grid %dx%d, %d heads of %d over %d channels not supported", H, W, nh, hd, dim);
        return LSAM_ERR_UNSUPPORTED;
    }
    KvAttnArgs a{static_cast<const bf16*>(xn), ldx, q_pooled, ldq, static_cast<const char*>(image), static_cast<bf16*>(out), ldo,
                 N, H, W, dim, hd, (1.0f / sqrtf((float)hd)) * 1.4426950408889634f};
    return launch_kvattn(a, static_cast<hipStream_t>(stream));
}

size_t lsam_qkv_window4_attention_image_bytes(int dim, int nh, int hd, int pooled_q) {
    return qkvattn4_supported(4, 4, 4, pooled_q ? 2 : 1, nh, hd, dim, pooled_q ? nh * hd : dim) ? qkvattn4_image_bytes(nh, pooled_q != 0) : 0;
}
int lsam_qkv_window4_attention_pack(const float* qkv_w, const float* qkv_b, int dim, int nh, int hd, int pooled_q, void* image,
                                    lsam_stream_t stream) {
    return launch_pack_qkvattn4(qkv_w, qkv_b, dim, nh, hd, pooled_q != 0, image, static_cast<hipStream_t>(stream));
}
int lsam_qkv_window4_attention_fwd(const void* xn, int ldx, const float* q_pooled, int ldq, const void* image, void* out, int ldo,
                                   int N, int H, int W, int dim, int nh, int hd, lsam_stream_t stream) {
    QkvAttn4Args a{static_cast<const bf16*>(xn), ldx, q_pooled, ldq, static_cast<const char*>(image), static_cast<bf16*>(out), ldo,
                   N, H, W, dim, nh, hd, (1.0f / sqrtf((float)hd)) * 1.4426950408889634f};
    return launch_qkvattn4(a, static_cast<hipStream_t>(stream));
}

size_t lsam_window_block_image_bytes(int nh) { return (nh == 1 || nh == 2) ? winblock_image_bytes(nh) : 0; }
size_t lsam_window_block_bias_floats(int nh) { return (nh == 1 || nh == 2) ? winblock_bias_floats(nh) : 0; }
int lsam_window_block_pack(const float* qkv_w, const float* qkv_b, const float* proj_w, const float* proj_b, int nh,
                           int hd, void* image, float* bias_table, lsam_stream_t stream) {
    return launch_pack_winblock(qkv_w, qkv_b, proj_w, proj_b, nh, hd, image, bias_table, static_cast<hipStream_t>(stream));
}
int lsam_window_block_fwd(const float* x, float* y, const float* norm1_w, const float* norm1_b, const void* image,
                          const float* bias_table, int N, int H, int W, int nh, int hd, float eps,
                          lsam_stream_t stream) {
    LSAM_REQUIRE(hd > 0, LSAM_ERR_BAD_ARG, "window_block: head_dim must be positive");
    const WinBlockArgs a{x, y, norm1_w, norm1_b, image, bias_table, N, H, W, hd, eps,
                         (1.0f / sqrtf((float)hd)) * 1.4426950408889634f};
    return launch_winblock_attn(a, nh, static_cast<hipStream_t>(stream));
}

int lsam_window_block_attention_fwd(const float* x, void* a_out, const float* norm1_w, const float* norm1_b, const void* image,
                                    const float* bias_table, int N, int H, int W, int nh, int hd, float eps, lsam_stream_t stream) {
    LSAM_REQUIRE(hd > 0 && a_out, LSAM_ERR_BAD_ARG, "window_block_attention: head_dim must be positive, a_out not null");
    WinBlockArgs a{x, nullptr, norm1_w, norm1_b, image, bias_table, N, H, W, hd, eps, (1.0f / sqrtf((float)hd)) * 1.4426950408889634f};
    a.a_out = a_out;
    return launch_winblock_attn(a, nh, static_cast<hipStream_t>(stream));
}
size_t lsam_mlp_block_proj_image_bytes(int d) { return mlpblock_proj_supported(d) ? mlpblock_proj_image_bytes(d) : 0; }
int lsam_mlp_block_proj_pack(const float* proj_w, const float* proj_b, int d, void* proj_image, float* proj_bias, lsam_stream_t stream) {
    return launch_pack_mlpblock_proj(proj_w, proj_b, d, proj_image, proj_bias, static_cast<hipStream_t>(stream));
}
int lsam_mlp_block_fwd_proj(const float* x, const void* a_in, int lda, float* y, const float* norm2_w, const float* norm2_b,
                            const void* image, const float* bias_table, const void* proj_image, const float* proj_bias,
                            void* xn_next, int ldn, const float* next_norm_w, const float* next_norm_b, int M, int d, float eps,
                            lsam_stream_t stream) {
    LSAM_REQUIRE(mlpblock_proj_supported(d), LSAM_ERR_UNSUPPORTED, "mlp_block_fwd_proj: width %d not supported", d);
    LSAM_REQUIRE(a_in && proj_image && proj_bias, LSAM_ERR_BAD_ARG, "mlp_block_fwd_proj: null argument");
    MlpBlockArgs a{x, y, norm2_w, norm2_b, image, bias_table, M, d, mlpblock_chunks(d), eps};
    a.xn_next = xn_next;
    a.ldn = ldn;
    a.gamma_n = next_norm_w;
    a.beta_n = next_norm_b;
    a.a = a_in;
    a.lda = lda;
    a.pimg = proj_image;
    a.pbias = proj_bias;
    return launch_mlpblock(a, static_cast<hipStream_t>(stream));
}

size_t lsam_mlp_block_image_bytes(int d) { return mlpblock_supported(d) ? mlpblock_image_bytes(d) : 0; }
size_t lsam_mlp_block_bias_floats(int d) { return mlpblock_supported(d) ? mlpblock_bias_floats(d) : 0; }
int lsam_mlp_block_pack(const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b, int d,
                        void* image, float* bias_table, lsam_stream_t stream) {
    return launch_pack_mlpblock(fc1_w, fc1_b, fc2_w, fc2_b, d, image, bias_table, static_cast<hipStream_t>(stream));
}
int lsam_mlp_block_fwd(const float* x, float* y, const float* norm2_w, const float* norm2_b, const void* image,
                       const float* bias_table, int M, int d, float eps, lsam_stream_t stream) {
    LSAM_REQUIRE(mlpblock_supported(d), LSAM_ERR_UNSUPPORTED, "mlp_block: width %d not supported", d);
    const MlpBlockArgs a{x, y, norm2_w, norm2_b, image, bias_table, M, d, mlpblock_chunks(d), eps};
    return launch_mlpblock(a, static_cast<hipStream_t>(stream));
}

int lsam_mlp_block_fwd_next_norm(const float* x, float* y, const float* norm2_w, const float* norm2_b, const void* image,
                                 const float* bias_table, void* xn_next, int ldn, const float* next_norm_w,
                                 const float* next_norm_b, int M, int d, float eps, lsam_stream_t stream) {
    LSAM_REQUIRE(mlpblock_supported(d), LSAM_ERR_UNSUPPORTED, "mlp_block: width %d not supported", d);
    LSAM_REQUIRE(xn_next, LSAM_ERR_BAD_ARG, "mlp_block_fwd_next_norm: xn_next is null");
    MlpBlockArgs a{x, y, norm2_w, norm2_b, image, bias_table, M, d, mlpblock_chunks(d), eps};
    a.xn_next = xn_next;
    a.ldn = ldn;
    a.gamma_n = next_norm_w;
    a.beta_n = next_norm_b;
    return launch_mlpblock(a, static_cast<hipStream_t>(stream));
}

size_t lsam_tail_block_image_bytes(int d) { return tailblock_supported(d) ? tailblock_image_bytes(d) : 0; }
size_t lsam_tail_block_vec_floats(int d) { return tailblock_supported(d) ? tailblock_vec_floats(d) : 0; }
int lsam_tail_block_pack(const float* proj_w, const float* proj_b, const float* norm2_w, const float* norm2_b,
                         const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b, int d,
                         void* image, float* vec, lsam_stream_t stream) {
    return launch_pack_tailblock(proj_w, proj_b, norm2_w, norm2_b, fc1_w, fc1_b, fc2_w, fc2_b, d, image, vec,
                                 static_cast<hipStream_t>(stream));
}
int lsam_tail_block_fwd(const void* a, int lda, const float* res, float* out, const void* image, const float* vec,
                        void* xn_next, int ldn, const float* next_norm_w, const float* next_norm_b, int M, int d,
                        float eps, lsam_stream_t stream) {
    LSAM_REQUIRE(tailblock_supported(d), LSAM_ERR_UNSUPPORTED, "tail_block: width %d not supported", d);
    const TailBlockArgs t{a, lda, res, out, image, vec, xn_next, ldn, next_norm_w, next_norm_b, M, d,
                          tailblock_chunks(d), eps};
    return launch_tailblock(t, static_cast<hipStream_t>(stream));
}

int lsam_prof_arm(int capacity) { return prof_arm(capacity); }
int lsam_prof_read(int max, float* host_ms, double* host_flops, int* host_cfg, int* host_mnk) {
    LSAM_REQUIRE(host_ms && host_flops && host_cfg && max >= 0, LSAM_ERR_BAD_ARG, "prof_read: null argument");
    return prof_read(max, host_ms, host_flops, host_cfg, host_mnk);
}

int lsam_prof_bracket_cost(lsam_stream_t stream, int pairs, float* host_avg_ms) {
    LSAM_REQUIRE(host_avg_ms && pairs > 0 && pairs <= 4096, LSAM_ERR_BAD_ARG, "prof_bracket_cost: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    LSAM_HIP(hipEventCreateWithFlags(&e0, hipEventDisableSystemFence));
    LSAM_HIP(hipEventCreateWithFlags(&e1, hipEventDisableSystemFence));
    double sum = 0.0;
    int rc = LSAM_OK;
    for (int i = 0; i < pairs && rc == LSAM_OK; ++i) {
        float ms = 0.f;
        if (hipEventRecord(e0, s) != hipSuccess || hipEventRecord(e1, s) != hipSuccess ||
            hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
            rc = LSAM_ERR_HIP;
        sum += ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    LSAM_REQUIRE(rc == LSAM_OK, LSAM_ERR_HIP, "prof_bracket_cost: event call failed");
    *host_avg_ms = (float)(sum / pairs);
    return LSAM_OK;
}

size_t lsam_outer_reduce_ws(int M, int C) { return (M > 0 && C > 0) ? outer_reduce_ws(M, C) : 0; }

int lsam_outer_reduce(int dtype, const void* X, int ldx, const void* Y, int ldy, int M, int C, int R, float* out,
                      long out_cs, long out_ks, void* partial, lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH(dtype,
             launch_outer_reduce<float>(static_cast<const float*>(X), ldx, static_cast<const float*>(Y), ldy, M, C, R,
                                        out, out_cs, out_ks, partial, s),
             launch_outer_reduce<bf16>(static_cast<const bf16*>(X), ldx, static_cast<const bf16*>(Y), ldy, M, C, R, out,
                                       out_cs, out_ks, partial, s));
}

size_t lsam_outer_reduce_static_max_ws(void) { return outer_reduce_static_max_bytes(); }
int lsam_outer_reduce_static_init(void) { return outer_reduce_static_init(); }
int lsam_outer_reduce_static(int dtype, const void* X, int ldx, const void* Y, int ldy, int M, int C, int R, float* out,
                             long out_cs, long out_ks, lsam_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    LSAM_REQUIRE(X && Y && out, LSAM_ERR_BAD_ARG, "outer_reduce_static: null operand");
    DISPATCH(dtype,
             launch_outer_reduce_static<float>(static_cast<const float*>(X), ldx, static_cast<const float*>(Y), ldy, M, C,
                                               R, out, out_cs, out_ks, s),
             launch_outer_reduce_static<bf16>(static_cast<const bf16*>(X), ldx, static_cast<const bf16*>(Y), ldy, M, C, R,
                                              out, out_cs, out_ks, s));
}

}  // extern "C"
