/* lsam.h -- C ABI of liblsam.so: MI355X (gfx950) kernels for the LoRA-SAM2 Hiera
 * image-encoder hot path.
 *
 * The reference (DonWick32/image-segmentation) has NO FFI on this path: it is pure
 * PyTorch behind nn.Module interfaces.  These entry points are what
 * this repo's drop-in modules (the .py files of image_segmentation_amd) bind through ctypes in
 * place of the torch ops the reference modules call; each cites the reference lines
 * it replaces.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions: plain pointers and sizes only; all pointers are DEVICE pointers unless
 * named host_*; every function returns LSAM_OK (0) or a negative error code and
 * never throws, allocates device memory, or synchronises the stream.  The message
 * for the calling thread's last error is lsam_last_error().  Work is enqueued on the
 * given hipStream_t (pass torch.cuda.current_stream().cuda_stream).  Functions are
 * re-entrant across threads (autograd-engine thread vs. main thread) as long as
 * two calls do not share a workspace.
 *
 * dtype: LSAM_F32 computes in exact-fp32 MFMA (v_mfma_f32_16x16x4_f32); LSAM_BF16
 * computes in bf16 MFMA with fp32 accumulation.  "T" below means float or bf16
 * storage accordingly.  The residual stream, LayerNorm statistics, biases, LSE and all
 * LoRA gradients are fp32 in both modes.
 */
#ifndef LSAM_H
#define LSAM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lsam_stream_t; /* hipStream_t */

enum { LSAM_F32 = 0, LSAM_BF16 = 1 };
enum {
    LSAM_OK = 0,
    LSAM_ERR_BAD_ARG = -1,     /* null / misaligned pointer, bad size */
    LSAM_ERR_UNSUPPORTED = -2, /* shape the kernels do not cover */
    LSAM_ERR_HIP = -3,         /* a HIP runtime call failed */
    LSAM_ERR_WORKSPACE = -4    /* caller workspace too small */
};

const char* lsam_last_error(void);
int lsam_version(void);

/* Measurement aid (no counterpart in the reference; used by bench.py for the live roofline figure):
 * lsam_prof_arm(n) brackets each of the next n lsam_linear-kernel launches (from any entry point)
 * with a hipEvent pair on the launch stream; lsam_prof_read disarms, waits for the events and
 * returns per-launch milliseconds, 2*M*N*K, and a kernel id (1000*sizeof(T) + tile width) into
 * HOST arrays; returns the count.  Process-global, not thread-safe, off by default. */
int lsam_prof_arm(int capacity);
int lsam_prof_read(int max, float* host_ms, double* host_flops, int* host_cfg, int* host_mnk /* nullable [3*max] */);
/* Calibration of the above (no counterpart in the reference): the average time between the two events of `pairs` EMPTY
 * brackets on `stream` (which it synchronises).  An event pair costs the stream this much even around nothing, so a
 * bracketed launch's duration over-states the kernel by about it; bench.py reports the sum of brackets both raw and
 * with this subtracted per launch. */
int lsam_prof_bracket_cost(lsam_stream_t stream, int pairs, float* host_avg_ms);

/* ---- epilogue flags for lsam_linear ---- */
enum {
    LSAM_EPI_GELU = 1,      /* out = gelu(acc + bias)           (sam2_utils.py:133, nn.GELU exact erf) */
    LSAM_EPI_GELU_BWD = 2,  /* out = acc * gelu'(aux)           (autograd of the same)                  */
    LSAM_EPI_OUT_F32 = 4,   /* out is float* even in bf16 mode                                         */
    LSAM_EPI_SAVE_PRE = 8,  /* also write (acc + bias) before the activation to out2 (T)                */
    LSAM_EPI_RELU = 16,     /* out = max(acc + bias, 0)         (sam/transformer.py MLPBlock, sam2_utils.MLP of the SAM heads) */
    LSAM_EPI_SIGMOID = 32,  /* out = sigmoid(acc + bias)        (MLP(sigmoid_output=True): the IoU head, mask_decoder.py:96-102) */
    LSAM_EPI_FEW_ROWS = 64  /* caller hint: A is a short stream (the SAM heads' 8 tokens per frame): run the streaming kernel, K split
                             * over a workgroup's waves.  Same result up to fp32 summation order; the order never depends on M, so
                             * results stay independent of the batch size.  Only with bias / RELU / SIGMOID / OUT_F32 / res.       */
};

/* out[M,N] = epilogue( A[M,K] . W[N,K]^T + bias[N] ) + res
 * Replaces nn.Linear -> addmm (hieradet.py:59,79,132; sam2_utils.py:133) with the bias,
 * GELU and residual add (hieradet.py:163,165) fused into the MFMA epilogue.
 * A, W are T with leading dimensions lda, ldw (elements; rows 16-byte aligned, K % 16 == 0);
 * bias (nullable) fp32; res (nullable) fp32 [M or res_row_mod rows][ldres], row index
 * m % res_row_mod when res_row_mod > 0 (used for the positional embedding, hieradet.py:312);
 * aux (GELU_BWD) is T [M][ldaux]; out is T (or fp32 with LSAM_EPI_OUT_F32) [M][ldo];
 * out2 (SAVE_PRE) is T [M][ldo2].  N % 8 == 0.  In-place res == out is allowed. */
int lsam_linear(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias,
                const float* res, int ldres, int res_row_mod, const void* aux, int ldaux,
                void* out, int ldo, void* out2, int ldo2, int flags, int M, int N, int K,
                lsam_stream_t stream);

/* dst[r][c] = (T) src[r*src_rs + c*src_cs] for r<rows, c<cols; zero for cols <= c < ldd.
 * Used to pack frozen fp32 weights (and their transposes for the backward) once. */
int lsam_pack_matrix(int dtype, const float* src, long src_rs, long src_cs, int rows, int cols,
                     void* dst, int ldd, lsam_stream_t stream);

/* y[m, 0:C] = (T) LayerNorm(x[m, 0:C]) * gamma + beta   (nn.LayerNorm(eps=1e-6), hieradet.py:100,136,165)
 * x fp32 [M][ldx]; y T [M][ldy]; stats (nullable) fp32 [M][2] = (mean, rstd). C % 4 == 0. */
int lsam_layernorm_fwd(int dtype, const float* x, int ldx, const float* gamma, const float* beta,
                       void* y, int ldy, float* stats, int M, int C, float eps, lsam_stream_t stream);

/* dx[m,:] = dres[m,:] + LN'(dy[m,:]) ; optional T copy of dx.  Only the input gradient: the
 * affine parameters are frozen (lora_qkv.py:159-160).  dy, dres(nullable), x, dx fp32. */
int lsam_layernorm_bwd(int dtype, const float* dy, int lddy, const float* x, int ldx,
                       const float* stats, const float* gamma, const float* dres, int lddres,
                       float* dx, int lddx, void* dx_t, int lddxt, int M, int C,
                       lsam_stream_t stream);

/* Windowed / global multi-head attention with Hiera q-pooling, reading the qkv GEMM output in
 * natural token order and writing the un-partitioned, cropped result.  Replaces
 * window_partition + reshape/unbind + do_pool(q) + F.scaled_dot_product_attention + transpose
 * + window_unpartition (backbones/utils.py:16-60, hieradet.py:59-77,142-161).
 *   qkv  T [N,H,W, 3*nh*hd]  channel order [q|k|v][head][hd]  (hieradet.py:59-61)
 *   bias fp32 [3*nh*hd]: value of qkv on the zero-pad tokens, which take part as keys
 *        (padding happens after LN1, hieradet.py:136,146)
 *   bias_t (nullable) the same bias converted to T; when given (bf16), windows with >= 128 keys and
 *        global attention use the LDS-DMA kernel, which fetches pad-token k/v from it
 *   ws   window size, 0 = global;  pool 1 or 2 (2x2 max over q inside each window)
 *   out  T [N, H/pool, W/pool, nh*hd];  lse (nullable) fp32 [N*(H/pool)*(W/pool)][nh]
 * hd % 8 == 0, hd <= 128 (bf16 windows of 8x8 .. 14x14 and global attention with hd <= 64 take the MFMA/LDS-DMA
 * kernels; everything else the generic tiled kernel). */
int lsam_attention_fwd(int dtype, const void* qkv, int ldqkv, const float* bias, const void* bias_t,
                       void* out, int ldo, float* lse, int N, int H, int W, int ws, int pool, int nh,
                       int hd, lsam_stream_t stream);
/* lsam_attention_fwd with pool = 2 and the pooled queries supplied: q_pooled fp32 [N*(H/2)*(W/2)][ld_qp] holds
 * do_pool(q) (hieradet.py:25-36,69-71) in output-token order, e.g. the q columns of the qkv projection taken
 * through lsam_linear_maxpool2x2; the q columns of qkv are then never read (they need not have been written).
 * Same result bit for bit: the max of rounded values is the rounded max. */
int lsam_attention_fwd_pooled_q(int dtype, const void* qkv, int ldqkv, const float* bias, const void* bias_t,
                                const float* q_pooled, int ld_qp, void* out, int ldo, float* lse, int N, int H, int W,
                                int ws, int nh, int hd, lsam_stream_t stream);

/* Backward of the above for the LoRA blocks: windows (ws > 0) with at most 64 queries each and any number of keys,
 * hd <= 96 (autograd of F.scaled_dot_product_attention, hieradet.py:70-74, as core_loss.backward() walks it):
 *   dout T [N,H,W,nh*hd] -> dqkv T [N,H,W, lddqkv] (q|k|v gradients of the real tokens; the
 *   zero-pad tokens' k/v are constants and get no gradient).  Every q, k, v element of a real token is written. */
int lsam_attention_bwd(int dtype, const void* qkv, int ldqkv, const float* bias, const void* dout,
                       int lddo, const float* lse, void* dqkv, int lddqkv, int N, int H, int W,
                       int ws, int nh, int hd, lsam_stream_t stream);
/* The same with Hiera's q-pool (pool = 2; pool = 1 is lsam_attention_bwd): dout T [N,H/2,W/2,nh*hd] and
 * lse [N*(H/2)*(W/2)][nh] are indexed by the pooled tokens, qkv / dqkv by the input tokens.  The gradient of a pooled
 * query goes, per channel, to the first maximal token of its 2x2 group (autograd of nn.MaxPool2d in do_pool,
 * hieradet.py:25-36,64-67); the other three receive 0.  (ws/2)^2 <= 64, ws and the grid even. */
int lsam_attention_bwd_qpool(int dtype, const void* qkv, int ldqkv, const float* bias, const void* dout,
                             int lddo, const float* lse, void* dqkv, int lddqkv, int N, int H, int W,
                             int ws, int pool, int nh, int hd, lsam_stream_t stream);

/* out[n,h/2,w/2,:] = max over the 2x2 block (do_pool on the shortcut, hieradet.py:140). fp32 NHWC. */
int lsam_maxpool2x2(const float* x, float* out, int N, int H, int W, int C, lsam_stream_t stream);

/* out[n][c] = mean over the HW tokens of x[n][hw][c] (fp32 NHWC): the per-frame pooled embedding of the
 * feature-extraction pass over Hiera.get_embedding's last stage (hieradet.py:284-302), the quantity the
 * reference's nearest-domain LoRA selection compares (train_cl_kmean.py:189-207). Deterministic. */
int lsam_spatial_mean(const float* x, float* out, int N, int HW, int C, lsam_stream_t stream);

/* fp32 -> T copy of a [rows][cols] matrix (ld in elements). */
int lsam_cast(int dtype, const float* src, int lds, void* dst, int ldd, int rows, int cols,
              lsam_stream_t stream);

/* im2col for PatchEmbed's 7x7 stride-4 pad-3 conv (backbones/utils.py:85-93):
 * img fp32 [N,3,S,S] -> cols T [N*(S/4)^2][ldc], k = c*49 + ky*7 + kx, zero beyond 147. */
int lsam_im2col_patch(int dtype, const float* img, void* cols, int ldc, int N, int S,
                      lsam_stream_t stream);

/* Lateral 1x1 convolution of the FPN neck straight from an fp32 stage output, bf16 MFMA, bf16 out:
 *   out[m][:] = x[m][:] . weight^T + bias      (FpnNeck.forward, image_encoder.py:107-110, kernel_size 1;
 *   x = the NHWC tokens of a trunk stage, fp32 [M][in_channels]; weight fp32 [out_channels][in_channels]).
 * Replaces the fp32 -> bf16 copy of x plus lsam_linear for the high-resolution levels.
 * out_channels == 256 and in_channels % 8 == 0, 16 <= in_channels <= 224; otherwise 0 bytes / LSAM_ERR_UNSUPPORTED. */
size_t lsam_lateral_conv_image_bytes(int in_channels, int out_channels);
int lsam_lateral_conv_pack(const float* weight, int in_channels, int out_channels, void* image, lsam_stream_t stream);
int lsam_lateral_conv_fwd(const float* x, const void* image, const float* bias, void* out, int M, int in_channels,
                          int out_channels, lsam_stream_t stream);

/* A Linear with a SMALL input width and LDS-resident weights, bf16 MFMA, bf16 in / bf16 out:
 *   out[m][:out_features] = A[m][:in_features] . weight^T + bias
 * (the qkv Linear of MultiScaleAttention, hieradet.py:59, in the stages where dim <= 224 -- Hiera-B+ stage 2, Hiera-L stage 1 --
 * and the k | v rows of it in a q-pool block whose queries come from lsam_linear_maxpool2x2, hieradet.py:64-67).
 * For such widths the tiled lsam_linear is all epilogue (3.5 k-tiles in front of 128 store instructions per output tile);
 * here each 224-column group of the weight stays in LDS and every wave streams its own 64-row groups with no barrier.
 * weight fp32 [out_features][in_features] row-major; A bf16 [M][lda]; out bf16 [M][ldo]; lda, ldo multiples of 8.
 * in_features % 8 == 0, 16 <= in_features <= 224, out_features % 32 == 0; otherwise 0 bytes / LSAM_ERR_UNSUPPORTED. */
size_t lsam_stream_linear_image_bytes(int in_features, int out_features);
int lsam_stream_linear_pack(const float* weight, int in_features, int out_features, void* image, lsam_stream_t stream);
int lsam_stream_linear_fwd(const void* A, int lda, const void* image, const float* bias, void* out, int ldo, int M,
                           int in_features, int out_features, lsam_stream_t stream);

/* One FPN level in one kernel, any level and both dtypes (FpnNeck.forward, image_encoder.py:102-134):
 *   out[n][y][x][:] = T( Conv2d(in_channels, out_channels, 1)(x)[n, :, y, x]  (+ top[n][y/2][x/2][:]) )
 * = the lateral convolution `self.convs[n - i](x)` (:112-114) with, when top != NULL, the nearest-x2 top-down add
 * `lateral_features + F.interpolate(prev_features.to(float32), scale_factor=2.0, mode="nearest")` (:116-125) in its
 * epilogue.  x fp32 NHWC [N,H,W,in_channels] (a trunk stage output), top / out T NHWC ([N,H/2,W/2,out_channels] /
 * [N,H,W,out_channels]); image from lsam_neck_conv_pack(weight fp32 [out_channels][in_channels]).
 * out_channels in {256, 64, 32} (d_model, and the 64 / 32 outputs of mask_decoder.conv_s1 / conv_s0 composed with the
 * lateral convolution: two 1x1 convolutions with nothing between them are one, sam2_base.py:473-478), in_channels % 8 == 0.
 * lsam_neck_topdown_bwd is the adjoint of the upsample for the backward: out[n][y][x][:] = sum of g_fine over the 2x2
 * children (+ g_coarse[n][y][x][:] when given), all T NHWC; the input gradient of the level is then one lsam_linear. */
size_t lsam_neck_conv_image_bytes(int dtype, int in_channels, int out_channels);
int lsam_neck_conv_pack(int dtype, const float* weight, int in_channels, int out_channels, void* image, lsam_stream_t stream);
int lsam_neck_level_fwd(int dtype, const float* x, const void* image, const float* bias, const void* top, void* out, int N,
                        int H, int W, int in_channels, int out_channels, lsam_stream_t stream);
int lsam_neck_topdown_bwd(int dtype, const void* g_fine, const void* g_coarse, void* out, int N, int Hc, int Wc, int C,
                          lsam_stream_t stream);

/* The same level (image_encoder.py:102-134) with fp32-ACCURATE products on the bf16 matrix pipe, for the levels of the
 * top-down chain under bf16 autocast: operands are split in two bf16 terms (x = xh + xl, W = Wh + Wl; three MFMAs per
 * step), the top-down addend `top32` and the optional second output `out32` (the next finer level's addend) are fp32, and
 * `out` (bf16 NHWC) is rounded once.  The reference's autocast convolution rounds inputs, weights and output; this level
 * feeds the SAM transformer, and the extra roundings cost 0.4 points of mask IoU against the reference's fp32 masks.
 * out_channels == 256 only; image from lsam_neck_conv_pack_split (lsam_neck_conv_split_image_bytes bytes). */
size_t lsam_neck_conv_split_image_bytes(int in_channels, int out_channels);
int lsam_neck_conv_pack_split(const float* weight, int in_channels, int out_channels, void* image, lsam_stream_t stream);
int lsam_neck_level_fwd_split(const float* x, const void* image, const float* bias, const float* top32, void* out, float* out32,
                              int N, int H, int W, int in_channels, int out_channels, lsam_stream_t stream);

/* Attention half of a q-pool stage-entry block with 8x8 windows (MultiScaleAttention.forward, hieradet.py:56-81, with
 * q = do_pool(q) :69-71; Hiera-B+ block 2: 112 -> 224 channels, 4 heads of 56) in one kernel per window: the k | v
 * projection of the window's 64 LN1 rows and softmax(q k^T / sqrt(hd)) v for its 16 pooled queries.  k | v (the largest
 * tensor of the trunk: 940 MB per 16-frame batch) never exist in memory.  xn bf16 [N*H*W][ldx] (norm1 output), q_pooled
 * fp32 [N*(H/2)*(W/2)][ldq] = do_pool(q) as lsam_linear_maxpool2x2 produces it, out bf16 [N*(H/2)*(W/2)][ldo] in the
 * reference's head-major channel order.  image: lsam_kv_window_attention_pack(qkv.weight [3*nh*hd][dim], qkv.bias).
 * bf16, nh == 4, hd % 8 == 0 and <= 64, dim % 8 == 0 and <= 128, H and W multiples of 8 (image_bytes returns 0 otherwise). */
size_t lsam_kv_window_attention_image_bytes(int dim, int nh, int hd);
int lsam_kv_window_attention_pack(const float* qkv_w, const float* qkv_b, int dim, int nh, int hd, void* image,
                                  lsam_stream_t stream);
int lsam_kv_window_attention_fwd(const void* xn, int ldx, const float* q_pooled, int ldq, const void* image, void* out, int ldo,
                                 int N, int H, int W, int dim, int nh, int hd, lsam_stream_t stream);

/* Attention half of a block with 4x4 windows (MultiScaleAttention.forward, hieradet.py:56-81; Hiera-B+ stage 2: 224
 * channels, 4 heads of 56) in one kernel: qkv = norm1(x) . W^T + b (hieradet.py:59) for the 16 tokens of a window and
 * softmax(q k^T / sqrt(hd)) v per head (:73-77); qkv (352 MB per 16-frame batch and block) never exists in memory.
 * q_pooled == NULL: dim == nh * hd, out bf16 [N*H*W][ldo].  q_pooled != NULL: the q-pool stage-entry block behind them
 * (q = do_pool(q), :69-71; B+ block 5: 224 -> 448 channels, 8 heads): k | v projected here, q_pooled fp32
 * [N*(H/2)*(W/2)][ldq] as lsam_linear_maxpool2x2 produces it, out bf16 [N*(H/2)*(W/2)][ldo].  xn bf16 [N*H*W][ldx] (norm1
 * output).  image: lsam_qkv_window4_attention_pack(qkv.weight [3*nh*hd][dim], qkv.bias) with the same pooled_q flag.
 * bf16, hd % 8 == 0 and <= 64, dim % 8 == 0 and <= 224, H and W multiples of 4 (image_bytes returns 0 otherwise). */
size_t lsam_qkv_window4_attention_image_bytes(int dim, int nh, int hd, int pooled_q);
int lsam_qkv_window4_attention_pack(const float* qkv_w, const float* qkv_b, int dim, int nh, int hd, int pooled_q, void* image,
                                    lsam_stream_t stream);
int lsam_qkv_window4_attention_fwd(const void* xn, int ldx, const float* q_pooled, int ldq, const void* image, void* out, int ldo,
                                   int N, int H, int W, int dim, int nh, int hd, lsam_stream_t stream);

/* ---- SAM prompt encoder / mask decoder pieces (SAM2Base._forward_sam_heads, sam2_base.py:257-413) that are not a
 * Linear or a LayerNorm.  See image_segmentation_amd/sam_heads.py for the composition. ---- */
/* out[m][c] = (T_out)(a[m][c] + b[m % b_rows][c]); b nullable (then a dtype conversion).  `queries + query_pe`,
 * `keys + key_pe` (sam/transformer.py:88-101, 186-209), `+ no_mem_embed` (sam2_base.py:669-674), `+ dense prompt` (mask_decoder.py:191). */
int lsam_add_rows(int a_dtype, const void* a, int lda, const float* b, int ldb, int b_rows, int out_dtype, void* out, int ldo, int M,
                  int C, lsam_stream_t stream);
/* F.scaled_dot_product_attention of sam/transformer.py:233-265 on separate q, k, v (cross attention):
 * q T [B][nq][ldq], k / v T [B][nk][ld], heads of hd in {16, 32, 64} side by side in a row; batch strides in elements. */
int lsam_mha_fwd(int dtype, const void* q, int ldq, long q_batch_stride, const void* k, int ldk, long k_batch_stride, const void* v,
                 int ldv, long v_batch_stride, void* out, int ldo, long out_batch_stride, int B, int nq, int nk, int nh, int hd,
                 lsam_stream_t stream);
/* PositionEmbeddingRandom._pe_encoding (position_encoding.py:147-155): out[m] = [sin(t) | cos(t)], t = 2 pi (2 c - 1) . gauss;
 * gauss fp32 [2][F], coords01 fp32 [M][2] in [0, 1], out fp32 [M][2F]. */
int lsam_pe_encode(const float* gauss, const float* coords01, float* out, int M, int F, lsam_stream_t stream);
/* Tail of a ConvTranspose2d(kernel 2, stride 2) computed as g = x . Wr^T + b with lsam_linear (g fp32 [B*H*W][4*C], columns
 * (dy, dx, co)): out[b][2y+dy][2x+dx][:] = GELU(LayerNorm2d?(g[b,y,x][(dy,dx),:] + skip[b][2y+dy][2x+dx][:])), T NHWC;
 * ln_w / ln_b NULL = no LayerNorm2d (mask_decoder.py:216-225, sam2_utils.py:139-153).  C in {32, 64}. */
int lsam_upscale2x2_fwd(int dtype, const float* g, const void* skip, const float* ln_w, const float* ln_b, float eps, void* out, int B,
                        int H, int W, int C, lsam_stream_t stream);
/* masks[b][j][p] = obj_score[b] > 0 ? logits[b][p][first + j] : no_obj_score   (sam2_base.py:359-368); logits fp32 [B][HW][ldl]. */
int lsam_mask_gate(const float* logits, int ldl, const float* obj_score, float* out, int B, int HW, int first, int n_masks,
                   float no_obj_score, lsam_stream_t stream);
/* F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=False) on fp32 [BC][H][W] (sam2_base.py:372-377). */
int lsam_bilinear_upsample(const float* in, float* out, int BC, int H, int W, int scale, lsam_stream_t stream);

/* ---- input gradients of the SAM heads and the mask losses.  The heads' weights are frozen (train_lora_sam.py:116-117), so --
 * as in the trunk -- there are no weight-gradient GEMMs: Linear / ConvTranspose2d backward is lsam_linear on packed
 * transposed weights, LayerNorm backward is lsam_layernorm_bwd, and the decoder-side LoRA factors (lora_qkv.py:11-67) get
 * their gradients from lsam_outer_reduce.  These are the remaining pieces (autograd of sam/transformer.py:233-265,
 * mask_decoder.py:161-250, sam2_base.py:359-380).  All fp32 gradients, deterministic (no atomics). */
/* dx = dy * act'(.) computed from the activation's OUTPUT y (T): kind 0 = ReLU, 1 = sigmoid.  n % 4 == 0. */
int lsam_act_bwd(int kind, const float* dy, int y_dtype, const void* y, float* dx, long n, lsam_stream_t stream);
/* Backward of lsam_mha_fwd on contiguous [B][n][nh*hd] operands (q, k, v, out: T; dout, dq, dk, dv: fp32).
 * hd in {16, 32}; min(nq, nk) <= 16 (the two-way transformer's shapes: 8 x 8, 8 x 4096, 4096 x 8). */
int lsam_mha_bwd(int dtype, const void* q, const void* k, const void* v, const void* out, const float* dout, float* dq, float* dk,
                 float* dv, int B, int nq, int nk, int nh, int hd, lsam_stream_t stream);
/* Backward of lsam_upscale2x2_fwd with respect to g (fp32 [B*H*W][4*C]); dout fp32 [B][2H][2W][C]; the pre-activation is
 * recomputed from g (+ skip, T, nullable).  The skip connection gets no gradient (it comes from frozen, LoRA-free stages). */
int lsam_upscale2x2_bwd(int dtype, const float* dout, const float* g, const void* skip, const float* ln_w, const float* ln_b, float eps,
                        float* dg, int B, int H, int W, int C, lsam_stream_t stream);
/* Adjoints of lsam_mask_gate (d_logits fp32 [B][HW][ldl], zero outside the selected columns and for gated frames) and of
 * lsam_bilinear_upsample. */
int lsam_mask_gate_bwd(const float* d_masks, const float* obj_score, float* d_logits, int ldl, int B, int HW, int first, int n_masks,
                       lsam_stream_t stream);
int lsam_bilinear_upsample_bwd(const float* dout, float* din, int BC, int H, int W, int scale, lsam_stream_t stream);
/* Mask losses (training/loss_fns.py:19-129): one pass over logits fp32 [N][M][HW] and targets fp32 [N][HW] (broadcast over
 * the M candidates, :240) gives sums fp32 [N][M][6] = { sum sigmoid_focal_loss elements (alpha, gamma; alpha < 0: no alpha
 * weighting), sum p t, sum p, sum t, |x > 0 and t > 0|, |x > 0 or t > 0| }, from which the focal mean, the dice ratio and the
 * actual IoU of every candidate follow.  lsam_mask_loss_bwd takes coef fp32 [N][M][3] = d loss / d sums[., 0..2] and writes
 * d_logits = coef0 focal'(x) + (coef1 t + coef2) p (1 - p)  (zeros for candidates whose coefficients are all zero). */
/* out[b][l][c] = sum_m G[b][m][l] U[b][m][c] (d hyper_in = d_masks . upscaled^T, the weight-like side of mask_decoder.py:238-239's
 * product): G fp32 [B][M][L], U T [B][M][C], out fp32 [B][L][C]; L <= 16, C <= 64; workspace of lsam_batched_outer_ws_floats floats. */
size_t lsam_batched_outer_ws_floats(int B, int L, int C);
int lsam_batched_outer(int u_dtype, const float* G, const void* U, float* out, float* workspace, int B, int M, int L, int C,
                       lsam_stream_t stream);
size_t lsam_mask_loss_ws_floats(int N, int M);   /* floats of `workspace` (per-slice partial sums, summed in a fixed order) */
int lsam_mask_loss_sums(const float* logits, const float* targets, float* sums, float* workspace, int N, int M, int HW, float alpha,
                        float gamma, lsam_stream_t stream);
int lsam_mask_loss_bwd(const float* logits, const float* targets, const float* coef, float* d_logits, int N, int M, int HW, float alpha,
                       float gamma, lsam_stream_t stream);

/* Shortcut of a q-pool block in one kernel: out = MaxPool2d(2, 2)(A . W^T + bias) over the token grid
 * (`shortcut = do_pool(self.proj(x_norm), self.pool)`, hieradet.py:139-140; do_pool: hieradet.py:27-41).
 * A bf16 [N_img*H*W][lda] (rows = NHWC tokens), W bf16 [N][ldw], bias fp32 [N]; out fp32 [N_img*(H/2)*(W/2)][ldo].
 * The full-resolution product never exists in memory: the pool is taken over accumulator tiles in the epilogue.
 * Needs W % 32 == 0, whole 256-token patches (min(W,128) wide) and N % 224 == 0, else LSAM_ERR_UNSUPPORTED
 * (callers then use lsam_linear + lsam_maxpool2x2). */
int lsam_linear_maxpool2x2(const void* A, int lda, const void* W, int ldw, const float* bias, float* out, int ldo,
                           int N_img, int H, int Wd, int N, int K, lsam_stream_t stream);

/* PatchEmbed + position embedding in one kernel (bf16 MFMA, fp32 out), no column matrix in memory:
 *   out[n][oy][ox][:] = Conv2d(3, E, kernel 7, stride 4, padding 3)(frames)[n, :, oy, ox] + proj_b + pos[oy][ox][:]
 * (PatchEmbed.forward, backbones/utils.py:85-93, incl. its NCHW -> NHWC permute, and the `x = x + self._get_pos_embed(...)`
 * of Hiera.forward, hieradet.py:307-312; pos is that embedding, fp32 [(S/4)^2][E]).
 * frames fp32 [N,3,S,S], S % 4 == 0; E % 16 == 0, E <= 144 (0 bytes / LSAM_ERR_UNSUPPORTED otherwise).
 * lsam_patch_embed_pack builds the weight image (lsam_patch_embed_image_bytes bytes) from proj.weight [E][3][7][7]. */
size_t lsam_patch_embed_image_bytes(int embed_dim);
int lsam_patch_embed_pack(const float* proj_w, int embed_dim, void* image, lsam_stream_t stream);
int lsam_patch_embed_fwd(const float* frames, const void* image, const float* proj_b, const float* pos, float* out, int N,
                         int S, int embed_dim, lsam_stream_t stream);

/* G[c][k] = sum_m X[m][c] * Y[m][k]  (k < R <= 16): the LoRA dA / dB token reductions
 * (autograd of lora_qkv.py:104-105), deterministic two-pass split over m.
 * X T [M][ldx] (C columns), Y T [M][ldy]; out fp32 at out[c*out_cs + k*out_ks];
 * partial: fp32 workspace of lsam_outer_reduce_ws(M, C) bytes. */
size_t lsam_outer_reduce_ws(int M, int C);
int lsam_outer_reduce(int dtype, const void* X, int ldx, const void* Y, int ldy, int M, int C, int R,
                      float* out, long out_cs, long out_ks, void* partial, lsam_stream_t stream);

/* lsam_outer_reduce WITHOUT a caller workspace and without any allocation.  No counterpart in the reference: it
 * exists for callers that cannot allocate device memory (code under HIP-graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  The fp32 partial sums of pass 1 live in a static device array inside liblsam.so
 * (lsam_outer_reduce_static_max_ws() bytes = 8 MiB, held on every GPU the library runs on until the process ends);
 * M, C with lsam_outer_reduce_ws(M, C) beyond that are rejected with LSAM_ERR_WORKSPACE (the LoRA blocks of the
 * path, M = 16 frames x 1024 tokens, C = 896, need 7.0 MiB).  Same kernels, same result bit for bit as
 * lsam_outer_reduce.  The array's per-device address is resolved once per device (hipGetSymbolAddress) on the
 * first call there, or by lsam_outer_reduce_static_init() with that device current -- call that before a stream
 * capture begins.  All calls on one GPU share the array: calls that overlap there (different streams) race. */
size_t lsam_outer_reduce_static_max_ws(void);
int lsam_outer_reduce_static_init(void);
int lsam_outer_reduce_static(int dtype, const void* X, int ldx, const void* Y, int ldy, int M, int C, int R,
                             float* out, long out_cs, long out_ks, lsam_stream_t stream);

/* Fused attention half of a same-width 8x8-window block, bf16 MFMA, in one kernel:
 *   y = x + proj(attn(qkv(LayerNorm1(x))))       (MultiScaleBlock.forward, hieradet.py:134-163 with
 *   window_partition/unpartition, backbones/utils.py:16-60, and MultiScaleAttention.forward,
 *   hieradet.py:56-81, for dim == dim_out, window 8, no q-pool, grid a multiple of 8: the stage-1 blocks).
 * x, y fp32 [N*H*W][nh*hd] token-major NHWC (y may alias x); nh in {1, 2}; hd <= 64, hd % 8 == 0.
 * lsam_window_block_pack builds the weight image (lsam_window_block_image_bytes bytes) and the
 * padded bias table (lsam_window_block_bias_floats floats) from the fp32 module parameters
 * (attn.qkv.weight [3d][d], attn.qkv.bias, attn.proj.weight [d][d], attn.proj.bias). */
size_t lsam_window_block_image_bytes(int nh);
size_t lsam_window_block_bias_floats(int nh);
int lsam_window_block_pack(const float* qkv_w, const float* qkv_b, const float* proj_w, const float* proj_b,
                           int nh, int hd, void* image, float* bias_table, lsam_stream_t stream);
int lsam_window_block_fwd(const float* x, float* y, const float* norm1_w, const float* norm1_b,
                          const void* image, const float* bias_table, int N, int H, int W, int nh, int hd,
                          float eps, lsam_stream_t stream);

/* Fused MLP half of a block, bf16 MFMA, in one kernel:
 *   y = x + fc2(GELU(fc1(LayerNorm2(x))))         (hieradet.py:165, MLP: sam2_utils.py:112-136)
 * x, y fp32 [M][d] (y may alias x); d % 8 == 0 and d <= 112 or 128 < d <= 224 (the stage-1/2 widths).
 * lsam_mlp_block_pack builds the chunked weight image (lsam_mlp_block_image_bytes(d) bytes) and the
 * fc2 bias table (lsam_mlp_block_bias_floats(d) floats) from mlp.layers.0.{weight [4d][d], bias} and
 * mlp.layers.1.{weight [d][4d], bias}. */
size_t lsam_mlp_block_image_bytes(int d);
size_t lsam_mlp_block_bias_floats(int d);
int lsam_mlp_block_pack(const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b, int d,
                        void* image, float* bias_table, lsam_stream_t stream);
int lsam_mlp_block_fwd(const float* x, float* y, const float* norm2_w, const float* norm2_b, const void* image,
                       const float* bias_table, int M, int d, float eps, lsam_stream_t stream);
/* As lsam_mlp_block_fwd, and also xn_next = LayerNorm(y) with the NEXT block's norm1 parameters (bf16 [M][ldn],
 * ldn % 4 == 0): `x = self.norm1(x)` of the following MultiScaleBlock (hieradet.py:136), taken from the registers
 * that hold y, so that block does not read y again for its LayerNorm. */
int lsam_mlp_block_fwd_next_norm(const float* x, float* y, const float* norm2_w, const float* norm2_b, const void* image,
                                 const float* bias_table, void* xn_next, int ldn, const float* next_norm_w,
                                 const float* next_norm_b, int M, int d, float eps, lsam_stream_t stream);

/* The same two kernels with the attention projection moved from the first into the second (what the plan runs for the
 * stage-1 blocks: at one wave per SIMD the attention kernel spends a third of its time in proj + residual, the MLP kernel is
 * VALU-bound and has the matrix pipe to spare):
 *   lsam_window_block_attention_fwd:  a = attn(qkv(LayerNorm1(x)))  -- the attention output rows (hieradet.py:59-77 inside
 *       window_partition / unpartition), bf16 [N*H*W][nh*hd]; same image and bias table as lsam_window_block_fwd;
 *   lsam_mlp_block_fwd_proj:  y1 = x + proj(a) (hieradet.py:79,163), y = y1 + fc2(GELU(fc1(LayerNorm2(y1)))) (:165), and
 *       optionally xn_next as in lsam_mlp_block_fwd_next_norm (NULL: not written).  a bf16 [M][lda], lda % 8 == 0.
 *       proj_image / proj_bias from lsam_mlp_block_proj_pack(attn.proj.weight [d][d], attn.proj.bias)
 *       (lsam_mlp_block_proj_image_bytes(d) bytes, 0 = width not supported: d <= 112 only; 224 floats). */
int lsam_window_block_attention_fwd(const float* x, void* a_out, const float* norm1_w, const float* norm1_b, const void* image,
                                    const float* bias_table, int N, int H, int W, int nh, int hd, float eps, lsam_stream_t stream);
size_t lsam_mlp_block_proj_image_bytes(int d);
int lsam_mlp_block_proj_pack(const float* proj_w, const float* proj_b, int d, void* proj_image, float* proj_bias, lsam_stream_t stream);
int lsam_mlp_block_fwd_proj(const float* x, const void* a, int lda, float* y, const float* norm2_w, const float* norm2_b,
                            const void* image, const float* bias_table, const void* proj_image, const float* proj_bias,
                            void* xn_next, int ldn, const float* next_norm_w, const float* next_norm_b, int M, int d, float eps,
                            lsam_stream_t stream);

/* Fused tail of a block, bf16 MFMA, in one kernel (stage-2 / stage-3 widths 224 and 448):
 *   y   = res + proj(a)                           (hieradet.py:79,163)
 *   out = y + fc2(GELU(fc1(LayerNorm2(y))))       (hieradet.py:165, MLP: sam2_utils.py:112-136)
 *   xn_next = LayerNorm(out; next_norm_w, next_norm_b)   (optional: norm1 of the next block, hieradet.py:136)
 * a bf16 [M][lda] (the attention output before attn.proj), res / out fp32 [M][d] (out may alias res),
 * xn_next bf16 [M][ldn] or NULL.  lsam_tail_block_pack builds the streamed weight image
 * (lsam_tail_block_image_bytes(d) bytes) and the packed vectors (lsam_tail_block_vec_floats(d) floats)
 * from attn.proj.{weight [d][d], bias}, norm2.{weight, bias}, mlp.layers.0.{weight [4d][d], bias} and
 * mlp.layers.1.{weight [d][4d], bias}.  Both size queries return 0 for an unsupported width. */
size_t lsam_tail_block_image_bytes(int d);
size_t lsam_tail_block_vec_floats(int d);
int lsam_tail_block_pack(const float* proj_w, const float* proj_b, const float* norm2_w, const float* norm2_b,
                         const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b, int d,
                         void* image, float* vec, lsam_stream_t stream);
int lsam_tail_block_fwd(const void* a, int lda, const float* res, float* out, const void* image, const float* vec,
                        void* xn_next, int ldn, const float* next_norm_w, const float* next_norm_b, int M, int d,
                        float eps, lsam_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Whole-trunk plan: Hiera.forward / Hiera.get_embedding (hieradet.py:284-334) with
 * LoRA_qkv on the last blocks (lora_qkv.py:102-109,162) and the autograd graph that
 * core_loss.backward() walks (train_lora_sam.py:172), as two calls.
 * ------------------------------------------------------------------------------------ */
typedef struct {
    int dtype;            /* LSAM_F32 / LSAM_BF16 */
    int img_size;         /* square input, multiple of 32 */
    int embed_dim;        /* Hiera(embed_dim=...) hieradet.py:176 */
    int num_heads;        /* initial heads :177 */
    int num_stages;       /* len(stages) <= 4 */
    int stages[4];        /* blocks per stage :181 */
    int window_spec[4];   /* :186 */
    int n_global;         /* len(global_att_blocks) */
    int global_att_blocks[8];
    int q_pool;           /* number of q-pool stages :179 */
    int lora_rank;        /* 0 = no LoRA; <= 16 */
    int n_lora_blocks;    /* trunk.blocks[-n:] wrapped & differentiated (2 in the reference, lora_qkv.py:162) */
} lsam_hiera_config_t;

typedef struct lsam_hiera lsam_hiera_t; /* host-side plan; owns no device memory */

/* fp32 master parameters of one MultiScaleBlock (device pointers, nn.Parameter layout). */
typedef struct {
    const float *norm1_w, *norm1_b;   /* [dim] */
    const float *qkv_w, *qkv_b;       /* [3*dim_out, dim], [3*dim_out] */
    const float *attn_proj_w, *attn_proj_b; /* [dim_out, dim_out] */
    const float *norm2_w, *norm2_b;   /* [dim_out] */
    const float *fc1_w, *fc1_b;       /* [4*dim_out, dim_out] */
    const float *fc2_w, *fc2_b;       /* [dim_out, 4*dim_out] */
    const float *proj_w, *proj_b;     /* [dim_out, dim] or NULL (hieradet.py:131-132) */
} lsam_block_params_t;

typedef struct {
    const float *a_q, *b_q, *a_v, *b_v; /* [r,d], [d,r], [r,d], [d,r]  (lora_qkv.py:168-171) */
} lsam_lora_params_t;

int lsam_hiera_create(const lsam_hiera_config_t* cfg, lsam_hiera_t** out);
void lsam_hiera_destroy(lsam_hiera_t* h);
int lsam_hiera_num_blocks(const lsam_hiera_t* h);
/* geometry of block i: dim, dim_out, heads, window, q-pool (0/1), input grid side */
int lsam_hiera_block_info(const lsam_hiera_t* h, int i, int* dim, int* dim_out, int* heads,
                          int* window, int* pool, int* grid);
/* stage s output: channels and grid side */
int lsam_hiera_stage_info(const lsam_hiera_t* h, int s, int* channels, int* grid);

size_t lsam_hiera_arena_bytes(const lsam_hiera_t* h);                 /* packed weights */
size_t lsam_hiera_workspace_bytes(const lsam_hiera_t* h, int batch, int with_grad);

/* Pack the frozen parameters into the arena (once, or after load_state_dict).
 * host_blocks: array of num_blocks structs (host memory) holding device pointers.
 * pos: fp32 [grid0*grid0][embed_dim], the interpolated+tiled positional embedding
 * (hieradet.py:273-281), constant per shape. */
int lsam_hiera_pack(lsam_hiera_t* h, const lsam_block_params_t* host_blocks, const float* patch_w,
                    const float* patch_b, const float* pos, void* arena, size_t arena_bytes,
                    lsam_stream_t stream);
/* Refresh the LoRA-dependent packed operands (every optimiser step). host_lora: n_lora_blocks. */
int lsam_hiera_pack_lora(lsam_hiera_t* h, const lsam_lora_params_t* host_lora, void* arena,
                         lsam_stream_t stream);

/* img fp32 [batch,3,S,S] -> host_stage_out[s] fp32 NHWC [batch, g_s, g_s, C_s] (device pointers
 * in a host array).  with_grad != 0 keeps what lsam_hiera_backward needs in the workspace. */
int lsam_hiera_forward(lsam_hiera_t* h, const float* img, int batch, float* const* host_stage_out,
                       void* arena, void* workspace, size_t workspace_bytes, int with_grad,
                       lsam_stream_t stream);
/* grad_out fp32 NHWC gradient of the LAST stage output -> host_lora_grads[4*i + {0:a_q,1:b_q,2:a_v,3:b_v}]
 * fp32, same shapes as the parameters, OVERWRITTEN (not accumulated).  Only LoRA factors get
 * gradients; no weight-gradient GEMM is run for any frozen tensor. */
int lsam_hiera_backward(lsam_hiera_t* h, const float* grad_out, int batch,
                        float* const* host_lora_grads, void* arena, void* workspace,
                        size_t workspace_bytes, lsam_stream_t stream);
/* One LoRA block of lsam_hiera_backward at a time, last block first (lora_block = n_lora_blocks-1 ... 0):
 * lets the caller start the all-reduce of a block's four gradients while the next block's backward
 * runs (DistributedDataParallel's bucketed overlap, train_lora_sam.py:126,172).  grad_out is read by
 * the last block only; the others continue from the dx their successor left in the workspace, so the
 * calls must be made in descending order on one stream with the same workspace.  host_lora_grads is
 * the same 4*n_lora_blocks table as for lsam_hiera_backward; only entries 4*lora_block.. are written. */
int lsam_hiera_backward_block(lsam_hiera_t* h, const float* grad_out, int batch, int lora_block,
                              float* const* host_lora_grads, void* arena, void* workspace, size_t workspace_bytes,
                              lsam_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LSAM_H */
