// Shared device/host helpers for the LoRA-SAM2 Hiera kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lsam.h"

namespace lsam {

using bf16 = __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x8 = __attribute__((ext_vector_type(8))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

// ---------------------------------------------------------------------------
// error reporting: never throw, never print; thread-local last-error string
// ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define LSAM_REQUIRE(cond, code, ...)        \
    do {                                     \
        if (!(cond)) {                       \
            ::lsam::set_error(__VA_ARGS__);  \
            return (code);                   \
        }                                    \
    } while (0)
#define LSAM_HIP(expr)                                                              \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            ::lsam::set_error("%s failed: %s", #expr, hipGetErrorString(e_));       \
            return LSAM_ERR_HIP;                                                    \
        }                                                                           \
    } while (0)
#define LSAM_TRY(expr)              \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != LSAM_OK) return rc_; \
    } while (0)
// kernel launches report configuration errors through hipGetLastError
#define LSAM_LAUNCH_CHECK() LSAM_HIP(hipGetLastError())

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------
// element traits.  A "chunk" is 16 bytes, a "fragment" is the 8 K-elements one
// lane feeds to a K=32 MFMA macro-step:  lane l holds  X[row = l&15][k = 8*(l>>4) + j].
// For bf16 that is one v_mfma_f32_16x16x32_bf16; for f32 the same mapping is
// realised by eight v_mfma_f32_16x16x4_f32 (element j of every lane forms one
// K=4 step), so both dtypes share all indexing.  C/D layout is dtype-independent:
// lane l, reg r  ->  D[row = 4*(l>>4) + r][col = l&15].
// ---------------------------------------------------------------------------
template <typename T> struct Tr;
template <> struct Tr<float> {
    static constexpr int EPC = 4;   // elements per 16-byte chunk
    static constexpr int CPF = 2;   // chunks per fragment
    using frag = f32x8;
    static constexpr int DTYPE = LSAM_F32;
};
template <> struct Tr<bf16> {
    static constexpr int EPC = 8;
    static constexpr int CPF = 1;
    using frag = bf16x8;
    static constexpr int DTYPE = LSAM_BF16;
};

__device__ __forceinline__ void mma16(const bf16x8& a, const bf16x8& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma16(const f32x8& a, const f32x8& b, f32x4& c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
}

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }

// fragment from one or two 16-byte chunks already in registers
__device__ __forceinline__ bf16x8 make_frag(const uint4& c0, const uint4&, bf16) {
    return __builtin_bit_cast(bf16x8, c0);
}
__device__ __forceinline__ f32x8 make_frag(const uint4& c0, const uint4& c1, float) {
    f32x4 a = __builtin_bit_cast(f32x4, c0), b = __builtin_bit_cast(f32x4, c1);
    f32x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

// 4 consecutive values -> packed store (16 B f32 / 8 B bf16)
__device__ __forceinline__ void store4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16* p, const f32x4& v) {
    bf16x4 o;
    o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16* p) {
    bf16x4 o = *reinterpret_cast<const bf16x4*>(p);
    f32x4 v;
    v[0] = (float)o[0]; v[1] = (float)o[1]; v[2] = (float)o[2]; v[3] = (float)o[3];
    return v;
}

// LDS tile rows are 128 B (8 chunks) or 256 B (16 chunks); chunk index XOR (row&7)
// makes ds_read_b128 of 16 rows x one chunk column conflict-free (guide T2).
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ (row & 7); }

// Exact-erf GELU (nn.GELU() default).  The fp32 path uses libm erff; the bf16 path, whose result is
// rounded to 8 significant bits anyway, uses the Abramowitz-Stegun 7.1.26 rational form
// (|abs err| < 1.5e-7, ~15 VALU ops instead of erff's ~50): the epilogue of the fc1 GEMM evaluates it
// on 4 x dim_out values per token, which is enough to make a compute-bound GEMM VALU-bound.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float y = fmaf(1.061405429f, t, -1.453152027f);
    y = fmaf(y, t, 1.421413741f);
    y = fmaf(y, t, -0.284496736f);
    y = fmaf(y, t, 0.254829592f);
    y = y * t * __expf(-ax * ax);
    return copysignf(1.0f - y, x);
}
// bf16 forward GELU: x * Phi(x) with Phi(x) = sigmoid(x * P(x^2)), P a quadratic in x^2 fitted (minimax, error
// measured relative to max(|gelu|, 0.02)) to the exact erf form on |x| <= 7: |abs err| <= 6.2e-5 everywhere and
// <= 0.04 % of max(|gelu|, 0.02), i.e. well below the 0.2-0.4 % half-ulp of the bf16 value it is rounded to
// (a cubic P buys 4.0e-5 for one more FMA; the tanh form's linear P is 1.1e-3).  9 VALU ops (2 transcendental)
// against ~17 for the erf form: the GELU is what bounds the fused MLP kernels (470 M evaluations per stage-1
// block at batch 16).  The coefficients carry the factor -log2(e) so that the sigmoid is 1 / (1 + exp2(x * P)).
__device__ __forceinline__ float gelu_sig(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -7.0f, 7.0f);
    const float x2 = xc * xc;
    float p = fmaf(1.1102255e-03f, x2, -1.0751102e-01f);   // -log2e * (-7.69549653e-04, 7.45209629e-02)
    p = fmaf(p, x2, -2.3001459e+00f);                      // -log2e * 1.59433958
    const float e = __builtin_amdgcn_exp2f(xc * p);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
template <bool FAST> __device__ __forceinline__ float gelu_t(float x) {
    if constexpr (FAST) return gelu_sig(x);
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
}
template <bool FAST> __device__ __forceinline__ float gelu_grad_t(float x) {
    const float e = FAST ? erf_fast(x * 0.70710678118654752f) : erff(x * 0.70710678118654752f);
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return 0.5f * (1.0f + e) + x * pdf;
}

// XCD-aware bijective block remap (guide T1): blocks b and b+8 share an XCD, so give
// every XCD a contiguous run of logical tile ids.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Barrier that publishes LDS written by LDS-DMA (buffer_load ... lds / global_load_lds): each wave
// first waits for its OWN outstanding DMA (vmcnt(0)), then joins the barrier.  __syncthreads() alone
// is not enough: hipcc does not count the LDS write of a DMA as something a workgroup release fence
// has to wait for, and is free to place its vmcnt(0) AFTER the s_barrier (seen in the generated
// code), which lets a wave read rows another wave's DMA has not landed yet.
// Reductions over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48: the lanes that hold one token's
// values in the MFMA D layout) on the VALU: v_permlane32_swap / v_permlane16_swap of a value with itself give
// (x.lo, x.lo | x.hi, x.hi) and (r0, r0, r2, r2 | r1, r1, r3, r3), so op(a, b) is the xor-32 / xor-16 butterfly.
// __shfl_xor compiles to ds_bpermute_b32: an LDS-pipe round trip with an lgkmcnt wait on the critical path of
// every softmax row maximum.
// Written as inline asm on two explicit copies: the instructions swap IN PLACE between two registers, and the
// builtin form, given one value twice, was seen to hand back a single register for both results.  The s_nop
// covers the VALU-write -> permlane-swap read hazard, which the compiler cannot see inside an asm statement.
__device__ __forceinline__ void swap32_lanes(float& a, float& b) {   // a = (a.lo, b.lo), b = (a.hi, b.hi)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16_rows(float& a, float& b) {    // a = (a.r0, b.r0, a.r2, b.r2), b = (a.r1, b.r1, a.r3, b.r3)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32_lanes(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16_rows(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// Two neighbouring 16-column tiles of the MFMA D layout (lane (g, lr): row lr, columns 16 nt + 4 g + r) as bf16:
// regrouped across the four lanes of a row so that lane g holds the 8 consecutive columns 32 (nt/2) + 8 g .. + 7 and
// stores them as ONE 16-byte word (64 contiguous bytes per row and instruction instead of 4 x 8).  Every lane of the
// wave must call this (the swaps need all lanes); `ok` only guards the store.  rowp points at column 16 nt of the row.
__device__ __forceinline__ void store_bf16_tile_pair(bf16* rowp, int g, const f32x4& v0, const f32x4& v1, bool ok) {
    bf16x4 p0, p1;
#pragma unroll
    for (int r = 0; r < 4; ++r) { p0[r] = (bf16)v0[r]; p1[r] = (bf16)v1[r]; }
    uint2 x = __builtin_bit_cast(uint2, p0), y = __builtin_bit_cast(uint2, p1);
    swap32_lanes(x.x, y.x);
    swap32_lanes(x.y, y.y);
    swap16_rows(x.x, y.x);
    swap16_rows(x.y, y.y);
    if (ok) *reinterpret_cast<uint4*>(rowp + 8 * g) = make_uint4(x.x, x.y, y.x, y.y);
}
__device__ __forceinline__ float group4_max(float x) {
    float a = x, b = x;
    swap32_lanes(a, b);
    a = b = fmaxf(a, b);
    swap16_rows(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float group4_sum(float x) {
    float a = x, b = x;
    swap32_lanes(a, b);
    a = b = a + b;
    swap16_rows(a, b);
    return a + b;
}

__device__ __forceinline__ void lds_dma_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

}  // namespace lsam
