// FPN neck levels as single kernels (FpnNeck.forward, image_encoder.py:102-134), straight from the fp32 NHWC stage
// outputs of the trunk:
//   out[m][:] = T( x[m][:] . W^T + b  (+ top[parent(m)][:]) )
// i.e. the lateral 1x1 convolution (`self.convs[n - i](x)`, :112-114) with the nearest-x2 top-down add
// (`F.interpolate(prev.to(float32), scale_factor=2, mode="nearest")` + `lateral + top_down`, :116-125) in its epilogue.
// x is read once as fp32 and converted to MFMA operand fragments in registers (a lane's two 16-byte loads ARE its
// k-contiguous fragment); the weights are a fragment-ordered image (1 KiB per 16 x 32 bf16 fragment, 16 bytes per lane,
// perfectly coalesced) that every wave streams from L2 -- the images are 16 .. 460 KB -- so the kernel needs no LDS and no
// barrier.  The product is formed transposed (W fragment as the A operand): a lane ends up with 4 consecutive output
// channels of one token.  Used for every level in fp32 mode, for the two low-resolution levels (K = 448 / 896, whose
// weights do not fit LDS) in bf16 mode, and for the composed conv_s0 / conv_s1 . lateral convolutions of
// SAM2Base.forward_image (sam2_base.py:473-478), which have 32 / 64 output channels.
//
// neck_topdown_bwd_kernel is the adjoint of the nearest-x2 upsample: the gradient of a coarse token is the sum over its
// 2 x 2 children (+ the coarse level's own gradient).
#include "kernels.hpp"

namespace lsam {

template <typename T>
__global__ void pack_neck_kernel(const float* __restrict__ w, int K, int N, int nks, T* __restrict__ img) {
    const int NT = N / 16;
    const long total = (long)nks * NT * 512;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i >> 9), lane = (int)((i & 511) >> 3), j = (int)(i & 7);
        const int ks = f / NT, nt = f % NT, g = lane >> 4, lr = lane & 15;
        const int k = 32 * ks + 8 * g + j;
        img[i] = from_f<T>(k < K ? w[(size_t)(16 * nt + lr) * K + k] : 0.f);
    }
}

// NT = output column tiles per WAVE, MT = token tiles per wave, NSPLIT = 4: the four waves of a workgroup share one set of
// 16 MT tokens and take a quarter of the output channels each (d_model = 256: every weight fragment then feeds MT MFMAs and a
// wave holds only MT x NT accumulators, which leaves registers for a one-k-step-ahead prefetch of both operands; the three
// other waves' re-reads of x hit L1); NSPLIT = 1: every wave has its own tokens and all output channels (32 / 64 channels).
template <typename T, int NT, int MT, int NSPLIT>
__global__ __launch_bounds__(256) void neck_level_kernel(const float* __restrict__ x, const T* __restrict__ wimg,
                                                         const float* __restrict__ bias, const T* __restrict__ top,
                                                         T* __restrict__ out, int M, int K, int H, int W) {
    using frag = typename Tr<T>::frag;
    constexpr int NTA = NT * NSPLIT, N = 16 * NTA;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, lr = lane & 15;
    const int m0 = (NSPLIT == 1 ? blockIdx.x * 4 + wave : blockIdx.x) * 16 * MT;
    const int nt0 = NSPLIT == 1 ? 0 : wave * NT;
    if (m0 >= M) return;   // wave-uniform; the kernel has no barrier
    const int nks = (K + 31) / 32;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f32x4 b = load4(bias + 16 * (nt0 + nt) + 4 * g);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = b;
    }
    const float* rows[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rows[mt] = x + (size_t)min(m0 + 16 * mt + lr, M - 1) * K;   // clamped rows are never stored
    const frag* wf = reinterpret_cast<const frag*>(wimg) + (size_t)nt0 * 64 + lane;

    auto load_x = [&](int ks, f32x4 (&xa)[MT], f32x4 (&xb)[MT]) {
        const int c0 = 32 * ks + 8 * g;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            xa[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            xb[mt] = xa[mt];
            if (c0 < K) { xa[mt] = load4(rows[mt] + c0); xb[mt] = load4(rows[mt] + c0 + 4); }
        }
    };
    auto load_w = [&](int ks, frag (&w)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) w[nt] = wf[(size_t)(ks * NTA + nt) * 64];
    };
    f32x4 xa[MT], xb[MT];
    frag w[NT];
    load_x(0, xa, xb);
    load_w(0, w);
    for (int ks = 0; ks < nks; ++ks) {
        frag xf[MT], wc[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) { xf[mt][j] = from_f<T>(xa[mt][j]); xf[mt][4 + j] = from_f<T>(xb[mt][j]); }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wc[nt] = w[nt];
        if (ks + 1 < nks) {   // the next k-step's operands are in flight under this one's MFMAs
            load_x(ks + 1, xa, xb);
            load_w(ks + 1, w);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) mma16(wc[nt], xf[mt], acc[mt][nt]);
    }

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + 16 * mt + lr;
        const bool ok = m < M;
        const int mc = ok ? m : M - 1;
        T* rowp = out + (size_t)mc * N + 16 * nt0;
        const T* trow = nullptr;
        if (top) {   // nearest x2: token (n, y, xx) of the H x W grid takes the coarse token (n, y / 2, xx / 2)
            const int xx = mc % W, y = (mc / W) % H, n = mc / (W * H);
            trow = top + ((size_t)(n * (H / 2) + y / 2) * (W / 2) + xx / 2) * N + 16 * nt0;
        }
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int nt = 0; nt < NT; nt += 2) {
                f32x4 v0 = acc[mt][nt], v1 = acc[mt][nt + 1];
                if (trow) {
                    v0 += load4(trow + 16 * nt + 4 * g);
                    v1 += load4(trow + 16 * (nt + 1) + 4 * g);
                }
                store_bf16_tile_pair(reinterpret_cast<bf16*>(rowp) + 16 * nt, g, v0, v1, ok);
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 v = acc[mt][nt];
                if (trow) v += load4(trow + 16 * nt + 4 * g);
                if (ok) store4(rowp + 16 * nt + 4 * g, v);
            }
        }
    }
}

// Two-term bf16 operands ("split"): x = xh + xl and W = Wh + Wl with xh = bf16(x), xl = bf16(x - xh) (likewise W), and
//   x . W ~= xh.Wh + xl.Wh + xh.Wl            (the dropped xl.Wl term is 2^-16 relative)
// on the bf16 MFMA pipe: the product is accurate to ~2^-16 instead of the 2^-8 of rounded bf16 operands, at 3 MFMAs per
// step -- these levels are bound by the x / weight streams, not by the matrix pipe, and fp32 MFMA would run at 1/16 of
// the rate.  Used for the top-down levels of the neck under bf16 autocast (what the SAM transformer reads: rounding
// inputs, weights AND outputs there costs 0.4 points of mask IoU on the north-star fixture).  The image is [hi | lo];
// the top-down addend and the optional second output (the next finer level's addend) are fp32, so the chain is rounded
// once per level, at the bf16 store.  d_model = 256 only (NSPLIT = 4 layout of neck_level_kernel).
__global__ void pack_neck_split_kernel(const float* __restrict__ w, int K, int N, int nks, bf16* __restrict__ img) {
    const int NT = N / 16;
    const long total = (long)nks * NT * 512;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i >> 9), lane = (int)((i & 511) >> 3), j = (int)(i & 7);
        const int ks = f / NT, nt = f % NT, g = lane >> 4, lr = lane & 15;
        const int k = 32 * ks + 8 * g + j;
        const float v = k < K ? w[(size_t)(16 * nt + lr) * K + k] : 0.f;
        const bf16 h = (bf16)v;
        img[i] = h;
        img[total + i] = (bf16)(v - (float)h);
    }
}

template <int NT, int MT>
__global__ __launch_bounds__(256) void neck_level_split_kernel(const float* __restrict__ x, const bf16* __restrict__ wimg,
                                                               const float* __restrict__ bias, const float* __restrict__ top,
                                                               bf16* __restrict__ out, float* __restrict__ out32, int M, int K,
                                                               int H, int W) {
    constexpr int NTA = NT * 4, N = 16 * NTA;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, lr = lane & 15;
    const int m0 = blockIdx.x * 16 * MT;
    const int nt0 = wave * NT;
    if (m0 >= M) return;   // wave-uniform; the kernel has no barrier
    const int nks = (K + 31) / 32;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f32x4 b = load4(bias + 16 * (nt0 + nt) + 4 * g);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = b;
    }
    const float* rows[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rows[mt] = x + (size_t)min(m0 + 16 * mt + lr, M - 1) * K;   // clamped rows are never stored
    const bf16x8* wf = reinterpret_cast<const bf16x8*>(wimg) + (size_t)nt0 * 64 + lane;
    const size_t lo = (size_t)nks * NTA * 64;

    auto load_x = [&](int ks, f32x4 (&xa)[MT], f32x4 (&xb)[MT]) {
        const int c0 = 32 * ks + 8 * g;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            xa[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            xb[mt] = xa[mt];
            if (c0 < K) { xa[mt] = load4(rows[mt] + c0); xb[mt] = load4(rows[mt] + c0 + 4); }
        }
    };
    auto load_w = [&](int ks, bf16x8 (&wh)[NT], bf16x8 (&wl)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            wh[nt] = wf[(size_t)(ks * NTA + nt) * 64];
            wl[nt] = wf[lo + (size_t)(ks * NTA + nt) * 64];
        }
    };
    f32x4 xa[MT], xb[MT];
    bf16x8 wh[NT], wl[NT];
    load_x(0, xa, xb);
    load_w(0, wh, wl);
    for (int ks = 0; ks < nks; ++ks) {
        bf16x8 xh[MT], xl[MT], ch[NT], cl[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16 a = (bf16)xa[mt][j], b = (bf16)xb[mt][j];
                xh[mt][j] = a;
                xh[mt][4 + j] = b;
                xl[mt][j] = (bf16)(xa[mt][j] - (float)a);
                xl[mt][4 + j] = (bf16)(xb[mt][j] - (float)b);
            }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { ch[nt] = wh[nt]; cl[nt] = wl[nt]; }
        if (ks + 1 < nks) {   // the next k-step's operands are in flight under this one's MFMAs
            load_x(ks + 1, xa, xb);
            load_w(ks + 1, wh, wl);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                mma16(cl[nt], xh[mt], acc[mt][nt]);   // small terms first
                mma16(ch[nt], xl[mt], acc[mt][nt]);
                mma16(ch[nt], xh[mt], acc[mt][nt]);
            }
    }

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + 16 * mt + lr;
        const bool ok = m < M;
        const int mc = ok ? m : M - 1;
        const float* trow = nullptr;
        if (top) {
            const int xx = mc % W, y = (mc / W) % H, n = mc / (W * H);
            trow = top + ((size_t)(n * (H / 2) + y / 2) * (W / 2) + xx / 2) * N + 16 * nt0;
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt += 2) {
            f32x4 v0 = acc[mt][nt], v1 = acc[mt][nt + 1];
            if (trow) {
                v0 += load4(trow + 16 * nt + 4 * g);
                v1 += load4(trow + 16 * (nt + 1) + 4 * g);
            }
            if (out32 && ok) {
                float* r32 = out32 + (size_t)mc * N + 16 * (nt0 + nt) + 4 * g;
                store4(r32, v0);
                store4(r32 + 16, v1);
            }
            store_bf16_tile_pair(out + (size_t)mc * N + 16 * (nt0 + nt), g, v0, v1, ok);
        }
    }
}

template <typename T>
__global__ void neck_topdown_bwd_kernel(const T* __restrict__ gf, const T* __restrict__ gc, T* __restrict__ out, int N, int Hc,
                                        int Wc, int C4) {
    const long total = (long)N * Hc * Wc * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long t = i / C4;
        const int xc = (int)(t % Wc);
        t /= Wc;
        const int yc = (int)(t % Hc), n = (int)(t / Hc);
        const size_t rs = (size_t)2 * Wc * C4 * 4;
        const T* p = gf + (((size_t)n * 2 * Hc + 2 * yc) * 2 * Wc + 2 * xc) * (size_t)(C4 * 4) + 4 * c;
        f32x4 v = (load4(p) + load4(p + C4 * 4)) + (load4(p + rs) + load4(p + rs + C4 * 4));
        if (gc) v += load4(gc + i * 4);
        store4(out + i * 4, v);
    }
}

static int neck_variant(int K, int N) {
    if (K % 8 != 0 || K < 8 || K > 4096) return -1;
    if (N == 256) return 0;
    if (N == 64) return 1;
    if (N == 32) return 2;
    return -1;
}
bool neck_level_supported(int K, int N) { return neck_variant(K, N) >= 0; }
size_t neck_image_bytes(int K, int N, size_t esz) { return neck_variant(K, N) < 0 ? 0 : (size_t)((K + 31) / 32) * (N / 16) * 512 * esz; }

template <typename T> int launch_pack_neck(const float* w, int K, int N, void* img, hipStream_t s) {
    LSAM_REQUIRE(w && img, LSAM_ERR_BAD_ARG, "neck_conv_pack: null argument");
    LSAM_REQUIRE(neck_variant(K, N) >= 0, LSAM_ERR_UNSUPPORTED, "neck_conv_pack: %d -> %d channels not supported", K, N);
    const int nks = (K + 31) / 32;
    hipLaunchKernelGGL(pack_neck_kernel<T>, dim3(128), dim3(256), 0, s, w, K, N, nks, static_cast<T*>(img));
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
template int launch_pack_neck<float>(const float*, int, int, void*, hipStream_t);
template int launch_pack_neck<bf16>(const float*, int, int, void*, hipStream_t);

template <typename T>
int launch_neck_level(const float* x, const void* img, const float* bias, const void* top, void* out, int Nimg, int H, int W,
                      int K, int N, hipStream_t s) {
    LSAM_REQUIRE(x && img && bias && out && Nimg > 0 && H > 0 && W > 0, LSAM_ERR_BAD_ARG, "neck_level: null/empty argument");
    const int v = neck_variant(K, N);
    LSAM_REQUIRE(v >= 0, LSAM_ERR_UNSUPPORTED, "neck_level: %d -> %d channels not supported", K, N);
    LSAM_REQUIRE(!top || (H % 2 == 0 && W % 2 == 0), LSAM_ERR_UNSUPPORTED, "neck_level: the top-down add needs an even grid");
    LSAM_REQUIRE(aligned16(x) && aligned16(img) && aligned16(bias) && aligned16(out) && (!top || aligned16(top)),
                 LSAM_ERR_BAD_ARG, "neck_level: pointers must be 16-byte aligned");
    LSAM_REQUIRE((double)Nimg * H * W < 2.0e9, LSAM_ERR_UNSUPPORTED, "neck_level: too many tokens");
    const int M = Nimg * H * W;
    ProfScope ps(s, (double)M * (4.0 * K + (double)sizeof(T) * N * (top ? 1.25 : 1.0)), PROF_NECK, M, N, K);
    const T* wi = static_cast<const T*>(img);
    const T* tp = static_cast<const T*>(top);
    T* o = static_cast<T*>(out);
#define NECK_LAUNCH(NT, MT, NSPLIT)                                                                                      \
    hipLaunchKernelGGL((neck_level_kernel<T, NT, MT, NSPLIT>), dim3(cdiv(M, (NSPLIT == 1 ? 64 : 16) * MT)), dim3(256), 0, s, x, \
                       wi, bias, tp, o, M, K, H, W)
    if (v == 0) {
        if constexpr (sizeof(T) == 2) {
            if (M >= 32768) NECK_LAUNCH(4, 4, 4);   // 64 tokens per workgroup
            else NECK_LAUNCH(4, 2, 4);              // small levels: 32 tokens, twice the workgroups
        } else {
            NECK_LAUNCH(4, 2, 4);
        }
    } else if (v == 1) {
        NECK_LAUNCH(4, 4, 1);
    } else {
        NECK_LAUNCH(2, 4, 1);
    }
#undef NECK_LAUNCH
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
template int launch_neck_level<float>(const float*, const void*, const float*, const void*, void*, int, int, int, int, int, hipStream_t);
template int launch_neck_level<bf16>(const float*, const void*, const float*, const void*, void*, int, int, int, int, int, hipStream_t);

size_t neck_split_image_bytes(int K, int N) { return N == 256 && neck_variant(K, N) >= 0 ? 2 * neck_image_bytes(K, N, 2) : 0; }

int launch_pack_neck_split(const float* w, int K, int N, void* img, hipStream_t s) {
    LSAM_REQUIRE(w && img, LSAM_ERR_BAD_ARG, "neck_conv_pack_split: null argument");
    LSAM_REQUIRE(neck_split_image_bytes(K, N) > 0, LSAM_ERR_UNSUPPORTED, "neck_conv_pack_split: %d -> %d channels not supported (d_model 256 only)", K, N);
    hipLaunchKernelGGL(pack_neck_split_kernel, dim3(128), dim3(256), 0, s, w, K, N, (K + 31) / 32, static_cast<bf16*>(img));
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

int launch_neck_level_split(const float* x, const void* img, const float* bias, const float* top, void* out, float* out32,
                            int Nimg, int H, int W, int K, int N, hipStream_t s) {
    LSAM_REQUIRE(x && img && bias && out && Nimg > 0 && H > 0 && W > 0, LSAM_ERR_BAD_ARG, "neck_level_split: null/empty argument");
    LSAM_REQUIRE(neck_split_image_bytes(K, N) > 0, LSAM_ERR_UNSUPPORTED, "neck_level_split: %d -> %d channels not supported (d_model 256 only)", K, N);
    LSAM_REQUIRE(!top || (H % 2 == 0 && W % 2 == 0), LSAM_ERR_UNSUPPORTED, "neck_level_split: the top-down add needs an even grid");
    LSAM_REQUIRE(aligned16(x) && aligned16(img) && aligned16(bias) && aligned16(out) && (!top || aligned16(top)) &&
                     (!out32 || aligned16(out32)), LSAM_ERR_BAD_ARG, "neck_level_split: pointers must be 16-byte aligned");
    LSAM_REQUIRE((double)Nimg * H * W < 2.0e9, LSAM_ERR_UNSUPPORTED, "neck_level_split: too many tokens");
    const int M = Nimg * H * W;
    ProfScope ps(s, (double)M * (4.0 * K + N * (2.0 + (top ? 1.0 : 0.0) + (out32 ? 4.0 : 0.0))), PROF_NECK, M, N, K);
    if (M >= 32768)
        hipLaunchKernelGGL((neck_level_split_kernel<4, 4>), dim3(cdiv(M, 64)), dim3(256), 0, s, x, static_cast<const bf16*>(img), bias,
                           top, static_cast<bf16*>(out), out32, M, K, H, W);
    else
        hipLaunchKernelGGL((neck_level_split_kernel<4, 2>), dim3(cdiv(M, 32)), dim3(256), 0, s, x, static_cast<const bf16*>(img), bias,
                           top, static_cast<bf16*>(out), out32, M, K, H, W);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

template <typename T>
int launch_neck_topdown_bwd(const void* gf, const void* gc, void* out, int Nimg, int Hc, int Wc, int C, hipStream_t s) {
    LSAM_REQUIRE(gf && out && Nimg > 0 && Hc > 0 && Wc > 0 && C > 0, LSAM_ERR_BAD_ARG, "neck_topdown_bwd: null/empty argument");
    LSAM_REQUIRE(C % 4 == 0 && (reinterpret_cast<uintptr_t>(gf) & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0 &&
                     (!gc || (reinterpret_cast<uintptr_t>(gc) & 7) == 0), LSAM_ERR_BAD_ARG, "neck_topdown_bwd: C %% 4, alignment");
    const long total = (long)Nimg * Hc * Wc * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(neck_topdown_bwd_kernel<T>, dim3(grid), dim3(256), 0, s, static_cast<const T*>(gf),
                       static_cast<const T*>(gc), static_cast<T*>(out), Nimg, Hc, Wc, C / 4);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}
template int launch_neck_topdown_bwd<float>(const void*, const void*, void*, int, int, int, int, hipStream_t);
template int launch_neck_topdown_bwd<bf16>(const void*, const void*, void*, int, int, int, int, hipStream_t);

}  // namespace lsam
