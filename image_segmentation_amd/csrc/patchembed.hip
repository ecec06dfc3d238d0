// PatchEmbed as a direct (implicit-GEMM) convolution, bf16 MFMA, fp32 out:
//   x[n][oy][ox][:] = conv7x7_stride4_pad3(img[n])[:, oy, ox] + b + pos[oy][ox][:]
// (backbones/utils.py:85-93 PatchEmbed.forward incl. the NCHW -> NHWC permute, and the position-embedding add of
// Hiera.forward, hieradet.py:307-312).  The im2col + GEMM formulation moves the 147-wide column matrix
// through HBM twice (1M x 160 bf16 = 335 MB at batch 16); here the frames are read once and only x is written.
//
// A workgroup (4 waves) owns a tile of 8 x 32 output pixels.  The input patch of the tile, 3 x 35 x 132
// pixels, is converted to bf16 and staged in LDS with column 0 = input column 4 ox0 - 3, so that the 7 taps
// of output column oxl start at LDS column 4 oxl: 8-byte aligned, and one extra (8th) tap with a zero weight
// makes a (channel, tap-row) pair exactly one 8-wide k-group of the MFMA.
//   K order: k = 8 q + kx, q = 7 c + ky (21 groups, padded to 24 = 6 k-steps of 32).
//   B operand (pixels): lane (g, lr) = pixel lr of a 16-pixel row segment, k-group q = 4 ks + g: 8 consecutive
//                       bf16 of one LDS row: two ds_read_b64.
//   A operand (weights): packed once into fragment order (1 KiB per (channel tile, k-step)), copied to LDS
//                       by every workgroup at start (persistent workgroups).
//   D: lane (g, lr) holds pixel lr, channels 16 nt + 4 g + r: 16-byte fp32 stores, 64 contiguous bytes per pixel.
#include "kernels.hpp"

namespace lsam {

namespace {
constexpr int PE_TY = 8, PE_TX = 32;                 // output tile
constexpr int PE_R = 4 * PE_TY + 3;                  // 35 input rows
constexpr int PE_PITCH = 4 * PE_TX + 8;              // 136 bf16 per LDS row (132 used)
constexpr int PE_C4 = (4 * PE_TX + 4) / 4;           // 33 float4 per input row
constexpr int PE_KS = 6;                             // k-steps (24 k-groups, 21 used)
constexpr int PE_TILE_B = 3 * PE_R * PE_PITCH * 2;   // 28560 bytes
typedef f32x4 f32x4_a4 __attribute__((aligned(4)));  // 16-byte vector at dword alignment
}  // namespace

template <int NT>
__global__ void pack_patch_kernel(const float* __restrict__ w, int e0, char* __restrict__ img) {
    const int total = NT * PE_KS * 512;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int f = i >> 9, lane = (i & 511) >> 3, j = i & 7;
        const int nt = f / PE_KS, ks = f % PE_KS, g = lane >> 4, lr = lane & 15;
        const int ch = 16 * nt + lr, q = 4 * ks + g;
        float v = 0.f;
        if (ch < e0 && q < 21 && j < 7) v = w[(size_t)ch * 147 + q * 7 + j];   // [e0][3][7][7]: c*49 + ky*7 + kx = 7 q + kx
        reinterpret_cast<bf16*>(img)[i] = (bf16)v;
    }
}

template <int NT>
__global__ __launch_bounds__(256, 2) void patch_embed_kernel(const float* __restrict__ frames, const char* __restrict__ wimg,
                                                             const float* __restrict__ bias, const float* __restrict__ pos,
                                                             float* __restrict__ out, int N, int S, int e0) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    bf16* tile = reinterpret_cast<bf16*>(smem);
    const char* wl = smem + ((PE_TILE_B + 1023) & ~1023);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, lr = lane & 15;
    const int G = S >> 2;
    const int tiles_x = (G + PE_TX - 1) / PE_TX, tiles_y = (G + PE_TY - 1) / PE_TY;
    const int ntiles = N * tiles_x * tiles_y;

    for (int i = tid; i < NT * PE_KS * 64; i += 256)   // weight image -> LDS, once
        *reinterpret_cast<uint4*>(smem + ((PE_TILE_B + 1023) & ~1023) + i * 16) = *reinterpret_cast<const uint4*>(wimg + i * 16);

    // per-lane LDS element offset of k-group q = 4 ks + g (groups past 20 carry zero weights: any valid row)
    int qoff[PE_KS];
#pragma unroll
    for (int ks = 0; ks < PE_KS; ++ks) {
        const int q = min(4 * ks + g, 20), c = q / 7, ky = q - 7 * c;
        qoff[ks] = (c * PE_R + ky) * PE_PITCH;
    }

    // A thread's input words of a tile are all requested together (one memory round trip per tile, not one per loop
    // iteration) and before the barrier that frees the LDS tile: they overlap the tail of the other waves' compute on
    // the previous tile.  (Requesting them a whole tile ahead needs 56 more live registers than the kernel has:
    // 199 spilled registers, measured.)
    constexpr int NLD = (3 * PE_R * PE_C4 + 255) / 256;
    f32x4 ld[NLD];
    auto request = [&](int t) {
        const int n = t / (tiles_x * tiles_y), tr = t % (tiles_x * tiles_y);
        const int iy0 = 4 * (tr / tiles_x) * PE_TY - 3, ix0 = 4 * (tr % tiles_x) * PE_TX - 3;
        int tid_ = tid;   // opaque per call: keeps the 14 words' lane-constant address arithmetic out of the tile loop's
        asm volatile("" : "+v"(tid_));   // live registers (hoisted, it spills the accumulators)
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int i = tid_ + 256 * it;
            const int c4 = i % PE_C4, rr = i / PE_C4, r = rr % PE_R, c = rr / PE_R;
            const int iy = iy0 + r, ix = ix0 + 4 * c4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < 3 * PE_R * PE_C4 && iy >= 0 && iy < S) {
                const float* row = frames + (((size_t)n * 3 + c) * S + iy) * S;
                if (ix >= 0 && ix + 3 < S) {
                    // 4-byte aligned 16-byte load (ix = 4 ox0 - 3 + 4 c4): dword alignment is all the hardware needs
                    v = *reinterpret_cast<const f32x4_a4*>(row + ix);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ix + e >= 0 && ix + e < S) v[e] = row[ix + e];
                }
            }
            ld[it] = v;
        }
    };

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int n = t / (tiles_x * tiles_y), tr = t % (tiles_x * tiles_y);
        const int oy0 = (tr / tiles_x) * PE_TY, ox0 = (tr % tiles_x) * PE_TX;
        request(t);
        // the accumulators start at the position embedding: requested here, behind the input words (vmcnt retires in
        // order, so staging does not wait for them), they land under the two barriers and the staging stores
        f32x4 acc[4][NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oy = min(oy0 + 2 * wave + (mt >> 1), G - 1), ox = min(ox0 + 16 * (mt & 1) + lr, G - 1);
            const float* pp = pos + ((size_t)oy * G + ox) * e0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c0 = 16 * nt + 4 * g;
                acc[mt][nt] = c0 < e0 ? load4(pp + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();   // every wave is done with the previous tile (and the weight copy is visible)
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int i = tid + 256 * it;
            if (i < 3 * PE_R * PE_C4) {
                const int c4 = i % PE_C4, rr = i / PE_C4;   // rr = c * PE_R + r
                bf16x4 o;
                o[0] = (bf16)ld[it][0]; o[1] = (bf16)ld[it][1]; o[2] = (bf16)ld[it][2]; o[3] = (bf16)ld[it][3];
                *reinterpret_cast<bf16x4*>(tile + rr * PE_PITCH + 4 * c4) = o;
            }
        }
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c0 = 16 * nt + 4 * g;
            const f32x4 b = c0 < e0 ? load4(bias + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] += b;
        }
#pragma unroll
        for (int ks = 0; ks < PE_KS; ++ks) {
            bf16x8 pf[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int oyl = 2 * wave + (mt >> 1), oxl = 16 * (mt & 1) + lr;
                const bf16* src = tile + qoff[ks] + 4 * oyl * PE_PITCH + 4 * oxl;
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(src), hi = *reinterpret_cast<const bf16x4*>(src + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { pf[mt][j] = lo[j]; pf[mt][4 + j] = hi[j]; }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wl + (nt * PE_KS + ks) * 1024 + lane * 16);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) mma16(wf, pf[mt], acc[mt][nt]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oy = oy0 + 2 * wave + (mt >> 1), ox = ox0 + 16 * (mt & 1) + lr;
            if (oy < G && ox < G) {
                float* op = out + ((size_t)n * G * G + (size_t)oy * G + ox) * e0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int c0 = 16 * nt + 4 * g;
                    if (c0 < e0) store4(op + c0, acc[mt][nt]);
                }
            }
        }
    }
}

bool patch_embed_supported(int e0) { return e0 % 16 == 0 && e0 >= 16 && e0 <= 144; }
size_t patch_embed_image_bytes(int e0) { return (size_t)(e0 <= 112 ? 7 : 9) * PE_KS * 1024; }

int launch_pack_patch_embed(const float* w, int e0, void* img, hipStream_t s) {
    LSAM_REQUIRE(w && img && patch_embed_supported(e0), LSAM_ERR_BAD_ARG, "pack_patch_embed: bad argument");
    if (e0 <= 112) hipLaunchKernelGGL(pack_patch_kernel<7>, dim3(32), dim3(256), 0, s, w, e0, static_cast<char*>(img));
    else hipLaunchKernelGGL(pack_patch_kernel<9>, dim3(32), dim3(256), 0, s, w, e0, static_cast<char*>(img));
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

template <int NT> static int launch_pe(const float* frames, const void* wimg, const float* bias, const float* pos, float* out,
                                       int N, int S, int e0, hipStream_t s) {
    constexpr int LDSB = ((PE_TILE_B + 1023) & ~1023) + NT * PE_KS * 1024;
    auto kern = patch_embed_kernel<NT>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        LSAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        attr_set = true;
    }
    const int G = S / 4;
    const int ntiles = N * cdiv(G, PE_TX) * cdiv(G, PE_TY);
    const int grid = ntiles < 512 ? ntiles : 512;   // persistent, two workgroups per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDSB, s, frames, static_cast<const char*>(wimg), bias, pos, out, N, S, e0);
    LSAM_LAUNCH_CHECK();
    return LSAM_OK;
}

int launch_patch_embed(const float* frames, const void* wimg, const float* bias, const float* pos, float* out, int N, int S,
                       int e0, hipStream_t s) {
    LSAM_REQUIRE(frames && wimg && bias && pos && out, LSAM_ERR_BAD_ARG, "patch_embed: null argument");
    LSAM_REQUIRE(N > 0 && S >= 4 && S % 4 == 0 && patch_embed_supported(e0) && e0 % 4 == 0, LSAM_ERR_BAD_ARG,
                 "patch_embed: bad shape");
    LSAM_REQUIRE(aligned16(wimg) && aligned16(bias) && aligned16(pos) && aligned16(out) &&
                     (reinterpret_cast<uintptr_t>(frames) & 3) == 0, LSAM_ERR_BAD_ARG, "patch_embed: alignment");
    ProfScope ps(s, (double)N * (3.0 * S * S * 4 + (double)(S / 4) * (S / 4) * e0 * 4), PROF_PATCH_EMBED, N, S, e0);
    if (e0 <= 112) return launch_pe<7>(frames, wimg, bias, pos, out, N, S, e0, s);
    return launch_pe<9>(frames, wimg, bias, pos, out, N, S, e0, s);
}

}  // namespace lsam
