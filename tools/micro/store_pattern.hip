// Dev aid: cost of a wave's 1 KiB store instruction (dwordx4 per lane) as a function of how many distinct rows it touches.
// R rows x (1024 / R) contiguous bytes per instruction, row stride `stride` bytes; 256 workgroups x 8 waves, each wave
// issues `iters` x 14 instructions walking along its rows like a GEMM epilogue (next 1024/R bytes of the same rows), then
// moves to the next block of rows.  Prints us and bytes/clk/CU.   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int R>
__global__ __launch_bounds__(512) void k(char* out, long stride, int iters, long total_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = 64 / R;                 // lanes per row
    const int r = lane / LPR, c = lane % LPR;
    uint4 v = make_uint4(lane, wave, blockIdx.x, 1);
    for (int it = 0; it < iters; ++it) {
        // block of R rows owned by this wave in this iteration
        const long row0 = (((long)it * gridDim.x + blockIdx.x) * 8 + wave) * R % total_rows;
        char* base = out + (row0 + r) * stride + c * 16;
#pragma unroll
        for (int j = 0; j < 14; ++j) {          // walk along the rows: 14 x (1024 / R) bytes per row
            if ((j + 1) * (1024 / R) <= stride) *reinterpret_cast<uint4*>(base + (long)j * (1024 / R)) = v;
        }
    }
}
int main(int argc, char** argv) {
    const long stride = argc > 1 ? atol(argv[1]) : 1344;   // bytes per output row
    const long coff = argc > 2 ? atol(argv[2]) : 0;         // byte offset of the wave's first column (alignment study)
    const long rows = 262144;
    char* d;
    hipMalloc(&d, rows * stride + 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int R, const char* name) {
        const int iters = 64;
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d + coff, stride, iters, rows - 64);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d + coff, stride, iters, rows - 64);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int per = 0; for (int j = 0; j < 14; ++j) if ((j + 1) * (1024 / R) <= stride) ++per;
        const double bytes = 256.0 * 8 * iters * per * 1024;
        printf("%-28s off %3ld stride %5ld: %8.1f us  %6.2f TB/s  %6.1f B/clk/CU (2.4 GHz)  %6.1f clk per store instr per CU\n", name, coff, stride,
               ms * 1e3, bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / (8.0 * iters * per));
    };
    run(k<1>, 1, "1 row  x 1024 B");
    run(k<2>, 2, "2 rows x 512 B");
    run(k<4>, 4, "4 rows x 256 B");
    run(k<8>, 8, "8 rows x 128 B");
    run(k<16>, 16, "16 rows x 64 B (gemm3 now)");
    run(k<32>, 32, "32 rows x 32 B");
    run(k<64>, 64, "64 rows x 16 B");
    return 0;
}
