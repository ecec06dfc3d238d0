"""Dev aid: what the vendor GEMM (torch -> hipBLASLt / rocBLAS) does on the trunk's plain Linear shapes, beside lsam_linear."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_segmentation_amd import _C
if os.environ.get("LSAM_LIB"):
    _C.LIB_PATH = os.path.abspath(os.environ["LSAM_LIB"])
from image_segmentation_amd import ops

def timeit(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

SHAPES_L = ((262144, 1728, 576), (262144, 2304, 576), (262144, 576, 2304), (262144, 576, 576), (1048576, 1152, 288), (1048576, 864, 288),
            (4194304, 432, 144))
for M, N, K in SHAPES_L if os.environ.get("BP_L") else ((65536, 1344, 448), (65536, 1792, 448), (65536, 448, 1792), (65536, 448, 448), (262144, 672, 224), (16384, 3584, 896),
                (16384, 896, 3584), (16384, 2688, 896), (1048576, 448, 112)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5)
    wb = w.bfloat16()
    b = torch.randn(N, device="cuda")
    bb = b.bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_t = timeit(lambda: torch.addmm(bb, x, wb.t(), out=out))
    t_m = timeit(lambda: torch.mm(x, wb.t(), out=out))
    wp = ops.pack_matrix(w, torch.bfloat16)
    t_l = timeit(lambda: ops.linear(x, wp, b, K=K, out=out))
    fl = 2.0 * M * N * K
    print(f"{M:8d} x {N:5d} x {K:5d}: torch addmm {t_t:7.1f} us ({fl / t_t / 1e6:6.0f} TF/s)  mm {t_m:7.1f} us ({fl / t_m / 1e6:6.0f})   lsam_linear {t_l:7.1f} us ({fl / t_l / 1e6:6.0f} TF/s)", flush=True)
