"""Dev aid: time lsam_linear on the trunk's GEMM shapes, optionally against an ablated build
(LSAM_LIB=path/to/liblsam_variant.so python tools/gemm_abl.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_segmentation_amd import _C
if os.environ.get("LSAM_LIB"):
    _C.LIB_PATH = os.path.abspath(os.environ["LSAM_LIB"])
from image_segmentation_amd import ops
dev = torch.device("cuda", 0)
shapes = [(65536, 448, 448), (65536, 1344, 448), (65536, 1792, 448), (65536, 448, 1792),
          (16384, 896, 3584), (16384, 3584, 896), (262144, 224, 224), (262144, 896, 224), (262144, 224, 896),
          (16384, 896, 896), (16384, 2688, 928), (16384, 896, 2720), (16384, 896, 256), (262144, 672, 224), (262144, 448, 224)]
if os.environ.get("GEMM_SHAPES"):
    shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ["GEMM_SHAPES"].split(",")]
print(os.environ.get("LSAM_LIB", "liblsam.so"))
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).bfloat16()
    W = torch.randn(N, K, device=dev).bfloat16()
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.linear(A, W, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 20
    e0.record()
    for _ in range(R):
        ops.linear(A, W, b, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / R
    print(f"{M:>7} {N:>5} {K:>5}  {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s  "
          f"{(M*K+M*N)*2/us/1e6:6.2f} TB/s(A+out)")
