"""Dev aid: Hiera-L GEMM shapes, bf16 (LSAM_LIB=... selects an alternative build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_segmentation_amd import _C
if os.environ.get("LSAM_LIB"):
    _C.LIB_PATH = os.path.abspath(os.environ["LSAM_LIB"])
from image_segmentation_amd import ops
dev = torch.device("cuda", 0)
print(os.environ.get("LSAM_LIB", "liblsam.so"))
for (M, N, K, f32) in [(262144, 1728, 576, False), (262144, 576, 576, True), (262144, 2304, 576, False), (262144, 576, 2304, True),
                       (65536, 3456, 1152, False), (65536, 1152, 1152, True), (65536, 4608, 1152, False), (65536, 1152, 4608, True)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(); b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev) if f32 else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    f = lambda: ops.linear(A, W, b, res=res, out_f32=f32, out=out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{M}x{N}x{K} {'f32+res' if f32 else 'bf16':8s} {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s")
