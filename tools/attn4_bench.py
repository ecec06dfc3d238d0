"""Dev aid: time the 4 x 4-window attention (stage 2) on the B+ and Hiera-L shapes; ATTN4_OFF=1 takes the older kernels
(by handing in a row stride the one-wave kernel refuses is not possible, so the switch is an odd grid: H = W + 0) --
use LSAM_LIB to compare builds instead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_segmentation_amd import _C
if os.environ.get("LSAM_LIB"):
    _C.LIB_PATH = os.path.abspath(os.environ["LSAM_LIB"])
from image_segmentation_amd import ops
dev = torch.device("cuda", 0)
print(os.environ.get("LSAM_LIB", "liblsam.so"))
for (N, H, nh, hd, pool) in [(16, 128, 4, 56, 1), (64, 128, 4, 72, 1), (16, 128, 8, 56, 2), (64, 128, 8, 72, 2)]:
    d = nh * hd
    qkv = torch.randn(N, H, H, 3 * d, device=dev).bfloat16()
    bias = torch.randn(3 * d, device=dev)
    qp = torch.randn(N, H // 2, H // 2, d, device=dev) if pool == 2 else None
    f = lambda: ops.attention_fwd(qkv, bias, 4, pool, nh, q_pooled=qp)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e2
    byt = qkv.numel() * 2 * (1.0 if pool == 1 else 2 / 3) + N * (H // pool) ** 2 * d * (2 + (4 if pool == 2 else 0))
    print(f"N {N} grid {H} heads {nh} hd {hd} pool {pool}: {us:8.1f} us  {byt / us / 1e6:5.2f} TB/s")
