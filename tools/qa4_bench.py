"""Dev aid: time the fused 4x4-window attention kernels at config 1's geometry (stage-2 blocks 3 / 4 and block 5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_segmentation_amd import _C
if os.environ.get("LSAM_LIB"):
    _C.LIB_PATH = os.path.abspath(os.environ["LSAM_LIB"])
from image_segmentation_amd import ops

def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

N, H, W, dim = 16, 128, 128, 224
xn = torch.randn(N, H, W, dim, device="cuda").bfloat16()
for nh, pooled in ((4, False), (8, True)):
    q = nh * 56
    w = torch.randn(3 * q, dim, device="cuda") / dim ** 0.5
    b = torch.randn(3 * q, device="cuda")
    img = ops.qkv_window4_attention_pack(w, b, nh, pooled_q=pooled)
    qp = torch.randn(N, H // 2, W // 2, q, device="cuda") if pooled else None
    us = timeit(lambda: ops.qkv_window4_attention_fwd(xn, img, nh, 56, q_pooled=qp))
    print(f"qkv_window4_attention nh {nh} pooled {pooled}: {us:.1f} us")
