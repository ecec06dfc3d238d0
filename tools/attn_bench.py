"""Dev aid: time lsam attention forward on the trunk's shapes (LSAM_LIB=... selects an alternative build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_segmentation_amd import _C
if os.environ.get("LSAM_LIB"):
    _C.LIB_PATH = os.path.abspath(os.environ["LSAM_LIB"])
from image_segmentation_amd import ops
dev = torch.device("cuda", 0)
print(os.environ.get("LSAM_LIB", "liblsam.so"))
# (grid, window, q-pool, heads): global stage-3 blocks, 14x14 windows, stage-2 8x8->4x4 windows, transitions
cases = [(64, 0, 1, 8), (64, 14, 1, 8), (128, 4, 1, 4), (256, 8, 2, 4), (128, 4, 2, 8), (64, 14, 2, 16)]
for (H, ws, pool, nh) in cases:
    d = nh * 56
    qkv = torch.randn(16, H, H, 3 * d, device=dev).bfloat16()
    bias = torch.randn(3 * d, device=dev)
    for _ in range(3):
        ops.attention_fwd(qkv, bias, ws, pool, nh)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 10
    e0.record()
    for _ in range(R):
        ops.attention_fwd(qkv, bias, ws, pool, nh)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / R
    nk = (H * H) if ws == 0 else ws * ws
    nq_tot = 16 * (H // pool) ** 2
    print(f"grid {H:>3} ws {ws:>2} pool {pool} heads {nh:>2}: {us:8.1f} us  {4.0 * nq_tot * nk * 56 * nh / us / 1e6:7.1f} TF/s  "
          f"{qkv.numel() * 2 / us / 1e6:5.2f} TB/s(qkv)")
