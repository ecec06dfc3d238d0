"""Dev aid: time the fused k|v projection + 8x8-window attention on Hiera-B+ block 2's shape (LSAM_LIB=... selects a build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_segmentation_amd import _C
if os.environ.get("LSAM_LIB"):
    _C.LIB_PATH = os.path.abspath(os.environ["LSAM_LIB"])
from image_segmentation_amd import ops
dev = torch.device("cuda", 0)
N, H, dim, nh, hd = 16, 256, 112, 4, 56
q = nh * hd
xn = torch.randn(N, H, H, dim, device=dev).bfloat16()
qp = torch.randn(N, H // 2, H // 2, q, device=dev)
w = torch.randn(3 * q, dim, device=dev) / dim ** 0.5
b = torch.randn(3 * q, device=dev)
img = ops.kv_window_attention_pack(w, b, nh)
for _ in range(3):
    ops.kv_window_attention_fwd(xn, qp, img, nh)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
R = 20
e0.record()
for _ in range(R):
    ops.kv_window_attention_fwd(xn, qp, img, nh)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / R
flop = 2.0 * N * H * H * dim * 2 * q + 4.0 * N * (H // 8) ** 2 * 16 * 64 * q
byt = xn.numel() * 2 + qp.numel() * 4 + qp.numel() * 2
print(f"{os.environ.get('LSAM_LIB', 'liblsam.so')}: {us:7.1f} us  {flop / us / 1e6:6.1f} TF/s  {byt / us / 1e3:6.1f} GB/s")
