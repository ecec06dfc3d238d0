"""Dev aid: time the 14x14-window attention (attnw) on the stage-3 shape only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_segmentation_amd import _C
if os.environ.get("LSAM_LIB"):
    _C.LIB_PATH = os.path.abspath(os.environ["LSAM_LIB"])
from image_segmentation_amd import ops
dev = torch.device("cuda", 0)
H, ws, pool, nh = 64, 14, 1, 8
d = nh * 56
qkv = torch.randn(16, H, H, 3 * d, device=dev).bfloat16()
bias = torch.randn(3 * d, device=dev)
for _ in range(3):
    ops.attention_fwd(qkv, bias, ws, pool, nh)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
R = 20
e0.record()
for _ in range(R):
    ops.attention_fwd(qkv, bias, ws, pool, nh)
e1.record()
torch.cuda.synchronize()
print(f"{os.environ.get('LSAM_LIB', 'liblsam.so')}: attnw {e0.elapsed_time(e1) * 1e3 / R:7.1f} us")
