"""Dev aid: the Hiera-L stage-3 window attention (16 x 16 windows, 8 heads of 72) and stage-1 (8 x 8, 2 heads) shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_segmentation_amd import _C
if os.environ.get("LSAM_LIB"):
    _C.LIB_PATH = os.path.abspath(os.environ["LSAM_LIB"])
from image_segmentation_amd import ops
dev = torch.device("cuda", 0)
print(os.environ.get("LSAM_LIB", "liblsam.so"))
for (N, H, ws, nh, hd) in [(64, 64, 16, 8, 72), (64, 64, 0, 8, 72), (16, 256, 8, 2, 72)]:
    d = nh * hd
    qkv = torch.randn(N, H, H, 3 * d, device=dev).bfloat16(); bias = torch.randn(3 * d, device=dev)
    f = lambda: ops.attention_fwd(qkv, bias, ws, 1, nh)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"N {N} grid {H} ws {ws} heads {nh} hd {hd}: {e0.elapsed_time(e1) * 100:8.1f} us")
