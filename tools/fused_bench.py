"""Dev aid: time the fused stage-1/2 kernels standalone (LSAM_LIB=... selects an alternative build)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_segmentation_amd import _C
if os.environ.get("LSAM_LIB"):
    _C.LIB_PATH = os.path.abspath(os.environ["LSAM_LIB"])
from image_segmentation_amd import ops
dev = torch.device("cuda", 0)
def timeit(fn, R=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / R
print(os.environ.get("LSAM_LIB", "liblsam.so"))
ONLY_TAIL = bool(os.environ.get("FB_TAIL"))
ONLY_MLP = bool(os.environ.get("FB_MLP"))
for d, M in ([] if ONLY_TAIL else [(112, 16 * 256 * 256), (224, 16 * 128 * 128)]):
    x = torch.randn(M, d, device=dev)
    g, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    img, bias = ops.mlp_block_pack(torch.randn(4 * d, d, device=dev) / math.sqrt(d), torch.randn(4 * d, device=dev),
                                   torch.randn(d, 4 * d, device=dev) / math.sqrt(4 * d), torch.randn(d, device=dev))
    y = torch.empty_like(x)
    us = timeit(lambda: ops.mlp_block_fwd(x, g, b, img, bias, out=y))
    print(f"mlp_block d={d} M={M}: {us:8.1f} us  {16.0*M*d*d/us/1e6:7.1f} TF/s  {M*d*8/us/1e6:5.2f} TB/s")
for nh, H in ([] if ONLY_TAIL or ONLY_MLP else [(2, 256)]):
    d = nh * 56
    x = torch.randn(16, H, H, d, device=dev)
    g, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    img, bias = ops.window_block_pack(torch.randn(3 * d, d, device=dev) / math.sqrt(d), torch.randn(3 * d, device=dev),
                                      torch.randn(d, d, device=dev) / math.sqrt(d), torch.randn(d, device=dev), nh)
    y = torch.empty_like(x)
    us = timeit(lambda: ops.window_block_fwd(x, g, b, img, bias, nh, out=y))
    print(f"window_block nh={nh} grid={H}: {us:8.1f} us  {x.numel()*8/us/1e6:5.2f} TB/s")

for d, M in ([] if ONLY_MLP else [(448, 16 * 64 * 64), (224, 16 * 128 * 128)]):
    res = torch.randn(M, d, device=dev)
    a = torch.randn(M, d, device=dev).bfloat16()
    r = lambda *sh: torch.randn(*sh, device=dev)
    img, vec = ops.tail_block_pack(r(d, d) / math.sqrt(d), r(d), torch.ones(d, device=dev), torch.zeros(d, device=dev),
                                   r(4 * d, d) / math.sqrt(d), r(4 * d), r(d, 4 * d) / math.sqrt(4 * d), r(d))
    y = torch.empty_like(res)
    nn_ = (torch.ones(d, device=dev), torch.zeros(d, device=dev))
    for name, kw in (("tail", {}), ("tail+nextLN", {"next_norm": nn_})):
        us = timeit(lambda: ops.tail_block_fwd(a, res, img, vec, out=y, **kw))
        print(f"{name} d={d} M={M}: {us:8.1f} us  {18.0*M*d*d/us/1e6:7.1f} TF/s")
