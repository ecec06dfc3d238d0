"""Dev aid: where does the fused tail differ from a torch reference (rows / columns)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from image_segmentation_amd import _C
if os.environ.get('LSAM_LIB'):
    _C.LIB_PATH = os.path.abspath(os.environ['LSAM_LIB'])
from image_segmentation_amd import ops
torch.manual_seed(0)
d, M = 448, int(os.environ.get("M", 128))
dev = "cuda"
r = lambda *s: torch.randn(*s, device=dev)
res, a = r(M, d), r(M, d).bfloat16()
wp, bp = r(d, d) / math.sqrt(d), 0.2 * r(d)
g2, b2 = 1 + 0.1 * r(d), 0.1 * r(d)
w1, c1 = r(4 * d, d) / math.sqrt(d), 0.2 * r(4 * d)
w2, c2 = r(d, 4 * d) / math.sqrt(4 * d), 0.2 * r(d)
y = res + a.float() @ wp.t() + bp
ref = y + F.gelu(F.layer_norm(y, (d,), g2, b2, 1e-6) @ w1.t() + c1) @ w2.t() + c2
img, vec = ops.tail_block_pack(wp, bp, g2, b2, w1, c1, w2, c2)
for rep in range(3):
    out = ops.tail_block_fwd(a, res, img, vec)
    torch.cuda.synchronize()
    bad = ~torch.isfinite(out)
    err = (out - ref).abs()
    err[bad] = 1e9
    rows = (err > 0.1).any(1).nonzero().flatten().tolist()
    cols = (err > 0.1).any(0).nonzero().flatten().tolist()
    print(f"rep {rep}: nan {int(bad.sum())} bad rows {len(rows)} {rows[:40]} bad cols {len(cols)} {cols[:20]}..{cols[-5:]}")
