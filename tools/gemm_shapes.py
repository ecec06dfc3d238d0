"""Per-shape GEMM timing inside one fwd+bwd step (dev aid): python tools/gemm_shapes.py [batch]"""
import collections, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from image_segmentation_amd import _C
from image_segmentation_amd.synth import synth_frames_device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
enc = bench.build(dev)
img = synth_frames_device(B, 1024, 0, dev)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = enc(img)
    bench.fpn_loss(out).backward()
for _ in range(3): step()
torch.cuda.synchronize()
cap = 1200
_C.check(_C.lib().lsam_prof_arm(cap))
for _ in range(3): step()
torch.cuda.synchronize()
ms=(C.c_float*cap)(); fl=(C.c_double*cap)(); cfg=(C.c_int*cap)(); mnk=(C.c_int*(3*cap))()
n=_C.lib().lsam_prof_read(cap, ms, fl, cfg, mnk)
g=collections.OrderedDict()
for i in range(n):
    g.setdefault((cfg[i], mnk[3*i], mnk[3*i+1], mnk[3*i+2]), []).append(ms[i])
tot=0
print(f"{'cfg':>5} {'M':>8} {'N':>5} {'K':>5} {'cnt/step':>8} {'us':>8} {'TF/s':>7} {'ms/step':>8}")
for (c,m,nn,k),v in g.items():
    us=1e3*sum(v)/len(v); tf=2.0*m*nn*k/(us*1e-6)/1e12
    tot+=sum(v)/3
    print(f"{c:>5} {m:>8} {nn:>5} {k:>5} {len(v)/3:>8.1f} {us:>8.1f} {tf:>7.1f} {sum(v)/3:>8.3f}")
print("total gemm ms/step", round(tot,3))
