"""Summarise a rocprofv3 kernel_trace.csv of `bench.py --steps K --warmup W` (config 1): per kernel (and grid) totals per step over
the K TIMED steps only -- the W warm-up steps (with the one-off weight packing) and the live-profiled steps behind the timed
region are cut away by locating every step's first kernel (the PatchEmbed convolution).
usage: prof_summary.py trace.csv K W ["header line"]"""
import collections, csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
K, W = int(sys.argv[2]), int(sys.argv[3])
starts = [i for i, r in enumerate(rows) if "patch_embed_kernel" in r["Kernel_Name"]]
assert len(starts) > W + K, (len(starts), W, K)
sel = rows[starts[W]:starts[W + K]]
wall = int(rows[starts[W + K]]["Start_Timestamp"]) - int(rows[starts[W]]["Start_Timestamp"])
g = collections.OrderedDict()
tot = lsam = 0
for r in sel:
    n = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
    key = (n, r["Grid_Size_X"], r["Workgroup_Size_X"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    g.setdefault(key, []).append(d)
    tot += d
    if "lsam" in n:
        lsam += d
if len(sys.argv) > 4:
    print(sys.argv[4])
print(f"{K} timed steps ({len(sel) / K:.1f} kernels per step): first kernel to first kernel {wall / K / 1e6:.3f} ms/step; kernel durations "
      f"{tot / K / 1e6:.3f} ms/step, of which liblsam {lsam / K / 1e6:.3f} and torch (loss, optimizer, fills) {(tot - lsam) / K / 1e6:.3f}")
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:48]:
    print(f"{sum(v) / K / 1e6:8.3f} ms/step  n/step {len(v) / K:6.1f}  avg {sum(v) / len(v) / 1e3:8.1f} us  {k}")
