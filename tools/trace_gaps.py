"""Dev aid: launch-gap accounting of the LAST `nsteps` bench steps in a rocprofv3 kernel_trace.csv.
usage: trace_gaps.py trace.csv launches_per_step [nsteps]
Prints per step: wall (first start .. last end), busy (sum of kernel durations), idle (gaps between consecutive kernels), and the
non-liblsam kernels (torch's: loss, optimizer, copies) with their time."""
import collections, csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[2])
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = rows[-per * nsteps:]
wall = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
pos = [g for g in gaps if g > 0]
print(f"{len(rows)} kernels over {nsteps} step(s): wall {wall/nsteps/1e6:.3f} ms/step, busy {busy/nsteps/1e6:.3f}, idle {sum(pos)/nsteps/1e6:.3f} "
      f"({len(pos)} gaps, median {sorted(pos)[len(pos)//2]/1e3:.2f} us, max {max(pos)/1e3:.1f} us), overlap {-sum(g for g in gaps if g < 0)/nsteps/1e6:.3f}")
other = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if "lsam" not in r["Kernel_Name"]:
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:90]
        other[k][0] += 1
        other[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in other.values())
print(f"non-liblsam kernels: {sum(v[0] for v in other.values())/nsteps:.1f} per step, {tot/nsteps/1e6:.3f} ms/step")
for k, v in sorted(other.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {v[1]/nsteps/1e3:8.1f} us/step  n/step {v[0]/nsteps:5.1f}  {k}")
big = sorted(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"]), a["Kernel_Name"][:50], b["Kernel_Name"][:50]) for a, b in zip(rows, rows[1:])), reverse=True)[:12]
for g, a, b in big:
    print(f"  gap {g/1e3:7.1f} us after {a} before {b}")
