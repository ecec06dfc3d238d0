"""Summarise a rocprofv3 kernel_trace.csv: per kernel (and grid) totals per step. usage: prof_summary.py trace.csv nsteps"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
g = collections.OrderedDict()
tot = 0
for r in rows:
    n = r['Kernel_Name']
    n = re.sub(r'\(.*', '', n)[:70]
    key = (n, r['Grid_Size_X'], r['Workgroup_Size_X'])
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    g.setdefault(key, []).append(d)
    tot += d
items = sorted(g.items(), key=lambda kv: -sum(kv[1]))
print(f"total GPU ms/step {tot/nsteps/1e6:.2f}")
for k, v in items[:45]:
    print(f"{sum(v)/nsteps/1e6:8.3f} ms/step  n/step {len(v)/nsteps:6.1f}  avg {sum(v)/len(v)/1e3:8.1f} us  {k}")
