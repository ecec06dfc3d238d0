"""Dev aid: per-kernel averages of the counters in a rocprofv3 --pmc counter_collection.csv.  usage: pmc_sq.py <csv> [name-substring]"""
import collections, csv, sys
sub = sys.argv[2] if len(sys.argv) > 2 else "lsam"
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    if sub not in r["Kernel_Name"]:
        continue
    a = agg[r["Kernel_Name"][:60]][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print(f"    {c:32s} {v / n:16.0f}  (n={n})")
