"""Per-kernel HBM counters from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs).
usage: pmc_per_kernel.py <fetch_counter_collection.csv> <write_counter_collection.csv>"""
import collections, csv, sys


def load(path, counter):
    per = collections.defaultdict(float)
    name = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r.get("Dispatch_Id") or r.get("Correlation_Id")
        per[k] += float(r["Counter_Value"])
        name[k] = r["Kernel_Name"]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for k, v in per.items():
        agg[name[k]][0] += 1
        agg[name[k]][1] += v
    return agg


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1`;")
print("KiB per launch, raw counters (FETCH_SIZE must be doubled on gfx950, see tools/pmc_traffic.py)")
print("launches    FETCH_KiB    WRITE_KiB  kernel")
for n in sorted(f, key=lambda n: -(2 * f[n][1] + w.get(n, [0, 0.0])[1])):
    if not n.startswith(("void lsam", "lsam", "_ZN4lsam")):
        continue
    c, fv = f[n]
    wv = w[n][1] / max(1, w[n][0]) if n in w else 0.0
    print(f"{c:8d} {fv / c:12.1f} {wv:12.1f}  {n[:150]}")
