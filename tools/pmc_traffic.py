"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, the TCC block cannot hold
both) into HBM bytes per launch of one kernel.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> <out.json>

Units and corrections follow MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950
FETCH_SIZE reports half the bytes of a wide coalesced streaming read, so the read side is doubled.
"""
import csv
import json
import sys


def per_launch(path, counter, kernel):
    tot, n = 0.0, 0
    per = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if kernel not in r["Kernel_Name"] or r["Counter_Name"] != counter:
                continue
            key = r.get("Dispatch_Id") or r.get("Correlation_Id")
            per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
    for v in per.values():
        tot += v
        n += 1
    return (tot / n if n else 0.0), n


def main():
    fetch_csv, write_csv, kernel, out = sys.argv[1:5]
    f_kib, nf = per_launch(fetch_csv, "FETCH_SIZE", kernel)
    w_kib, nw = per_launch(write_csv, "WRITE_SIZE", kernel)
    res = {
        "kernel": kernel, "launches_fetch_pass": nf, "launches_write_pass": nw,
        "FETCH_SIZE_KiB_per_launch_raw": round(f_kib, 1), "WRITE_SIZE_KiB_per_launch": round(w_kib, 1),
        "read_bytes_per_launch": int(2 * f_kib * 1024), "write_bytes_per_launch": int(w_kib * 1024),
        "hbm_bytes_per_launch": int((2 * f_kib + w_kib) * 1024),
        "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), KiB -> bytes",
    }
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
