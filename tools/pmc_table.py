#!/usr/bin/env python3
"""Build profiles/<round>_pmc_traffic.json from two rocprofv3 counter-collection CSVs (FETCH_SIZE pass and
WRITE_SIZE pass of the same bench command).  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE
counts half of the bytes our 8-byte-per-lane coalesced reads move (calibrated on k_lincomb, see `method`), WRITE_SIZE
is exact (MI355X_MICROARCH.md, HBM / rocprofv3 section).
usage: pmc_table.py fetch.csv write.csv steps out.json [workload text]"""
import csv, json, sys, collections

def load(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc

fetch = load(sys.argv[1], "FETCH_SIZE"); write = load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
out = {"steps": steps, "kernels": {},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of `python3 bench.py --steps S "
                 "--warmup W --no-cpu-baseline` (" + (sys.argv[5] if len(sys.argv) > 5 else "ne30 np4 L30, 1 GPU") + "; `steps` = all "
                 "steps the run executed: development, warm-up, timed and profiled pass), averages per launch.  Units KiB.  WRITE_SIZE is exact in our access pattern, "
                 "FETCH_SIZE counts 1/2 of the bytes (calibrated on a plain copy kernel: r01 k_lincomb wrote 101925.0 KiB "
                 "= one state instance exactly and read 5 instances = 509625 KiB while FETCH_SIZE said 254826 KiB).  "
                 "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."}
for k in sorted(set(fetch) | set(write)):
    nf, f = fetch.get(k, [0, 0.0]); nw, w = write.get(k, [0, 0.0])
    n = max(nf, nw)
    if n == 0:
        continue
    fa = f / max(nf, 1); wa = w / max(nw, 1)
    out["kernels"][k] = {"launches": n, "FETCH_SIZE_KiB": fa, "WRITE_SIZE_KiB": wa,
                         "hbm_bytes_per_launch": (2.0 * fa + wa) * 1024.0}
json.dump(out, open(sys.argv[4], "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-70s n=%5d  %.1f MB/launch" % (k[:70], v["launches"], v["hbm_bytes_per_launch"] / 1e6))
