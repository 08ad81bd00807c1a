#!/usr/bin/env python3
"""L2 hit rate and LDS bank-conflict share per kernel from the `l2` and `lds` passes of tools/profile_round.sh:
   tools/pmc_l2_lds_table.py <tag>  ->  profiles/<tag>_l2_lds_counters.txt   (averages per launch; kernels of the bench workload)"""
import csv, os, sys, glob, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
def counters(name):
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(R, "gpurun_out", "%s_%s" % (tag, name), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            c = out[r["Kernel_Name"]][r["Counter_Name"]]
            c[0] += 1; c[1] += float(r["Counter_Value"])
    return out
l2, lds = counters("l2"), counters("lds")
sha = ""
p = os.path.join(R, "gpurun_out", "%s_l2.sha" % tag)
if os.path.exists(p): sha = open(p).read().strip()
with open(os.path.join(R, "profiles", "%s_l2_lds_counters.txt" % tag), "w") as f:
    f.write("rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum  and  --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES (separate passes) -- python3 bench.py "
            "--steps 5 --warmup 2 --develop 3 --no-cpu-baseline (ne30 L30, 1 GPU); averages per launch; kernel sources %s.\n" % sha)
    f.write("l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS); lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (share of the LDS array's active cycles that are conflict cycles).\n\n")
    for k in sorted(set(l2) | set(lds), key=lambda k: -l2.get(k, {}).get("TCC_REQ_sum", [0, 0.0])[1]):
        a, b = l2.get(k, {}), lds.get(k, {})
        av = lambda d, n: (d[n][1] / d[n][0]) if n in d and d[n][0] else float("nan")
        n = (a.get("TCC_REQ_sum") or b.get("SQ_INSTS_LDS") or [0])[0]
        hit, miss = av(a, "TCC_HIT_sum"), av(a, "TCC_MISS_sum")
        bc, ia = av(b, "SQ_LDS_BANK_CONFLICT"), av(b, "SQ_LDS_IDX_ACTIVE")
        f.write("%-64s launches %4d  TCC_REQ %12.0f  l2_hit %.3f   LDS insts %10.0f  bank-conflict cycles %10.0f of %10.0f active  lds_conflict %.3f\n" %
                (k[:64], n, av(a, "TCC_REQ_sum"), hit / (hit + miss) if hit + miss > 0 else float("nan"), av(b, "SQ_INSTS_LDS"), bc, ia, bc / ia if ia > 0 else float("nan")))
print(open(os.path.join(R, "profiles", "%s_l2_lds_counters.txt" % tag)).read())
