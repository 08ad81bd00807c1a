#!/bin/bash
# GPU box: kernel trace + memory counters of the column-solve kernels only (bench workload).  usage: tools/profile_vi.sh <tag> [ENV=VAL ...]
set -u
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 5 --warmup 2 --develop 3 --no-cpu-baseline --sustained-steps 0"
one() { local name=$1; shift; rocprofv3 "$@" > $OUT/${TAG}_${name}.log 2>&1 || echo "$name failed"; }
one trace --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o ${TAG} -- python3 $B
one fetch --pmc FETCH_SIZE --kernel-include-regex "k_vi" --output-format csv -d $OUT/${TAG}_fetch -o ${TAG} -- python3 $B
one write --pmc WRITE_SIZE --kernel-include-regex "k_vi" --output-format csv -d $OUT/${TAG}_write -o ${TAG} -- python3 $B
one l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-include-regex "k_vi" --output-format csv -d $OUT/${TAG}_l2 -o ${TAG} -- python3 $B
one sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-include-regex "k_vi" --output-format csv -d $OUT/${TAG}_sq -o ${TAG} -- python3 $B
python3 - <<PY
import csv, collections, os
G="$OUT"; tag="$TAG"
st=os.path.join(G,tag+"_trace",tag+"_kernel_stats.csv")
if os.path.exists(st):
    print("kernel stats (top 8):")
    for i,r in enumerate(csv.DictReader(open(st))):
        if i<8: print("  %-70s calls %5s avg_ns %10s pct %6s" % (r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"]))
for name in ("fetch","write","l2","sq"):
    p=os.path.join(G,tag+"_"+name,tag+"_counter_collection.csv")
    if not os.path.exists(p): print(name,"missing"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
    for r in csv.DictReader(open(p)):
        a=acc[r["Kernel_Name"][:40]][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,c in acc.items():
        print("  %-40s" % k, "  ".join("%s %.4g (n %d)" % (n, x[1]/x[0], x[0]) for n,x in sorted(c.items())))
PY
