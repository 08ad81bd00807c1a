#!/bin/bash
# Memory-path counters of the explicit / hyperviscosity / DSS kernels, element-major against node-unique layout (GPU box).
#   usage: tools/profile_unique_mem.sh <tag> [ENV=VAL ...]      e.g. TMX_UNIQUE=1 TMX_UNIQUE_TILE=0
set -u
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 5 --warmup 2 --develop 3 --no-cpu-baseline --sustained-steps 0"
RX="k_h_explicit|k_hypervis|k_dss|k_h_w_update"
one() { local name=$1; shift; timeout 240 rocprofv3 "$@" --kernel-include-regex "$RX" --output-format csv -d $OUT/${TAG}_${name} -o ${TAG} -- python3 $B > $OUT/${TAG}_${name}.log 2>&1 || echo "$name failed"; }
one rd --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
one wr --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
one l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
one l1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_READ_sum
if [ "${FULL:-0}" = 1 ]; then      # (a request for more counters than the hardware has makes rocprofv3 abort and then hang: every pass runs under `timeout`)
one ta --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
one tas --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
one tcps --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
one sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
fi
python3 - <<PY
import csv, collections, os
G="$OUT"; tag="$TAG"
tab=collections.defaultdict(dict)
for name in ("rd","wr","l2","l1","ta","tas","tcps","sq"):
    p=os.path.join(G,tag+"_"+name,tag+"_counter_collection.csv")
    if not os.path.exists(p): print(name,"missing"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].replace("void ","").split("(")[0]
        a=acc[k][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,cs in acc.items():
        for c,(n,v) in cs.items(): tab[k][c]=v/n
with open(os.path.join(G,tag+"_mem_counters.txt"),"w") as f:
    for k in sorted(tab):
        line=k+"\n   "+"  ".join("%s=%.4g" % (c.replace("_sum",""),v) for c,v in sorted(tab[k].items()))
        t=tab[k]
        if "TCC_EA0_RDREQ_32B_sum" in t:
            n32,n64,n128=t["TCC_EA0_RDREQ_32B_sum"],t["TCC_EA0_RDREQ_64B_sum"],t.get("TCC_EA0_RDREQ_128B_sum",0)
            tot=t["TCC_EA0_RDREQ_sum"]
            line+="\n   read bytes (32/64/128-B requests, rest of RDREQ taken as 64 B): %.1f MB" % ((32*n32+64*n64+128*n128+64*max(tot-n32-n64-n128,0))/1e6)
        print(line); f.write(line+"\n")
PY
