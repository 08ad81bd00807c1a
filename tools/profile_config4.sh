#!/bin/bash
# kernel durations of BASELINE config 4's shape (tools/config4_timing.py) on the GPU box: tools/profile_config4.sh <tag>
set -u
TAG=${1:-r04_config4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/config4_timing.py 2>&1 | grep "supercell shape"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o c4 -- python3 $R/tools/config4_timing.py > $OUT/${TAG}_trace.log 2>&1
f=$(find $OUT/${TAG}_trace -name "c4_kernel_stats.csv" | head -1)
cp $f $OUT/${TAG}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print("%-84s calls %5s avg_us %8.1f pct %5.1f" % (r["Name"][:84], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/tot*100))
PY
