#!/bin/bash
# kernel durations of the bench workload with and without the stored stage-combination prefix (runs on the GPU box)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="$R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained-steps 0 --parity-steps 0"
for m in 0 1; do
	export TMX_UNIQUE_PREFIX=$m TMX_FLAVOUR=experiments
	timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prefix_trace_$m -o pfx -- python3 $P > $OUT/prefix_trace_$m.log 2>&1
	f=$(find $OUT/prefix_trace_$m -name "pfx_kernel_stats.csv" | head -1)
	echo "== unique_prefix=$m"
	python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-100s calls %6s avg_us %9.1f total_ms %9.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
