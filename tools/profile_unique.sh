#!/bin/bash
# A/B of the node-unique state layout against the element-major one on the bench workload (runs on the GPU box):
# bench lines, kernel-trace stats and the FETCH / WRITE counter passes of both.   usage: tools/profile_unique.sh <tag>
set -u
TAG=${1:-r04u}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained-steps 0"
for v in ${VARIANTS:-"d TMX_UNIQUE=0" "u0 TMX_UNIQUE=1 TMX_UNIQUE_TILE=0" "u1 TMX_UNIQUE=1 TMX_UNIQUE_TILE=1" "u2 TMX_UNIQUE=1 TMX_UNIQUE_TILE=2" "d_b TMX_UNIQUE=0" "u2_b TMX_UNIQUE=1 TMX_UNIQUE_TILE=2"}; do
	set -- $v; name=$1; shift
	env "$@" python3 $B > $OUT/${TAG}_bench_$name.json 2> $OUT/${TAG}_bench_$name.err
	python3 -c "import json,sys; d=json.load(open('$OUT/${TAG}_bench_$name.json')); print('$name', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms_per_step'].items() if v>0})"
done
P="$R/bench.py --steps 5 --warmup 2 --develop 3 --no-cpu-baseline --sustained-steps 0"
for m in 0 1; do
	export TMX_UNIQUE=$m
	rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_u$m -o ${TAG} -- python3 $P > $OUT/${TAG}_trace_u$m.log 2>&1
	rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch_u$m -o ${TAG} -- python3 $P > $OUT/${TAG}_fetch_u$m.log 2>&1
	rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write_u$m -o ${TAG} -- python3 $P > $OUT/${TAG}_write_u$m.log 2>&1
done
find $OUT -name "${TAG}_kernel_stats.csv" | head
