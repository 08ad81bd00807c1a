#!/bin/bash
# ne15 (BASELINE configs[2] shape) passes of tools/profile_round.sh on their own
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B15="$R/bench.py --ne 15 --ppd 1 --steps 5 --warmup 2 --develop 3 --no-cpu-baseline --sustained-steps 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch_ne15 -o ${TAG} -- python3 $B15 > $OUT/${TAG}_fetch_ne15.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write_ne15 -o ${TAG} -- python3 $B15 > $OUT/${TAG}_write_ne15.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_ne15 -o ${TAG} -- python3 $B15 > $OUT/${TAG}_trace_ne15.log 2>&1
