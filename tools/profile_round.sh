#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats and the two PMC passes of the bench command.
# usage: tools/profile_round.sh <tag>        outputs under gpurun_out/<tag>_*
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o ${TAG} -- $CMD > $OUT/${TAG}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o ${TAG} -- $CMD > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o ${TAG} -- $CMD > $OUT/${TAG}_write.log 2>&1
ls -R $OUT | grep -i csv | head -20
