#!/bin/bash
# 8-rank-share passes of tools/profile_round.sh on their own (lone rank-0 engine of the 8-rank layout, wire left out)
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
S8="$R/tools/rank_share_timing.py 8"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch_n8 -o ${TAG} -- python3 $S8 > $OUT/${TAG}_fetch_n8.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write_n8 -o ${TAG} -- python3 $S8 > $OUT/${TAG}_write_n8.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_n8 -o ${TAG} -- python3 $S8 > $OUT/${TAG}_trace_n8.log 2>&1
