#!/bin/bash
# round 5, small grids: column-solve kernel choices at ne15 (BASELINE config 2) and at the 4- / 8-rank shares of ne30 (GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r5_small_${1:-a}.txt; : > $out
for rep in 1 2; do
for v in "TMX_VI_PRODUCERS=1" "TMX_VI_PRODUCERS=2" "TMX_VI_GROUP=1"; do
	env $v python3 bench.py --ne 15 --ppd 1 --steps 20 --warmup 3 --develop 3 --no-cpu-baseline --sustained-steps 0 2>>gpurun_out/r5_small.err | \
		python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('ne15 %-22s step %.3f ms | ' % ('$v', d['ms_per_step']) + ' '.join('%s %.3f' % (a, b) for a, b in sorted(k.items()) if b > 0))" >> $out
done
done
for v in "TMX_VI_GROUP=1" "TMX_VI_GROUP=0,TMX_VI_PRODUCERS=1" "TMX_VI_GROUP=0,TMX_VI_PRODUCERS=2"; do
	echo "rank shares with $v" >> $out
	env $(echo $v | tr ',' ' ') python3 tools/rank_share_timing.py 4 6 8 2>>gpurun_out/r5_small.err | grep "^ranks" >> $out
done
cat $out
