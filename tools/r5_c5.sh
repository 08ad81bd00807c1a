#!/bin/bash
# BASELINE config 5's shape (ne120 np4 L60, Held-Suarez forcing after every step) with and without "unique_mixed" (GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r5_c5_${1:-a}.txt; : > $out
for v in ${VARIANTS:-1 0}; do
	TMX_UNIQUE_MIXED=$v timeout 1200 python3 bench.py --ne 120 --levels 60 --dt 37.5 --physics hs --steps 3 --warmup 1 --develop 2 --no-cpu-baseline --sustained-steps 0 2>>gpurun_out/r5_c5.err | \
		python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('unique_mixed=$v step %.3f ms | ' % d['ms_per_step'] + ' '.join('%s %.3f' % (a, b) for a, b in sorted(k.items()) if b > 0), '|', d['config']['state_layout'][:60])" >> $out
done
cat $out
