#!/bin/bash
# round 5: ring depth 2 / 3 of the column kernel on grids of 60 levels (GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r5_l60_${1:-a}.txt; : > $out
for rep in 1 2; do
for v in 3 0; do
	TMX_VI_RING_DEPTH=$v python3 bench.py --ne 30 --levels 60 --dt 75 --steps 10 --warmup 2 --develop 3 --no-cpu-baseline --sustained-steps 0 2>>gpurun_out/r5_l60.err | \
		python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('ne30 L60 ring depth $v (0 = auto) step %.3f ms | ' % d['ms_per_step'] + ' '.join('%s %.3f' % (a, b) for a, b in sorted(k.items()) if b > 0))" >> $out
done
done
for v in 3 0; do
	TMX_VI_RING_DEPTH=$v python3 bench.py --ne 60 --levels 60 --dt 75 --steps 5 --warmup 2 --develop 2 --no-cpu-baseline --sustained-steps 0 2>>gpurun_out/r5_l60.err | \
		python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('ne60 L60 ring depth $v (0 = auto) step %.3f ms | ' % d['ms_per_step'] + ' '.join('%s %.3f' % (a, b) for a, b in sorted(k.items()) if b > 0))" >> $out
done
cat $out
