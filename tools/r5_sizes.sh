#!/bin/bash
# one-GPU rate over problem sizes (what a weak-scaling run holds per GPU): ne30 / 42 / 60 / 84 at L30, final code of round 5 (GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r5_sizes.txt; : > $out
for spec in "30 150" "42 107" "60 75" "84 53"; do
	set -- $spec
	python3 bench.py --ne $1 --dt $2 --steps 10 --warmup 2 --develop 3 --no-cpu-baseline --sustained-steps 0 2>>gpurun_out/r5_sizes.err | \
		python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('ne$1 L30 dt $2: step %.3f ms  %.2f G DOF-updates/s | ' % (d['ms_per_step'], d['value']/1e9) + ' '.join('%s %.3f' % (a, b) for a, b in sorted(k.items()) if b > 0))" >> $out
done
cat $out
