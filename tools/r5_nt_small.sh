#!/bin/bash
# round 5: the non-temporal hint of the column solve's U-row stream on small grids: the production library against a build without it
# (tools/ablate.sh nt0 "-DTMX_USTREAM_NT=0" first; GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r5_nt_small.txt; : > $out
for rep in 1 2; do
for lib in main nt0; do
	L=$PWD/tempestmodel_amd/libtempest_mi355x.so; [ $lib = nt0 ] && L=$PWD/build/abl/nt0.so
	TMX_LIBRARY=$L python3 bench.py --ne 15 --ppd 1 --steps 20 --warmup 3 --develop 3 --no-cpu-baseline --sustained-steps 0 2>/dev/null | \
		python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('ne15 $lib step %.3f ms | ' % d['ms_per_step'] + ' '.join('%s %.3f' % (a, b) for a, b in sorted(k.items()) if b > 0))" >> $out
	TMX_LIBRARY=$L python3 bench.py --ne 20 --steps 20 --warmup 3 --develop 3 --no-cpu-baseline --sustained-steps 0 2>/dev/null | \
		python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('ne20 $lib step %.3f ms | ' % d['ms_per_step'] + ' '.join('%s %.3f' % (a, b) for a, b in sorted(k.items()) if b > 0))" >> $out
done
done
cat $out
