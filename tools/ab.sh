#!/bin/bash
# A/B of library builds on the bench workload (GPU box): tools/ab.sh <tag> <variant> ...
# variant = lib[:ENV=VAL[,ENV=VAL...]]; lib = main | base | a name under build/abl | a path
# prints per-operation kernel times (HIP events, bench.py's profile pass) per variant; two alternations to see box noise
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p gpurun_out
out=gpurun_out/ab_$tag.txt
: > $out
for rep in 1 2; do
for spec in "$@"; do
	v=${spec%%:*}; envs=""
	[ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
	lib=$v; [ -f "$lib" ] || lib=$PWD/build/abl/$v.so
	[ "$v" = main ] && lib=$PWD/tempestmodel_amd/libtempest_mi355x.so
	env $envs TMX_LIBRARY=$lib python3 bench.py --steps ${STEPS:-10} --warmup 2 --develop ${DEVELOP:-3} --no-cpu-baseline --sustained-steps 0 ${BENCH_ARGS} 2>>gpurun_out/ab_$tag.err | \
		python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('%-44s step %.3f ms | ' % ('$spec', d['ms_per_step']) + ' '.join('%s %.3f' % (a, b) for a, b in sorted(k.items()) if b > 0))" >> $out
done
done
cat $out
