#!/bin/bash
# Where the time of the column solve goes (k_vi_pair at ne30 L30, one GPU): ablation builds of the library (never shipped,
# results of the ablated builds are wrong by construction) timed with bench.py's HIP-event profile.
# Build here (tools/column_solve_ablation.sh build), run on the GPU box (tools/column_solve_ablation.sh run > profiles/rNN_column_solve_ablation.txt)
cd "$(dirname "$0")/.."
VARIANTS="full: noback:-DTMX_ABL_NO_BACK noback_noustore:-DTMX_ABL_NO_BACK,-DTMX_ABL_NO_USTORE noback_fastproducer:-DTMX_ABL_NO_BACK,-DTMX_ABL_FAST_PRODUCER noback_noelim:-DTMX_ABL_NO_BACK,-DTMX_ABL_NO_ELIM noback_noustore_fastproducer:-DTMX_ABL_NO_BACK,-DTMX_ABL_NO_USTORE,-DTMX_ABL_FAST_PRODUCER"
if [ "$1" = build ]; then
	for v in $VARIANTS; do
		name=${v%%:*}; flags=$(echo "${v#*:}" | tr ',' ' ')
		tools/ablate.sh $name "$flags"
	done
	exit 0
fi
echo "column solve ablation, ne30 np4 L30, ARS343 (3 launches of k_vi_pair<true,true,2> per step); ms of the column solve per step (HIP events)"
echo "variants: noback = no back substitution; noustore = the elimination does not store its U rows; fastproducer = the block row is"
echo "evaluated once and re-used (assembly cost removed, ring traffic kept); noelim = the consumer wavefront only keeps the barrier rhythm"
for v in $VARIANTS; do
	name=${v%%:*}
	TMX_LIBRARY=$PWD/build/abl/$name.so python3 bench.py --steps 10 --warmup 2 --develop 3 --no-cpu-baseline --sustained-steps 0 2>/dev/null | \
		python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-32s vi_solve %.3f ms/step = %.1f us/launch   (step %.3f ms)' % ('$name', d['kernel_ms_per_step']['vi_solve'], d['kernel_ms_per_step']['vi_solve']/3*1e3, d['ms_per_step']))"
done
