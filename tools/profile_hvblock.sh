#!/bin/bash
# GPU box: bytes of the hyperviscosity / DSS kernels with the fused block kernel (TMX_HVIS_BLOCK=1) and without
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 5 --warmup 2 --develop 3 --no-cpu-baseline --sustained-steps 0"
for blk in 1 0; do
	export TMX_HVIS_BLOCK=$blk TMX_FLAVOUR=experiments
	rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_hypervis|k_dss" --output-format csv -d $OUT/hvb${blk}_fetch -o hvb -- python3 $B > /dev/null 2>&1
	rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_hypervis|k_dss" --output-format csv -d $OUT/hvb${blk}_write -o hvb -- python3 $B > /dev/null 2>&1
	rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hvb${blk}_trace -o hvb -- python3 $B > /dev/null 2>&1
	echo "TMX_HVIS_BLOCK=$blk"
	python3 $R/tools/pmc_table.py $OUT/hvb${blk}_fetch/hvb_counter_collection.csv $OUT/hvb${blk}_write/hvb_counter_collection.csv 15 $OUT/hvb${blk}_traffic.json "ne30 np4 L30, 1 GPU, TMX_HVIS_BLOCK=$blk, k_hypervis / k_dss only" | cut -c1-110
	grep -E "k_hypervis|k_dss" $OUT/hvb${blk}_trace/hvb_kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
done
