#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats and the separate PMC passes of the bench command, for the workload
# shapes bench.py reports traffic for.  Never combines --pmc with a trace domain other than the kernel trace.
# usage: tools/profile_round.sh <tag> [quick]        outputs under gpurun_out/<tag>_*
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 5 --warmup 2 --develop 3 --no-cpu-baseline --sustained-steps 0"
# PASSES="trace fetch write ..." restricts the run to the named passes (default: all).  Every pass runs under `timeout`: a counter
# request the hardware cannot serve makes rocprofv3 abort and then hang on its incomplete dispatch.
one() {   # name, rocprof args..., -- program
	local name=$1; shift
	if [ -n "${PASSES:-}" ] && ! echo " $PASSES " | grep -q " $name "; then return; fi
	timeout ${PASS_TIMEOUT:-600} rocprofv3 "$@" > $OUT/${TAG}_${name}.log 2>&1 || echo "$name failed"
	echo $SHA > $OUT/${TAG}_${name}.sha      # hash of the kernel sources this pass ran (bench.csrc_sha16; tools/summarise_profiles.py stamps the tables with it)
}
SHA=$(cd $R && python3 -c "import bench; print(bench.csrc_sha16())" 2>/dev/null)
one trace --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o ${TAG} -- python3 $B
one fetch --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o ${TAG} -- python3 $B
one write --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o ${TAG} -- python3 $B
one valu --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_valu -o ${TAG} -- python3 $B
one mfma0 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_mfma0 -o ${TAG} -- python3 $B
# the element-major layout (TMX_UNIQUE=0) beside the default node-unique one
export TMX_UNIQUE=0
one trace_d --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_d -o ${TAG} -- python3 $B
one fetch_d --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch_d -o ${TAG} -- python3 $B
one write_d --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write_d -o ${TAG} -- python3 $B
unset TMX_UNIQUE
one l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/${TAG}_l2 -o ${TAG} -- python3 $B
one lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/${TAG}_lds -o ${TAG} -- python3 $B
export TMX_MFMA=1
one mfma1 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_mfma1 -o ${TAG} -- python3 $B
one trace_mfma1 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_mfma1 -o ${TAG} -- python3 $B
unset TMX_MFMA
# configs[2] shape (ne15) and the 8-rank share of configs[1] (a lone rank-0 engine of the 8-rank layout, wire left out)
B15="$R/bench.py --ne 15 --ppd 1 --steps 5 --warmup 2 --develop 3 --no-cpu-baseline --sustained-steps 0"
one fetch_ne15 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch_ne15 -o ${TAG} -- python3 $B15
one write_ne15 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write_ne15 -o ${TAG} -- python3 $B15
one trace_ne15 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_ne15 -o ${TAG} -- python3 $B15
S8="$R/tools/rank_share_timing.py 8"
one fetch_n8 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch_n8 -o ${TAG} -- python3 $S8
one write_n8 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write_n8 -o ${TAG} -- python3 $S8
one trace_n8 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_n8 -o ${TAG} -- python3 $S8
# BASELINE config 4's shape (ne30 L40, 3 tracers, uniform diffusion, explicit vertical mode, Kessler) and config 5's (ne120 L60, Held-Suarez)
C4="$R/tools/config4_timing.py"
one fetch_c4 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch_c4 -o ${TAG} -- python3 $C4
one write_c4 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write_c4 -o ${TAG} -- python3 $C4
one trace_c4 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_c4 -o ${TAG} -- python3 $C4
B5="$R/bench.py --ne 120 --levels 60 --dt 37.5 --physics hs --steps 3 --warmup 1 --develop 2 --no-cpu-baseline --sustained-steps 0"
one fetch_c5 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch_c5 -o ${TAG} -- python3 $B5
one write_c5 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write_c5 -o ${TAG} -- python3 $B5
one trace_c5 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_c5 -o ${TAG} -- python3 $B5
# the DSS pulled into the second hyperviscosity pass (archived experiment): bytes of k_hypervis / k_dss with and without
export TMX_HVIS_PULL=1 TMX_FLAVOUR=experiments
one fetch_pull --pmc FETCH_SIZE --kernel-include-regex "k_hypervis|k_dss" --output-format csv -d $OUT/${TAG}_fetch_pull -o ${TAG} -- python3 $B
one write_pull --pmc WRITE_SIZE --kernel-include-regex "k_hypervis|k_dss" --output-format csv -d $OUT/${TAG}_write_pull -o ${TAG} -- python3 $B
unset TMX_HVIS_PULL TMX_FLAVOUR
find $OUT -name "*.csv" | head -80
