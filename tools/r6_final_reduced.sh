#!/bin/bash
# end of round 6, last change (per-CU role order in k_vi_pair): everything of tools/r6_final.sh except the config-5 (ne120 L60) passes and runs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|ERROR|skipped" | tail -6 > gpurun_out/r06_gpu_tests.log
cat gpurun_out/r06_gpu_tests.log
PASSES="trace fetch write valu l2 lds trace_d fetch_d write_d trace_ne15 fetch_ne15 write_ne15 trace_n8 fetch_n8 write_n8 trace_c4 fetch_c4 write_c4" tools/profile_round.sh r06 > gpurun_out/r06_profile_final.log 2>&1
TMX_RANK_SHARE_JSON=$PWD/gpurun_out/r06_rank_shares.json python3 tools/rank_share_timing.py 1 2 4 8 > gpurun_out/r06_rank_shares.txt 2>&1
tail -3 gpurun_out/r06_rank_shares.txt
{ python3 tools/config4_timing.py 2>/dev/null | tail -3
  for args in "--ne 15 --ppd 1" "--ne 60 --levels 30 --dt 75 --steps 10" "--ne 84 --levels 30 --dt 50 --steps 5"; do
	python3 bench.py $args --no-cpu-baseline --sustained-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('$args: step %.3f ms = %.3g DOF-updates/s | ' % (d['ms_per_step'], d['value']) + ' '.join('%s %.3f' % (a, b) for a, b in sorted(k.items()) if b > 0))"
  done; } > gpurun_out/r06_other_shapes_reduced.txt 2>&1
cat gpurun_out/r06_other_shapes_reduced.txt | cut -c1-200
python3 bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
tail -c 300 gpurun_out/r06_bench.json
