#!/bin/bash
# round 5, column solve: correctness of the decoupled ring, then A/B of library variants on the bench workload (GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_percall.py tests/test_gpu_steppers.py tests/test_gpu_column_kernels.py tests/test_gpu_live_reference.py -x -q -m gpu -k "fused_column or lane_divergent or hundred_steps_ne15 or bench_workload_vs_live or smoke or ars343_steps" > gpurun_out/r5_col_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_col_tests.log
tail -5 gpurun_out/r5_col_tests.log
STEPS=10 timeout 1500 tools/ab.sh r5_col "$@"
