#!/bin/bash
# end of round 5 (GPU box): the GPU test suite, the headline profile passes with the final sources, the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1
tail -3 gpurun_out/r05_gpu_tests.log
PASSES="trace fetch write trace_ne15 fetch_ne15 write_ne15" tools/profile_round.sh r05 > gpurun_out/r05_profile_final.log 2>&1
python3 bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
tail -c 600 gpurun_out/r05_bench.json
