#!/bin/bash
# per-role cycle accounting of k_vi_pair (GPU box): tools/r5_pairtiming.sh <tag> [lib under build/abl, default pairtiming]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=${1:-r5}; lib=${2:-pairtiming}
TMX_LIBRARY=$PWD/build/abl/$lib.so timeout 600 python3 tools/vi_timing.py --step > gpurun_out/pt_$tag.out 2> gpurun_out/pt_$tag.err
grep "segments" gpurun_out/pt_$tag.out | head -4
python3 tools/vi_timing_summary.py < gpurun_out/pt_$tag.err | tee gpurun_out/pt_${tag}_summary.txt
