#!/bin/bash
# the headline profile passes only (kernel trace, FETCH_SIZE, WRITE_SIZE of ne30 L30 and ne15) with the sources as they are (GPU box)
cd "$(dirname "$0")/.."
PASSES="trace fetch write trace_ne15 fetch_ne15 write_ne15" tools/profile_round.sh r05 > gpurun_out/r05_profile_final.log 2>&1
cat gpurun_out/r05_trace.sha
