#!/usr/bin/env python3
"""Phase cycles of the lane-group column kernel at the 8-rank share of ne30 L30 (library built with -DTMX_VI_TIMING -DTMX_EXPERIMENTS:
tools/ablate.sh vitiming "-DTMX_VI_TIMING -DTMX_EXPERIMENTS"; TMX_LIBRARY=build/abl/vitiming.so).  Round 5: first assembly round 16 500 cycles, elimination
(+ the second assembly round) 73 000, back substitution 44 500 per wavefront; a hand-pipelined back substitution (the next row's gather and partial
sum formed beside the current row's division) was bit-identical and SLOWER (51 000): not kept."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
os.environ["TMX_DEBUG_SKIP_EXCHANGE"] = "1"; os.environ["TMX_FLAVOUR"] = "experiments"; os.environ["TMX_QUIET"] = "1"
from tempestmodel_amd.cubed_sphere import CubedSphereGrid, BaroclinicWaveJW
from tempestmodel_amd.engine import Engine
grid = CubedSphereGrid(30, 30, 30000.0, ppd=2)
states = grid.evaluate_test_case(BaroclinicWaveJW(0.0, 30000.0, "exp"))
e = Engine(grid, rank=0, n_ranks=8)
e.upload_state(0, states)
devnull = os.open(os.devnull, os.O_WRONLY); keep = os.dup(1); os.dup2(devnull, 1)
for _ in range(3):
    e.step_ars343(150.0)
e.sync()
e.copy_data(0, 1)
for _ in range(2):
    e.v_step_implicit(1, 1, 65.0)
e.sync()
os.dup2(keep, 1)
print("---- one column solve, 8-rank share of ne30 L30 (%d unique columns) ----" % e.info(1), flush=True)
e.v_step_implicit(1, 1, 65.0)
e.sync()
e.close()
