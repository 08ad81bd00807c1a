#!/usr/bin/env python3
"""In-kernel cycle accounting of the column kernel on the bench workload (ne30 L30, developed state): run with a library built
with -DTMX_PAIR_TIMING (tools/ablate.sh pairtiming "-DTMX_PAIR_TIMING"; TMX_LIBRARY=build/abl/pairtiming.so); one column
solve is launched and its per-wavefront cycle records are printed when the statistics are switched off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tempestmodel_amd.cubed_sphere import CubedSphereGrid, BaroclinicWaveJW
from tempestmodel_amd.engine import Engine
grid = CubedSphereGrid(30, 30, 30000.0, ppd=2)
states = grid.evaluate_test_case(BaroclinicWaveJW(0.0, 30000.0, "exp"))
e = Engine(grid)
e.upload_state(0, states)
os.environ["TMX_QUIET"] = "1"
sys.stdout.flush()
# develop the state with stdout parked (the instrumented kernel prints on every launch)
devnull = os.open(os.devnull, os.O_WRONLY); keep = os.dup(1); os.dup2(devnull, 1)
for _ in range(3):
    e.step_ars343(150.0)
e.sync()
if "--step" in sys.argv:      # the column solves of a whole step (on the node-unique layout where that is the default): records of the last one
    os.dup2(keep, 1)
    print("---- the last column solve of one ARS343 step, ne30 L30, developed state, %d instances in node-unique form ----" % e.info(13), flush=True)
    e.pivot_stats(True)
    e.step_ars343(150.0)
    e.sync()
    e.pivot_stats(False)
    e.close()
    sys.exit(0)
e.copy_data(0, 1)
for _ in range(3):
    e.v_step_implicit(1, 1, 65.0)      # warm
e.sync()
os.dup2(keep, 1)
print("---- one column solve, ne30 L30, developed state ----", flush=True)
e.pivot_stats(True)
e.v_step_implicit(1, 1, 65.0)
e.sync()
e.pivot_stats(False)      # the diagnostic build prints its per-wavefront records to stderr here
e.close()
