#!/usr/bin/env python3
"""Where a wavefront of the column-segment walk (explicit stage, node-unique layout) spends its life: shader cycles per phase, bench workload.
Needs a diagnostic build: tools/ablate.sh htiming "-DTMX_H_TIMING"; TMX_LIBRARY=build/abl/htiming.so python3 tools/h_walk_timing.py [ne] [L] [segments ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TMX_QUIET"] = "1"
import numpy as np
from tempestmodel_amd.cubed_sphere import CubedSphereGrid, BaroclinicWaveJW
from tempestmodel_amd.engine import Engine
ne = int(sys.argv[1]) if len(sys.argv) > 1 else 30
L = int(sys.argv[2]) if len(sys.argv) > 2 else 30
segs = [int(a) for a in sys.argv[3:]] or [-1000]
grid = CubedSphereGrid(ne, L, 30000.0, ppd=2)
states = grid.evaluate_test_case(BaroclinicWaveJW(0.0, 30000.0, "exp"))
names = ["prologue", "loads issued", "pointwise + Exner + LDS writes", "contractions", "updates", "wait for terms + reduce", "seam exchange + stores"]
for sg in segs:
    e = Engine(grid, options={"h_walk": sg})
    e.upload_state(0, states)
    for _ in range(4):
        e.step_ars343(150.0)
    e.sync()
    buf = (ctypes.c_ulonglong * 128)()
    e.lib.tmx_debug_h_walk_timing(e.h, buf)      # clear
    e.step_ars343(150.0)      # one step: one launch per instantiation, one record per wavefront
    e.sync()
    e.lib.tmx_debug_h_walk_timing(e.h, buf)
    a = np.array(buf[:], dtype=np.float64).reshape(16, 8)
    print("ne%d L%d h_walk=%d" % (ne, L, sg))
    for nt in range(8):
        if a[nt, 7] == 0:
            continue
        per = a[nt, :7] / a[nt, 7]
        print("  terms %2d: %7d wavefronts, cycles per wavefront %7.0f = " % (nt, a[nt, 7], per.sum()) + ", ".join("%s %.0f" % (n, v) for n, v in zip(names, per)))
    e.close()
