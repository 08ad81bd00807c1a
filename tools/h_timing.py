#!/usr/bin/env python3
"""Where a wavefront of the fused explicit kernel spends its life (shader cycles per phase), bench workload, both layouts.
Needs a diagnostic build: tools/ablate.sh htiming "-DTMX_H_TIMING"; TMX_LIBRARY=build/abl/htiming.so python3 tools/h_timing.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TMX_QUIET"] = "1"
import numpy as np
from tempestmodel_amd.cubed_sphere import CubedSphereGrid, BaroclinicWaveJW
from tempestmodel_amd.engine import Engine
grid = CubedSphereGrid(30, 30, 30000.0, ppd=2)
states = grid.evaluate_test_case(BaroclinicWaveJW(0.0, 30000.0, "exp"))
names = ["loads issued", "data there + pointwise", "Exner + LDS writes", "contractions", "updates", "barrier + W", "stores"]
for layout in (0, 1):
    e = Engine(grid, options={"unique_layout": layout})
    e.upload_state(0, states)
    for _ in range(4):
        e.step_ars343(150.0)
    e.sync()
    buf = (ctypes.c_ulonglong * 128)()
    e.lib.tmx_debug_h_timing(e.h, buf)      # clear
    e.step_ars343(150.0)      # one step: one launch per instantiation, one record per wavefront
    e.sync()
    e.lib.tmx_debug_h_timing(e.h, buf)
    a = np.array(buf[:], dtype=np.float64).reshape(16, 8)
    print("layout:", "node-unique" if layout else "element-major")
    for nt in range(16):
        if a[nt, 7] == 0:
            continue
        per = a[nt, :7] / a[nt, 7]
        print("  terms %2d: %7d wavefronts, cycles per wavefront %7.0f = " % (nt, a[nt, 7], per.sum()) + ", ".join("%s %.0f" % (n, v) for n, v in zip(names, per)))
    e.close()
