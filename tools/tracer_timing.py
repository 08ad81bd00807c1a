#!/usr/bin/env python3
"""ARS343 step of the JW wave with tracers (implicit vertical dynamics: the column update of every tracer is a band solve per
column), ne30 np4 L30 by default, one GPU.  usage: tracer_timing.py [ne] [levels] [ntracers]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import golden_util as gu
from tempestmodel_amd.engine import Engine
ne = int(sys.argv[1]) if len(sys.argv) > 1 else 30
L = int(sys.argv[2]) if len(sys.argv) > 2 else 30
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g, states = gu.make_grid(ne, L, 24, ntracers=nt)
tr = [g.initial_tracers[P.index] for P in g.patches] if nt else None
e = Engine(g)
e.upload_state(0, states)
if tr is not None:
    e.upload_tracers(0, tr)
dt = 150.0 * 30 / ne
for _ in range(5):
    e.step_ars343(dt)
e.sync()
t0 = time.perf_counter()
for _ in range(10):
    e.step_ars343(dt)
e.sync()
wall = (time.perf_counter() - t0) / 10
e.profile_enable(True); e.profile_reset()
for _ in range(5):
    e.step_ars343(dt)
e.sync()
prof = e.profile()
print("JW ne%d L%d with %d tracers: %.3f ms per step" % (ne, L, nt, 1e3 * wall), {k: round(v[0] / 5, 3) for k, v in prof.items() if v[1]})
e.close()
