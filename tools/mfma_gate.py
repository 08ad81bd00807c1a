#!/usr/bin/env python3
"""North-star gate for the opt-in MFMA contractions (TMX_MFMA=1): BASELINE config 2 (ne15 np4 L30, dt 300 s), 100 ARS343 steps
from a developed state, per-variable max-norm difference (relative to the variable's max norm) against the default, bit-exact
path on identical inputs.  The default path equals the reference bit for bit (test_hundred_steps_ne15_L30), so this IS the
difference to the reference.  Prints the differences after 1, 10, 30 and 100 steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, q)
import golden_util as gu
from tempestmodel_amd.engine import Engine
ne, L, dt = 15, 30, 300.0
g, states = gu.make_grid(ne, L, 6)
exact = Engine(g)
os.environ["TMX_MFMA"] = "1"
mfma = Engine(g)
os.environ.pop("TMX_MFMA")
exact.upload_state(0, states)
for _ in range(3):          # developed start (the first step of the balanced state is discontinuous in its inputs)
    exact.step_ars343(dt)
exact.sync()
start = exact.download_state(0)
mfma.upload_state(0, start)
for n in range(1, 101):
    exact.step_ars343(dt); mfma.step_ars343(dt)
    if n in (1, 10, 30, 100):
        exact.sync(); mfma.sync()
        errs = gu.prognostic_errors(mfma.download_state(0), exact.download_state(0))
        print("MFMA vs exact path after %3d steps: U %.2e V %.2e rho*theta %.2e W %.2e rho %.2e" % ((n,) + tuple(errs)), flush=True)
exact.close(); mfma.close()
