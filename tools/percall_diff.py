"""Per-call device-vs-oracle differences on the per-call fixture (identical inputs): which call is not bit-exact.
Run on the GPU box:  python tools/percall_diff.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import golden_util as gu
from oracle_lib import Oracle
from tempestmodel_amd.engine import Engine

d = gu.load("percall_ne3_L5_p6.npz")
g, _ = gu.grid_from_fixture(d, override=True)
npatch = 6
dt = float(d["cfg/dt"][0]); sdt = dt * gu.ARS343_GAMMA
o = Oracle(g); e = Engine(g)
start = gu.full_states(d, "warm", npatch)


def cmp(tag, ie, io, ref=None):
    e.sync()
    got = e.download_state(ie)
    print("%-12s dev-oracle %s" % (tag, ["%.1e" % v for v in gu.prognostic_errors(got, o.get_state(io))]),
          ("dev-ref %s" % ["%.1e" % v for v in gu.prognostic_errors(got, gu.full_states(d, ref, npatch))]) if ref else "")


o.set_state(0, start); e.upload_state(0, start)
o.copy_data(0, 1); e.copy_data(0, 1)
o.h_step_explicit(0, 1, sdt); e.h_step_explicit(0, 1, sdt); cmp("h_explicit", 1, 1, "h_explicit")
# restart each call from the oracle's state so that differences do not accumulate
e.upload_state(1, o.get_state(1))
o.v_step_explicit(0, 1, sdt); e.v_step_explicit(0, 1, sdt); cmp("v_explicit", 1, 1, "v_explicit")
e.upload_state(1, o.get_state(1))
o.apply_dss(1); e.apply_dss(1); cmp("dss", 1, 1, "dss")
e.upload_state(1, o.get_state(1))
o.copy_data(1, 2); e.copy_data(1, 2)
o.v_step_implicit(2, 2, sdt); e.v_step_implicit(2, 2, sdt); cmp("v_implicit", 2, 2, "v_implicit")
e.upload_state(2, o.get_state(2))
o.h_step_after_subcycle(2, 3, 4, dt); e.h_step_after_subcycle(2, 3, 4, dt); cmp("hypervis", 3, 3, "hypervis"); cmp("hv work", 4, 4)
e.upload_state(3, o.get_state(3))
c = [0.25, -0.5, 1.75, 0.5, 0, 0, 0]
o.linear_combine_data(c, 3); e.linear_combine_data(c, 3); cmp("lincomb", 3, 3, "lincomb")
# whole steps from the warm state
o.set_state(0, start); e.upload_state(0, start)
for s in range(3):
    o.step_ars343(dt); e.step_ars343(dt); cmp("step %d" % (s + 1), 0, 0)
e.close()
