#!/usr/bin/env python3
"""BASELINE config 4's shape on one GPU: DCMIP2016 supercell set-up at ne30 np4 L40 (planet / 120, no rotation, uniform diffusion,
fully explicit vertical dynamics, three tracers, Kessler microphysics after every step), ARS343, dt 0.5 s.  The stock initial
state needs the reference's Fortran initialiser (> 25 min of host time at this size), so the run starts from the synthesiser's
resting stratified column with a warm moist bubble-free atmosphere and prescribed vapour / cloud / rain profiles: the dynamics
cost is the real one, the microphysics takes its saturated and rain branches; parity of this configuration is pinned at ne2-ne4
(tests).  Run on the GPU box."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import golden_util as gu
from tempestmodel_amd.engine import Engine

ne, L, dt = (int(sys.argv[1]) if len(sys.argv) > 1 else 30), 40, 0.5
g, states = gu.make_grid(ne, L, 24, ztop=20000.0, case="supercell")
tracers = []
for P, (n, e_) in zip(g.patches, states):
    z = P.geom["z_levels"]; rho = n[4]
    qv = 0.014 * np.exp(-z / 2500.0); qc = 2.0e-4 * np.exp(-((z - 3000.0) / 1500.0) ** 2); qr = 1.0e-4 * np.exp(-((z - 2000.0) / 1500.0) ** 2)
    tracers.append(np.stack([rho * qv, rho * qc, rho * qr]))
    # reference state of the uniform diffusion = the initial state, as SupercellTest sets it
    P.geom["ref_node"] = n.copy(); P.geom["ref_redge"] = e_.copy(); P.geom["ref_tracers"] = tracers[-1].copy()
e = Engine(g, fully_explicit=True, uniform_diffusion=(1500.0, 500.0), nu=(0.0, 0.0, 0.0))
e.set_level_heights()
e.upload_state(0, states); e.upload_tracers(0, tracers)
for _ in range(5):
    e.step_ars343(dt); e.kessler(0, dt)
e.sync()
t0 = time.perf_counter()
K = 20
for _ in range(K):
    e.step_ars343(dt); e.kessler(0, dt)
e.sync()
wall = (time.perf_counter() - t0) / K
e.profile_enable(True); e.profile_reset()
for _ in range(10):
    e.step_ars343(dt); e.kessler(0, dt)
e.sync()
prof = e.profile()
st = e.download_state(0)
assert all(np.isfinite(n).all() for n, _ in st)
dof = 6 * ne * ne * 16 * L * (5 + 3)
print("supercell shape ne%d L%d, 3 tracers, Kessler every step: %.3f ms per step, %.2f G (state + tracer) DOF-updates/s, HBM %.2f GiB"
      % (ne, L, 1e3 * wall, dof / wall / 1e9, e.info(5) / 2.0 ** 30), {k: round(v[0] / 10, 3) for k, v in prof.items() if v[1]})
e.close()
