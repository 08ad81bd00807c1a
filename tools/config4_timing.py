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
# Byte table of this shape (the rule of SURVEY 8(d): every kernel reads / writes each array it needs once; unit = one (stored node, level),
# element-major layout -- the configuration has no node-unique form; doubles per point):
#   per ARS343 stage (4): H explicit 25 (8(d) row: 5 R state + 5 R + 5 W update + 10 metric) | stage combination 18.75 (75 per step, 8(d) row)
#     | tracer fluxes 12 (3 x [R q, R base, W] + rho, U, V of the initial instance) | uniform diffusion 20 (5 R state, 5 R reference, 5 RMW update)
#     | V explicit 10 (8(d) row) | explicitly evaluated column terms 11 (5 R + 3 RMW) | explicit tracer column update 11 (3 RMW + 3 R + W, rho)
#     | DSS of state and tracers 12 (0.75 of the nodes x 8 variables x RMW)                                              = 119.75 per stage
#   per step: 4 stages 479 | viscosity-free StepAfterSubCycle 0 (nu = 0: the copy is shared) | Kessler 13 (R rho*theta, rho, 3 tracers, z; RMW rho*theta + 3 tracers)
ALG_DOUBLES_PER_POINT = {"h_explicit": 4 * 25.0, "stage_combinations": 75.0, "tracer_fluxes": 4 * 12.0, "uniform_diffusion": 4 * 20.0, "v_explicit": 4 * 10.0,
                         "column_terms_explicit": 4 * 11.0, "tracer_column_update": 4 * 11.0, "dss_state_and_tracers": 4 * 12.0, "kessler": 13.0}
points = 6 * (ne * 4) ** 2 * L
alg_bytes = 8.0 * sum(ALG_DOUBLES_PER_POINT.values()) * points
traffic = None
try:      # the newest committed PMC table of this shape (tools/profile_round.sh, passes fetch_c4 / write_c4)
    import glob, json as _json
    c = sorted(glob.glob(os.path.join(R, "profiles", "r[0-9][0-9]_pmc_traffic_ne%d_L%d_n1.json" % (ne, L))))
    tab = _json.load(open(c[-1]))
    traffic = sum(r["hbm_bytes_per_launch"] * r["launches"] for r in tab["kernels"].values()) / tab["steps"]
    traffic_src = os.path.relpath(c[-1], R)
except Exception:
    traffic_src = None
import json
print(json.dumps({"workload": "BASELINE config 4's shape: supercell set-up ne%d np4 L%d, 3 tracers, uniform diffusion, explicit vertical mode, Kessler every step, ARS343 dt %g s" % (ne, L, dt),
                  "ms_per_step": 1e3 * wall, "value": dof / wall, "unit": "(state + tracer) element-DOF-updates/s", "dtype": "f64",
                  "roofline": {"bound": "hbm", "peak": 8000.0, "unit": "GB/s",
                               "algorithmic_doubles_per_point_per_step": ALG_DOUBLES_PER_POINT, "algorithmic_bytes_per_step": alg_bytes,
                               "achieved": alg_bytes / wall / 1e9, "frac": alg_bytes / wall / 1e9 / 8000.0,
                               "traffic_bytes_per_step": traffic, "traffic_source": traffic_src,
                               "hbm_rate_GBs": (traffic / wall / 1e9) if traffic else None, "hbm_rate_frac": (traffic / wall / 1e9 / 8000.0) if traffic else None},
                  "kernel_ms_per_step": {k: v[0] / 10 for k, v in prof.items() if v[1]}}))
e.close()
