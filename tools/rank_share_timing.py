#!/usr/bin/env python3
"""Per-rank compute share of an ARS343 step at N ranks, measured on ONE GPU: a lone rank-0 engine of the N-rank
24-patch ne30 layout runs whole steps with the wire left out (TMX_DEBUG_SKIP_EXCHANGE: packing still runs, nothing is
sent, results at the rank boundary are wrong).  What a multi-GPU step costs besides the exchange itself."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TMX_DEBUG_SKIP_EXCHANGE"] = "1"
os.environ["TMX_FLAVOUR"] = "experiments"      # the timing aid is compiled into that flavour of the library only
from tempestmodel_amd.cubed_sphere import CubedSphereGrid, BaroclinicWaveJW
from tempestmodel_amd.engine import Engine

grid = CubedSphereGrid(30, 30, 30000.0, ppd=2)
states = grid.evaluate_test_case(BaroclinicWaveJW(0.0, 30000.0, "exp"))
shares = {}
kernels = {}
RANKS = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]      # e.g. `rank_share_timing.py 8` under rocprofv3
for n in RANKS:
    e = Engine(grid, rank=0, n_ranks=n)
    e.upload_state(0, states)
    for _ in range(3):
        e.step_ars343(150.0)
    e.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        e.step_ars343(150.0)
    e.sync()
    wall = (time.perf_counter() - t0) / 20
    e.profile_enable(True); e.profile_reset()
    for _ in range(10):
        e.step_ars343(150.0)
    e.sync()
    prof = e.profile()
    print("ranks %d: rank-0 share %d patches, %.3f ms/step without the wire" % (n, len(e.local_patches), 1e3 * wall),
          {k: round(v[0] / 10, 3) for k, v in prof.items() if v[1]})
    shares[n] = 1e3 * wall
    kernels[n] = {k: round(v[0] / 10, 4) for k, v in prof.items() if v[1]}
    e.close()
# Critical-path estimate of an N-rank step: the per-rank compute share measured above + the part of the six exchanges per
# ARS343 step that the boundary-first launch order does not hide.  The wire itself cannot be measured on a one-GPU box; the
# estimate takes T_X microseconds of exposed latency per exchange (RCCL grouped send/recv of 0.1-0.9 MB to <= 4 peers over
# xGMI) for two values of T_X.
if 1 in shares:
    for tx in (10.0, 30.0):
        print("exposed exchange latency %2.0f us x 6 per step:" % tx,
              "  ".join("N=%d %.3f ms -> speed-up %.2f" % (n, shares[n] + (6 * tx * 1e-3 if n > 1 else 0.0),
                                                          shares[1] / (shares[n] + (6 * tx * 1e-3 if n > 1 else 0.0))) for n in sorted(shares)))

# machine-readable copy (bench.py's N > 1 lines quote it as multi_rank.projected_from_rank_share): TMX_RANK_SHARE_JSON=<path>
if os.environ.get("TMX_RANK_SHARE_JSON"):
    import json
    json.dump({"what": "compute share of rank 0 of an N-rank ne30 np4 L30 ARS343 step on ONE MI355X, wire left out (tools/rank_share_timing.py)",
               "ms_per_step": {str(n): shares[n] for n in sorted(shares)}, "kernel_ms_per_step": {str(n): kernels[n] for n in sorted(kernels)}},
              open(os.environ["TMX_RANK_SHARE_JSON"], "w"), indent=1)
