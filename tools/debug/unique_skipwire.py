#!/usr/bin/env python3
"""Why does a lone rank-0 engine of an N-rank layout (wire left out) not stay on the node-unique layout?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TMX_DEBUG_SKIP_EXCHANGE"] = "1"
import numpy as np
from tempestmodel_amd.cubed_sphere import CubedSphereGrid, BaroclinicWaveJW
from tempestmodel_amd.engine import Engine
grid = CubedSphereGrid(12, 6, 30000.0, ppd=2)
states = grid.evaluate_test_case(BaroclinicWaveJW(0.0, 30000.0, "exp"))
e = Engine(grid, rank=0, n_ranks=2, options={"unique_layout": 1})
e.upload_state(0, states)
for k in range(4):
    e.step_ars343(150.0)
    print("step", k, "unique instances", e.info(13), "conversions", e.info(14), "layout built", e.info(12))
e.sync()
st = e.download_state(0)
np4 = 4
for p in e.local_patches:
    node = st[p][0]      # [5][na][nb][L]
    na, nb = node.shape[1], node.shape[2]
    bad = 0
    for a in range(1, (na - 2) // np4):
        i0, i1 = a * np4, a * np4 + 1      # last node of element a-1, first of element a (1-based interior)
        d = node[:, i0, 1:-1, :] != node[:, i1, 1:-1, :]
        bad += int(d.sum())
    print("patch", p, "alpha-seam mismatches", bad, "nan", int(np.isnan(node).sum()))
