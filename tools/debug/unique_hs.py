#!/usr/bin/env python3
"""Do the copies of a node still agree after the Held-Suarez forcing?  (why a run with column physics leaves the node-unique layout)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_util as gu
from tempestmodel_amd.engine import Engine

def seams(st, npatch):
    bad = np.zeros(5, dtype=np.int64)
    for p in range(npatch):
        x, r = st[p]
        na = x.shape[1]
        for a in range(1, (na - 2) // 4):
            for v in range(x.shape[0]):
                bad[v] += (x[v, a * 4, 1:-1, :].view(np.uint64) != x[v, a * 4 + 1, 1:-1, :].view(np.uint64)).sum()
                bad[v] += (x[v, 1:-1, a * 4, :].view(np.uint64) != x[v, 1:-1, a * 4 + 1, :].view(np.uint64)).sum()
    return bad

g, start = gu.make_grid(4, 8, 6)
e = Engine(g, options={"unique_layout": 1})
e.set_physics_inputs(None)
e.upload_state(0, start)
print("initial:", seams(start, 6))
for k in range(4):
    e.step_ars343(100.0); e.sync()
    print("after step", k, "unique instances", e.info(13), "copies differing per node variable:", seams(e.download_state(0), 6))
    e.held_suarez(0, 100.0); e.sync()
    st = e.download_state(0)
    rb = sum(int((st[p][1][v, a * 4, 1:-1, :].view(np.uint64) != st[p][1][v, a * 4 + 1, 1:-1, :].view(np.uint64)).sum()) for p in range(6) for v in range(st[p][1].shape[0]) for a in range(1, 4))
    print("after forcing", k, "copies differing:", seams(st, 6), "interface arrays:", rb, "conversions", e.info(14))
