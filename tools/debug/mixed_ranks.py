#!/usr/bin/env python3
"""Debug aid: n rank engines (loopback wire) where only one rank's patches carry differing seam copies, against one element-major engine."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import golden_util as gu
from parity_common import _rank_engines_step
from tempestmodel_amd.engine import Engine

def perturb(arrs, rng):
    out = []
    for a in arrs:
        a = np.array(a, copy=True); k = rng.integers(-1, 2, size=a.shape)
        out.append(np.where(k > 0, np.nextafter(a, np.inf), np.where(k < 0, np.nextafter(a, -np.inf), a)))
    return tuple(out)

n_ranks = 3
g, start = gu.make_grid(12, 6, 24)
for scheme, mixed, who, nsteps in [("strang", 1, None, 1), ("strang", 1, 1, 1), ("strang", 1, 1, 2), ("ars343", 0, 1, 1), ("ars343", 1, 1, 1), ("ars232", 1, 1, 1), ("strang", 1, 0, 1)]:
    rng = np.random.default_rng(3)
    ni = max(7, Engine.scheme_instances(scheme))
    single = Engine(g, n_instances=ni, options={"unique_layout": 0})
    ranks = [Engine(g, rank=r, n_ranks=n_ranks, n_instances=ni, options={"unique_layout": 1, "unique_mixed": mixed}) for r in range(n_ranks)]
    single.upload_state(0, start)
    for k in range(2):
        single.step(scheme, 200.0, first=(k == 0))
    single.sync()
    cur = single.download_state(0)
    odd = {p: (perturb(cur[p], rng) if (who is not None and p in ranks[who].local_patches) else cur[p]) for p in range(24)}
    single.upload_state(0, odd)
    for k in range(nsteps):
        single.step(scheme, 200.0)
    single.sync()
    ref = single.download_state(0)
    for e in ranks:
        e.upload_state(0, odd)
    Engine.loopback_group(ranks)
    _rank_engines_step(ranks, lambda e, k: e.step(scheme, 200.0), nsteps)
    Engine.loopback_dissolve(ranks[0])
    line = "%s mixed=%d perturbed rank %s steps %d:" % (scheme, mixed, who, nsteps)
    for e in ranks:
        got = e.download_state(0)
        worst = 0.0; bad = []
        for p in e.local_patches:
            d = max(float(np.max(np.abs(got[p][0][[0, 1, 2, 4], 1:-1, 1:-1] - ref[p][0][[0, 1, 2, 4], 1:-1, 1:-1]))), float(np.max(np.abs(got[p][1][3, 1:-1, 1:-1] - ref[p][1][3, 1:-1, 1:-1]))))
            if d > 0: bad.append(p)
            worst = max(worst, d)
        line += "  rank %d [U inst %d, mixed steps %d] max|diff| %.2e patches %s" % (e.rank, e.info(13), e.info(19), worst, bad)
    print(line, flush=True)
    single.close()
    for e in ranks:
        e.close()
