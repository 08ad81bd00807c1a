#!/usr/bin/env python3
"""Every scheme on a 24-patch grid (3 x 3 element patches), node-unique layout against element-major, step by step: where does the first
difference appear?  (Found in round 6: ARK232's StepImplicitTermsExplicitly on grids whose metric copies differ in the last bit.)  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["TMX_QUIET"] = "1"
import numpy as np
import golden_util as gu
from tempestmodel_amd.engine import Engine
ne, npatch, shape = 6, 24, 0
g, start = gu.make_grid(ne, 6, npatch)
for scheme in ("ars343", "ars232", "ars222", "ars443", "ark232", "strang", "strang_fe", "strang_rk4", "strang_ssp3", "strang_ssprk53"):
    for blocks, flav, extra in ((0, None, {}), (1, "experiments", {})):
        ni = max(7, Engine.scheme_instances(scheme))
        d = Engine(g, n_instances=ni, options={"unique_layout": 0})
        opts = {"unique_layout": 1, "unique_tile_shape": shape}
        if flav: opts["unique_blocks"] = blocks
        opts.update(extra)
        u = Engine(g, n_instances=ni, options=opts, flavour=flav)
        d.upload_state(0, start); u.upload_state(0, start)
        for k in range(4):
            d.step(scheme, 150.0, first=(k == 0)); u.step(scheme, 150.0, first=(k == 0))
            d.sync(); u.sync()
            a, b = d.download_state(0), u.download_state(0)
            bad = [(p_, v) for p_ in range(npatch) for v in range(5) if not np.array_equal(a[p_][0][v] if v != 3 else a[p_][1][v], b[p_][0][v] if v != 3 else b[p_][1][v])]
            print(scheme, "blocks", blocks, flav, extra, "step", k, "unique instances", u.info(13), "mismatching (patch, var):", bad[:6], len(bad))
            if bad:
                p_, v = bad[0]
                x = a[p_][0][v] if v != 3 else a[p_][1][v]; y = b[p_][0][v] if v != 3 else b[p_][1][v]
                idx = np.argwhere(x != y)
                print("   first differing entries (i, j, k):", idx[:8].tolist(), "of", len(idx), "values", x[tuple(idx[0])], y[tuple(idx[0])])
                break
        d.close(); u.close()
