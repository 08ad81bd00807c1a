"""Device vs oracle on the 24-patch Held-Suarez case, op by op (GPU box; needs oracle/_ref)."""
import os, sys, subprocess
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, tmxd, golden_util as gu
from oracle_lib import Oracle
from tempestmodel_amd.engine import Engine
ne, L, npatch, dt = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), (int(sys.argv[2]) if len(sys.argv) > 2 else 20), 24, 200.0
out = "/tmp/hs8.tmxd"
env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
subprocess.run([os.path.join(R, "oracle/_ref/tmx_ref_dump_lapack2"), "--case", "heldsuarez", "--physics", "hs", "--mode", "steps", "--ne", str(ne),
                "--levels", str(L), "--dt", str(dt), "--patches", str(npatch), "--warm", "2", "--steps", "1", "--geom", "1", "--out", out],
               env=env, check=True, stdout=subprocess.DEVNULL)
rd = tmxd.read(out)
g, _ = gu.grid_from_fixture(rd, override=True, case="heldsuarez")
for P in g.patches:
    P.lat = rd["p%d/hs_latitude" % P.index]
start = [(rd["state/warm/p%d/node" % p], rd["state/warm/p%d/redge" % p]) for p in range(npatch)]
o = Oracle(g); o.set_state(0, start)
e = Engine(g); e.set_physics_inputs(None); e.upload_state(0, start)

def cmp(tag, inst=0):
    e.sync()
    got, want = e.download_state(inst), o.get_state(inst)
    errs = gu.prognostic_errors(got, want)
    ss = max(float(np.max(np.abs(a[1][[2, 4], 1:-1, 1:-1, 0] - b[1][[2, 4], 1:-1, 1:-1, 0]))) for a, b in zip(got, want))
    worst = None
    if max(errs) > 0:
        for P, (a, _), (b, _) in zip(g.patches, got, want):
            d = np.abs(a[0, 1:-1, 1:-1] - b[0, 1:-1, 1:-1])
            if d.max() > 0:
                i, j, k = np.unravel_index(np.argmax(d), d.shape)
                worst = (P.index, int(i) + 1, int(j) + 1, int(k), float(d.max()), a.shape)
                break
    print("%-28s" % tag, ["%.1e" % v for v in errs], "surface slots %.1e" % ss, worst)

cmp("upload")
# one ARS343 step as the reference's call sequence, stage by stage (tmx_step fuses; here: whole step, then physics)
o.step_ars343(dt); e.step_ars343(dt); cmp("after dynamics step")
o.held_suarez(0, dt); e.held_suarez(0, dt); cmp("after forcing")
o.step_ars343(dt); e.step_ars343(dt); cmp("after dynamics step 2")
o.held_suarez(0, dt); e.held_suarez(0, dt); cmp("after forcing 2")
e.close()
