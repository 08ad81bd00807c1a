"""Reference-vs-reference spread of the 100-step north-star gates (CPU only, build container: needs oracle/_ref).

The reference does not pin a LAPACK (mk/system/*.make: -llapack).  oracle/_ref/tmx_ref_dump resolves dgbsv_/dgbtrf_/
dgbtrs_ to MKL, oracle/_ref/tmx_ref_dump_lapack2 (oracle/refaux/lapack_alt_shim.c) to the netlib LAPACK inside scipy's
OpenBLAS; every other byte of the two binaries is the same.  Both run 2 warm-up + 100 ARS343 steps from the same
initial state; the max-norm relative difference per variable (U V rho*theta W rho) between their end states is the
spread two *valid builds of the reference itself* show, which is the floor of any meaningful tolerance.

Writes tests/golden/lapack_spread.json (end-state checksums of both binaries + the spread + the C oracle's distance
to both, the oracle running on the reference's own geometry and operators: `--geom 1`), which tests/test_oracle_vs_reference.py::test_w_gate_tolerance_follows_reference_spread and
the tests/test_gpu_*.py parity files read to set the W tolerance.

usage: python tests/analysis/lapack_spread.py [--cases ne4,ne15]
"""
import json
import os
import subprocess
import sys
import hashlib
import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import golden_util as gu  # noqa: E402
import tmxd  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

CASES = {"ne4_L10_dt200": (4, 10, 200.0), "ne15_L30_dt300": (15, 30, 300.0)}


def run(binary, ne, L, dt, out, geom="0"):
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    txt = subprocess.run([os.path.join(R, "oracle", "_ref", binary), "--mode", "steps", "--ne", str(ne), "--levels", str(L),
                          "--dt", str(dt), "--warm", "2", "--steps", "100", "--every", "100", "--geom", geom, "--out", out],
                         env=env, check=True, stdout=subprocess.PIPE, universal_newlines=True).stdout
    sums = {}
    for line in txt.splitlines():
        if line.startswith("checksum "):
            f = line.split()
            sums[f[1]] = [float(v) for v in f[2:]]
    rd = tmxd.read(out)
    os.remove(out)
    grid = gu.grid_from_fixture(rd, override=True)[0] if geom == "1" else None      # the reference's own geometry
    return gu.full_states(rd, "warm", 6), gu.full_states(rd, "step100", 6), sums, grid


def digest(states):
    h = hashlib.sha256()
    for node, redge in states:
        h.update(np.ascontiguousarray(node[[0, 1, 2, 4]][:, 1:-1, 1:-1]).tobytes())
        h.update(np.ascontiguousarray(redge[3][1:-1, 1:-1]).tobytes())
    return h.hexdigest()


def main():
    which = sys.argv[sys.argv.index("--cases") + 1].split(",") if "--cases" in sys.argv else ["ne4", "ne15"]
    path = os.path.join(R, "tests", "golden", "lapack_spread.json")
    res = json.load(open(path)) if os.path.exists(path) else {}
    for name, (ne, L, dt) in CASES.items():
        if name.split("_")[0] not in which:
            continue
        w1, e1, s1, g = run("tmx_ref_dump", ne, L, dt, "/tmp/spread_a.tmxd", geom="1")
        w2, e2, s2, _ = run("tmx_ref_dump_lapack2", ne, L, dt, "/tmp/spread_b.tmxd")
        assert digest(w1) != "" and len(w1) == len(w2)
        warm_diff = gu.prognostic_errors(w2, w1)
        spread = gu.prognostic_errors(e2, e1)
        o = Oracle(g); o.set_state(0, w1)
        for _ in range(100):
            assert o.step_ars343(dt) == 0
        oe = o.get_state(0)
        res[name] = {
            "ne": ne, "levels": L, "dt": dt, "warm_steps": 2, "steps": 100, "scheme": "ars343",
            "variables": ["U", "V", "RhoTheta", "W", "Rho"],
            "mkl": {"checksum_step100": s1.get("step100"), "sha256_prognostic": digest(e1)},
            "openblas_netlib": {"checksum_step100": s2.get("step100"), "sha256_prognostic": digest(e2)},
            "spread_after_warmup": [float(v) for v in warm_diff],
            "spread_ref_vs_ref": [float(v) for v in spread],
            "oracle_vs_mkl": [float(v) for v in gu.prognostic_errors(oe, e1)],
            "oracle_vs_openblas_netlib": [float(v) for v in gu.prognostic_errors(oe, e2)],
        }
        print(name, json.dumps(res[name], indent=1))
        with open(path, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
