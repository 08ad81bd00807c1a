"""Shared by the tests/test_gpu_*.py parity files: the tolerances and a few helpers (round 5: test_gpu_parity.py split by operator family)."""
import os
import numpy as np
import pytest
import golden_util as gu

# Parity bar of this file:
#  * device vs C oracle on IDENTICAL inputs: bit for bit (EXACT) -- every operation of the path is IEEE-exact on both
#    sides, the band LU uses the fused update of the reference's BLAS, exp / log are glibc's (tmx_refmath.h);
#  * device vs the REAL reference where the fixture carries the reference's own inputs (geometry, operators, halo
#    transforms: the per-call fixtures, the live ne15 run): bit for bit as well -- the oracle is the reference there;
#  * device vs reference fixtures on a SYNTHESISED grid (tempestmodel_amd/cubed_sphere.py reproduces the reference's
#    geometry to 1e-15, not bit for bit): the north-star tolerance 1e-10, every variable, no exception.
EXACT = gu.exact_tolerance()      # 0.0 on the project's hosts (golden_util.exact_tolerance)


TOL_CALL = EXACT      # one hot-path call


TOL_STEPS = 1e-10     # north-star tolerance for multi-step state differences on synthesised inputs


def _cmp(e, o, ie, io, tol, what):
    e.sync()
    errs = gu.prognostic_errors(e.download_state(ie), o.get_state(io))
    assert max(errs) <= tol, (what, errs)
    return errs


UDIFF = (1500.0, 500.0)


def _supercell_case():
    d = gu.load("supercell_ne2_L40_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=True, case="supercell")      # the reference's own (flat) geometry
    for P in g.patches:
        for nm in ("ref_node", "ref_redge", "ref_tracers"):
            P.geom[nm] = d["p%d/%s" % (P.index, nm)]
    return d, g


def _rank_engines_step(ranks, step_fn, nsteps):
    """nsteps of step_fn(engine, k) on every rank engine, one host thread each (the loopback wire has barriers inside)."""
    import threading
    errors = []

    def run(e):
        try:
            for k in range(nsteps):
                step_fn(e, k)
            e.sync()
        except Exception as ex:
            errors.append(ex)

    th = [threading.Thread(target=run, args=(e,)) for e in ranks]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errors and not any(t.is_alive() for t in th), errors


INFO_EARLY_TILES, INFO_LATE_TILES = 7, 8      # tmx_info: tiles launched before / while the exchange is in flight
