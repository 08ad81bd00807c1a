"""Archived experiments and the opt-in matrix-unit contractions (experiments flavour of the library): each against the production path."""
import os
import numpy as np
import pytest
import golden_util as gu
from parity_common import EXACT, TOL_CALL, TOL_STEPS, _cmp, UDIFF, _supercell_case, _rank_engines_step, INFO_EARLY_TILES, INFO_LATE_TILES      # noqa: F401

pytestmark = pytest.mark.gpu


def test_mfma_contractions_agree_to_rounding(monkeypatch):
    """TMX_MFMA=1: the 4x4 contractions of the fused explicit kernel on the matrix unit (v_mfma_f64_4x4x4_4b_f64).  The
    matrix unit accumulates with fused multiply-adds, so the result is NOT the reference's bit for bit (which is why the
    path is opt-in): three whole steps agree with the oracle to 1e-13 (W, the residual of the hydrostatic balance: 1e-11)."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    monkeypatch.setenv("TMX_MFMA", "1")
    d = gu.load("percall_ne3_L5_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=True)
    dt = float(d["cfg/dt"][0])
    start = gu.full_states(d, "warm", 6)
    o = Oracle(g); o.set_state(0, start)
    e = Engine(g); e.upload_state(0, start)
    try:
        worst = 0.0
        for _ in range(3):
            assert o.step_ars343(dt) == 0
            e.step_ars343(dt)
            e.sync()
            errs = gu.prognostic_errors(e.download_state(0), o.get_state(0))
            worst = max(worst, max(errs))
            assert max(errs[c] for c in (0, 1, 2, 4)) < 1e-13 and errs[3] < 1e-11, errs      # W amplifies last-bit changes
        print("MFMA contractions, 3 steps vs oracle:", errs)
        assert worst > 0.0          # it really is a different arithmetic (otherwise the opt-in would be pointless)
    finally:
        e.close()


def test_mfma_contractions_against_the_north_star_gate(monkeypatch):
    """Judge row N1, settled with the gate: BASELINE config 2 (ne15 np4 L30, dt 300 s), 100 ARS343 steps from a developed
    state with the contractions of the explicit kernel on the matrix unit (TMX_MFMA=1) against the default path on
    identical inputs -- which is the reference bit for bit (test_hundred_steps_ne15_L30), so the difference measured here
    IS the difference to the reference.  Measured (tools/mfma_gate.py, deterministic): U 6.6e-13, V 5.8e-13,
    rho*theta 5.4e-15, W 1.5e-10, rho 2.8e-14 -- every variable but W is far inside the 1e-10 gate, W (the small residual
    of the hydrostatic balance, which amplifies last-bit changes about a hundredfold per hundred steps) is just outside.
    The matrix unit accumulates with fused multiply-adds, the reference with separate multiplies and adds; that cannot be
    made exact, so the path stays opt-in.  The test pins the measured bound."""
    from tempestmodel_amd.engine import Engine
    ne, L, dt = 15, 30, 300.0
    g, states = gu.make_grid(ne, L, 6)
    exact = Engine(g)
    monkeypatch.setenv("TMX_MFMA", "1")
    mfma = Engine(g)
    monkeypatch.delenv("TMX_MFMA")
    try:
        exact.upload_state(0, states)
        for _ in range(3):
            exact.step_ars343(dt)
        exact.sync()
        mfma.upload_state(0, exact.download_state(0))
        for _ in range(100):
            exact.step_ars343(dt); mfma.step_ars343(dt)
        exact.sync(); mfma.sync()
        errs = gu.prognostic_errors(mfma.download_state(0), exact.download_state(0))
        print("MFMA contractions vs the exact path, 100 steps at ne15 L30:", errs, "gate (1e-10) passed by W:", errs[3] < 1e-10)
        assert max(errs[c] for c in (0, 1)) < 5e-12 and max(errs[c] for c in (2, 4)) < 5e-13, errs
        assert 0.0 < errs[3] < 5e-10, errs
    finally:
        exact.close(); mfma.close()


def test_dss_pulled_into_the_second_hyperviscosity_pass_is_bit_identical(monkeypatch):
    """TMX_HVIS_PULL=1 (an archived experiment: slower): the second hyperviscosity pass of the steppers'
    programs averages the first pass's Laplacians while loading them, the DSS pass between the two is not run.  Three
    ARS343 steps on 24 patches (cube corners, panel edges with rotated (U,V), patch seams) equal the default path's
    bit for bit, and so does a 3-rank run through the loopback wire (remote copies from the ghost buffer)."""
    import threading
    from tempestmodel_amd.engine import Engine
    g, start = gu.make_grid(12, 6, 24)
    ref_e = Engine(g)
    monkeypatch.setenv("TMX_HVIS_PULL", "1")      # (an archived kernel: experiments flavour of the library)
    pull = Engine(g, flavour="experiments")
    ranks = [Engine(g, rank=r, n_ranks=3, flavour="experiments") for r in range(3)]
    monkeypatch.delenv("TMX_HVIS_PULL")
    try:
        for e in [ref_e, pull] + ranks:
            e.upload_state(0, start)
        for e in (ref_e, pull):
            for k in range(3):
                e.step("ars343", 60.0, first=(k == 0))
            e.sync()
        ref = ref_e.download_state(0)
        got = pull.download_state(0)
        for p in range(24):
            assert np.array_equal(got[p][0][[0, 1, 2, 4], 1:-1, 1:-1], ref[p][0][[0, 1, 2, 4], 1:-1, 1:-1]), p
            assert np.array_equal(got[p][1][3, 1:-1, 1:-1], ref[p][1][3, 1:-1, 1:-1]), p
        Engine.loopback_group(ranks)
        errors = []

        def run(e):
            try:
                for k in range(3):
                    e.step("ars343", 60.0, first=(k == 0))
                e.sync()
            except Exception as ex:
                errors.append(ex)

        th = [threading.Thread(target=run, args=(e,)) for e in ranks]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=120)
        assert not errors and not any(t.is_alive() for t in th), errors
        Engine.loopback_dissolve(ranks[0])
        for e in ranks:
            got = e.download_state(0)
            for p in e.local_patches:
                assert np.array_equal(got[p][0][[0, 1, 2, 4], 1:-1, 1:-1], ref[p][0][[0, 1, 2, 4], 1:-1, 1:-1]), (e.rank, p)
                assert np.array_equal(got[p][1][3, 1:-1, 1:-1], ref[p][1][3, 1:-1, 1:-1]), (e.rank, p)
    finally:
        for e in [ref_e, pull] + ranks:
            e.close()


@pytest.mark.parametrize("ne,npatch", [(12, 24), (4, 24), (5, 6), (7, 6)])
def test_hyperviscosity_fused_with_the_in_patch_dss_is_bit_identical(ne, npatch, monkeypatch):
    """k_hypervis_block (TMX_HVIS_BLOCK=1: an archived experiment, slower) against k_hypervis + the full k_dss: patches of 6 x 6, 2 x 2, 5 x 5 and
    7 x 7 elements (blocks of 5 x 5 inner elements: clipped blocks, blocks without a ring, single-element remainders), three
    ARS343 steps, every patch, bit for bit -- the per-call ABI too (the working instance afterwards holds the DSS'ed Laplacians)."""
    from tempestmodel_amd.engine import Engine
    g, start = gu.make_grid(ne, 6, npatch)
    res = []
    for block in ("1", "0"):
        monkeypatch.setenv("TMX_HVIS_BLOCK", block)
        e = Engine(g, flavour="experiments")      # (an archived kernel: experiments flavour of the library)
        try:
            e.upload_state(0, start)
            for k in range(3):
                e.step("ars343", 720.0 / ne, first=(k == 0))
            e.h_step_after_subcycle(0, 1, 2, 100.0)
            e.sync()
            res.append([e.download_state(i) for i in (0, 1, 2)])
        finally:
            e.close()
    for a_inst, b_inst in zip(res[0], res[1]):
        for (an, ae), (bn, be) in zip(a_inst, b_inst):
            assert np.array_equal(an[[0, 1, 2, 4], 1:-1, 1:-1], bn[[0, 1, 2, 4], 1:-1, 1:-1])
            assert np.array_equal(ae[3, 1:-1, 1:-1], be[3, 1:-1, 1:-1])


@pytest.mark.parametrize("ne,npatch,shape,blocks", [(5, 6, 4, -1), (7, 6, 4, -1), (6, 24, 4, -1), (5, 6, 3, 1), (6, 24, 0, 1), (5, 6, 4, 0)])
@pytest.mark.parametrize("scheme", ["ars343", "strang", "ark232"])
def test_block_kernels_and_block_thread_orders_are_bit_identical(scheme, ne, npatch, shape, blocks):
    """Round 6's archived experiment: thread orders 3 (generalised Hilbert curve) and 4 (4 x 4 element blocks, padded at ragged patch
    edges: 5 x 5 and 7 x 7 element patches here) and the block kernels -- a workgroup of four wavefronts averages the seams between its
    wavefronts through LDS (k_h_walk<.., BLK>, k_hypervis<.., BLK>), the DSS kernel runs the shorter list.  Three steps after the
    element-major first one, against the element-major engine, bit for bit."""
    from tempestmodel_amd.engine import Engine
    g, start = gu.make_grid(ne, 6, npatch)
    ni = max(7, Engine.scheme_instances(scheme))
    d = Engine(g, n_instances=ni, options={"unique_layout": 0})
    u = Engine(g, n_instances=ni, options={"unique_layout": 1, "unique_tile_shape": shape, "unique_blocks": blocks}, flavour="experiments")
    try:
        assert u.info(12) == 1 and u.get_option("unique_tile_shape") == shape
        d.upload_state(0, start); u.upload_state(0, start)
        for k in range(4):
            d.step(scheme, 150.0, first=(k == 0)); u.step(scheme, 150.0, first=(k == 0))
            if k >= 1 and not (scheme == "ark232" and npatch == 24):      # (ARK232 on grids whose metric copies differ runs element-major: test_steps_on_the_unique_layout_are_bit_identical)
                assert u.info(13) > 0, "step %d did not run on the node-unique layout" % k
        d.sync(); u.sync()
        a, b = d.download_state(0), u.download_state(0)
        for p_ in range(npatch):
            assert np.array_equal(a[p_][0], b[p_][0]) and np.array_equal(a[p_][1], b[p_][1]), (scheme, p_)
    finally:
        d.close(); u.close()
