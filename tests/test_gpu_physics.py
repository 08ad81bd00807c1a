"""Column physics on the device (Held-Suarez forcing, Kessler microphysics) and the stock cases that use them (supercell + Kessler at L40, Held-Suarez at L60) against the oracle and the reference fixtures."""
import os
import numpy as np
import pytest
import golden_util as gu
from parity_common import EXACT, TOL_CALL, TOL_STEPS, _cmp, UDIFF, _supercell_case, _rank_engines_step, INFO_EARLY_TILES, INFO_LATE_TILES      # noqa: F401

pytestmark = pytest.mark.gpu


def test_held_suarez_physics():
    """the column physics kernel against the C oracle and the reference's own output
    (HeldSuarezPhysics::Perform on the JW state after 2 steps); only exp/log/pow/sin/cos differ from the CPU."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    d = gu.load("physics_hs_ne2_L30_p6.npz")
    g, _ = gu.make_grid(2, 30, 6)
    for P in g.patches:
        P.geom["hs_surface_pressure"] = d["p%d/hs_surface_pressure" % P.index]
    before, after = gu.full_states(d, "hs_before", 6), gu.full_states(d, "hs_after", 6)
    o = Oracle(g); o.set_state(0, before); o.held_suarez(0, 200.0)
    e = Engine(g)
    try:
        with pytest.raises(Exception):
            e.held_suarez(0, 200.0)                    # inputs not set yet
        e.set_physics_inputs([d["p%d/hs_surface_pressure" % p] for p in range(6)])
        e.upload_state(0, before)
        e.held_suarez(0, 200.0)
        e.sync()
        got = e.download_state(0)
        # pow / sin / cos of the forcing are ocml's on the device and glibc's in the oracle: rounding-level tolerance
        assert max(gu.prognostic_errors(got, o.get_state(0))) < 1e-12
        assert max(gu.prognostic_errors(got, after)) < 1e-12
    finally:
        e.close()


def test_kessler_physics_on_the_supercell_state():
    """KesslerPhysics::Perform on the device against the reference dump: the stock supercell state after 2 steps
    (no cloud yet: nearly the identity) and a moistened copy on which every branch of SUBROUTINE KESSLER acts.  exp / log /
    pow are glibc's restated bit for bit (tmx_refmath.h), sqrt is IEEE: the results are the reference's doubles -- state,
    tracers and precipitation."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    d, g = _supercell_case()
    dt = float(d["cfg/dt"][0])
    e = Engine(g, fully_explicit=True, uniform_diffusion=(1500.0, 500.0), nu=(0.0, 0.0, 0.0))
    try:
        e.set_level_heights()
        for before, after in (("warm", "kessler_after"), ("kessler_moist_before", "kessler_moist_after")):
            st = gu.expand_compact(d, before, g); tr = gu.expand_compact_tracers(d, before, g)
            e.upload_state(0, st); e.upload_tracers(0, tr)
            o = Oracle(g, fully_explicit=True, uniform_diffusion=(1500.0, 500.0))
            o.set_state(0, st); o.set_tracers(0, tr)
            o.kessler(0, dt, [P.geom["z_levels"] for P in g.patches])
            e.kessler(0, dt); e.sync()
            gs, gt = e.download_state(0), e.download_tracers(0)
            errs_o = gu.prognostic_errors(gs, o.get_state(0)); terr_o = gu.tracer_errors(gt, o.get_tracers(0))
            errs_r = gu.prognostic_errors(gs, gu.expand_compact(d, after, g)); terr_r = gu.tracer_errors(gt, gu.expand_compact_tracers(d, after, g))
            print(after, "vs oracle", errs_o, terr_o, "vs reference", errs_r, terr_r)
            assert max(errs_o) <= EXACT and max(errs_r) <= EXACT, (after, errs_o, errs_r)
            assert max(terr_o) <= EXACT and max(terr_r) <= EXACT, (after, terr_o, terr_r)
        pr = e.download_precipitation()
        for P in g.patches:
            ref = d["p%d/kessler_prect" % P.index][0][1:-1, 1:-1]
            assert np.array_equal(pr[P.index][1:-1, 1:-1], ref)
        assert max(float(np.max(a)) for a in pr.values()) > 1e-6
    finally:
        e.close()


@pytest.mark.parametrize("walk", [None, 0, 4, 6, 10, -1, -2, -3, -8])
def test_supercell_steps_with_kessler(walk, monkeypatch):
    """BASELINE config 4 end to end at ne2 L40: SupercellTest's own state (reference dump after 2 steps), 3 ARS343 steps
    with uniform diffusion, fully explicit vertical dynamics, three tracers, nu = 0, KesslerPhysics::Perform after every
    step (Model.cpp:470-481) -- against the oracle on identical inputs and against the reference's state.  walk: the explicit
    tracer update by column segments of that many levels, negative: by a sliding window over each of that many segments of a column
    (option "vt_explicit_walk"; 0 = the LDS-tiled kernel; None = the default)."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    if walk is not None:
        monkeypatch.setenv("TMX_VT_WALK", str(walk))
    d, g = _supercell_case()
    dt = float(d["cfg/dt"][0])
    st = gu.expand_compact(d, "warm", g); tr = gu.expand_compact_tracers(d, "warm", g)
    o = Oracle(g, nu=(0.0, 0.0, 0.0), fully_explicit=True, uniform_diffusion=(1500.0, 500.0))
    o.set_state(0, st); o.set_tracers(0, tr)
    e = Engine(g, fully_explicit=True, uniform_diffusion=(1500.0, 500.0), nu=(0.0, 0.0, 0.0), flavour="experiments" if (walk or 0) > 0 else None)
    try:
        e.set_level_heights()
        e.upload_state(0, st); e.upload_tracers(0, tr)
        zl = [P.geom["z_levels"] for P in g.patches]
        for _ in range(3):
            assert o.step_ars343(dt) == 0
            o.kessler(0, dt, zl)
            e.step_ars343(dt)
            e.kessler(0, dt)
        e.sync()
        gs, gt = e.download_state(0), e.download_tracers(0)
        errs_o = gu.prognostic_errors(gs, o.get_state(0)); terr_o = gu.tracer_errors(gt, o.get_tracers(0))
        errs_r = gu.prognostic_errors(gs, gu.expand_compact(d, "step3", g)); terr_r = gu.tracer_errors(gt, gu.expand_compact_tracers(d, "step3", g))
        print("supercell 3 steps vs oracle", errs_o, terr_o, "vs reference", errs_r, terr_r)
        # identical doubles: exp / log / pow are glibc's, restated bit for bit (tmx_refmath.h)
        assert max(errs_o) <= EXACT and max(errs_r) <= EXACT
        assert max(terr_o) <= EXACT and max(terr_r) <= EXACT
    finally:
        e.close()


@pytest.mark.parametrize("graph", [False, True])
def test_held_suarez_steps_L60(graph, monkeypatch):
    """HeldSuarezTest (ztop 35 km, Rayleigh layer of strength zero) at ne3 L60: 10 ARS343 steps with
    HeldSuarezPhysics::Perform after every step (Model.cpp:470-481), from the reference's state after 2 such steps, on
    the reference's own grid; the forcing reads the tracked surface slots.  graph: the step replayed from a hipGraph
    (TMX_GRAPH=1, BASELINE config 5's 'hipGraph-captured timestep').  The oracle equals the reference bit for bit here
    (tests/test_oracle_vs_reference.py); so does the device: exp / log / pow are glibc's, restated bit for bit
    (tmx_refmath.h), sin / cos of the latitude are evaluated once on the host with the reference's libm."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    if graph:
        monkeypatch.setenv("TMX_GRAPH", "1")
    d = gu.load("heldsuarez_ne3_L60_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=True, case="heldsuarez")
    for P in g.patches:
        P.lat = d["p%d/hs_latitude" % P.index]
    dt = float(d["cfg/dt"][0])
    start = gu.expand_compact(d, "warm", g)
    o = Oracle(g); o.set_state(0, start)
    e = Engine(g)
    try:
        e.set_physics_inputs(None)
        e.upload_state(0, start)
        for _ in range(10):
            assert o.step_ars343(dt) == 0
            o.held_suarez(0, dt)
            e.step_ars343(dt)
            e.held_suarez(0, dt)
        e.sync()
        got = e.download_state(0)
        errs_o = gu.prognostic_errors(got, o.get_state(0))
        errs_r = gu.prognostic_errors(got, gu.expand_compact(d, "step10", g))
        print("Held-Suarez L60, 10 steps (graph=%s): vs oracle %s vs reference %s" % (graph, errs_o, errs_r))
        assert max(gu.prognostic_errors(o.get_state(0), gu.expand_compact(d, "step10", g))) == 0.0
        assert max(errs_o) <= EXACT and max(errs_r) <= EXACT
        # the tracked surface slots came back with the state: the reference's interface entries, bit for bit
        for (gn, ge), (on, oe) in zip(got, o.get_state(0)):
            assert np.array_equal(ge[[2, 4], 1:-1, 1:-1, 0], oe[[2, 4], 1:-1, 1:-1, 0])
    finally:
        e.close()


def test_kessler_level_parallel_kernel_with_several_rain_sub_cycles(monkeypatch):
    """SUBROUTINE KESSLER's rain loop on the moistened supercell state with time steps long enough for the CFL limit to split it
    (kessler.f90:117-127; the columns need different numbers of sub-cycles): the level-parallel kernel (default), the
    one-lane-per-column kernel (TMX_KESSLER_COLUMN=1) and the oracle give the same doubles -- state, tracers, precipitation."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    d, g = _supercell_case()
    st = gu.expand_compact(d, "kessler_moist_before", g); tr = gu.expand_compact_tracers(d, "kessler_moist_before", g)
    zl = [P.geom["z_levels"] for P in g.patches]
    for dt in (60.0, 400.0):
        o = Oracle(g, fully_explicit=True, uniform_diffusion=(1500.0, 500.0))
        o.set_state(0, st); o.set_tracers(0, tr)
        o.kessler(0, dt, zl); o.kessler(0, dt, zl)
        out = []
        for column in (False, True):
            monkeypatch.delenv("TMX_KESSLER_COLUMN", raising=False)
            if column:
                monkeypatch.setenv("TMX_KESSLER_COLUMN", "1")
            e = Engine(g, fully_explicit=True, uniform_diffusion=(1500.0, 500.0), nu=(0.0, 0.0, 0.0))
            try:
                e.set_level_heights()
                e.upload_state(0, st); e.upload_tracers(0, tr)
                e.kessler(0, dt); e.kessler(0, dt); e.sync()
                out.append((e.download_state(0), e.download_tracers(0), e.download_precipitation()))
            finally:
                e.close()
        for gs, gt, pr in out:
            assert max(gu.prognostic_errors(gs, o.get_state(0))) <= EXACT, dt
            assert max(gu.tracer_errors(gt, o.get_tracers(0))) <= EXACT, dt
        for P in g.patches:
            assert np.array_equal(out[0][2][P.index][1:-1, 1:-1], out[1][2][P.index][1:-1, 1:-1])
        assert max(float(np.max(a)) for a in out[0][2].values()) > 0.0
