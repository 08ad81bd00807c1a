"""Whole steps of the ten stepper programs (ARS343 / 232 / 222 / 443, ARK232, Strang and its explicit discretisations, shallow water) against the oracle and the golden vectors of the real reference, and the bit-identity of the fused production programs (shared copies, shared U,V slabs, graph replay)."""
import os
import numpy as np
import pytest
import golden_util as gu
from parity_common import EXACT, TOL_CALL, TOL_STEPS, _cmp, UDIFF, _supercell_case, _rank_engines_step, INFO_EARLY_TILES, INFO_LATE_TILES      # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fixture,ne,L,npatch,dt,tags", [
    ("steps_ne4_L6_p24.npz", 4, 6, 24, 200.0, ["step1", "step2", "step3"]),
    ("steps_ne3_L5_p6.npz", 3, 5, 6, 300.0, ["step1", "step2", "step5", "step10"]),
    ("steps_schar_ne3_L6_p6.npz", 3, 6, 6, 0.5, ["step1", "step2", "step3"]),
])
def test_ars343_steps_vs_reference_and_oracle(fixture, ne, L, npatch, dt, tags):
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    d = gu.load(fixture)
    g, _ = gu.make_grid(ne, L, npatch, case="schar" if "schar" in fixture else "jw")
    start = gu.expand_compact(d, tags[0], g)
    o = Oracle(g); o.set_state(0, start)
    e = Engine(g); e.upload_state(0, start)
    step = int(tags[0][4:])
    try:
        for tag in tags[1:]:
            while step < int(tag[4:]):
                assert o.step_ars343(dt) == 0
                e.step_ars343(dt)
                step += 1
            e.sync()
            got = e.download_state(0)
            assert max(gu.prognostic_errors(got, o.get_state(0))) == 0.0, tag
            assert max(gu.prognostic_errors(got, gu.expand_compact(d, tag, g))) < TOL_STEPS, tag
    finally:
        e.close()


def test_tracer_steps_vs_reference_and_oracle():
    """Whole ARS343 steps with two tracers (production program incl. the fused paths) from the reference's state after
    step 1; also the other steppers' generic operations through ARS232."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    d = gu.load("steps_tracers_ne3_L6_p6.npz")
    g, _ = gu.make_grid(3, 6, 6, ntracers=2)
    st, tr = gu.expand_compact(d, "step1", g), gu.expand_compact_tracers(d, "step1", g)
    o = Oracle(g); o.set_state(0, st); o.set_tracers(0, tr)
    e = Engine(g); e.upload_state(0, st); e.upload_tracers(0, tr)
    try:
        for step in (2, 3, 4):
            assert o.step_ars343(200.0) == 0
            e.step_ars343(200.0)
            e.sync()
            got = e.download_tracers(0)
            assert max(gu.tracer_errors(got, o.get_tracers(0))) == 0.0, step
            assert max(gu.tracer_errors(got, gu.expand_compact_tracers(d, "step%d" % step, g))) < 1e-11, step
            assert max(gu.prognostic_errors(e.download_state(0), gu.expand_compact(d, "step%d" % step, g))) < TOL_STEPS, step
        for _ in range(2):
            assert o.step("ars232", 200.0) == 0
            e.step("ars232", 200.0)
        e.sync()
        assert max(gu.tracer_errors(e.download_tracers(0), o.get_tracers(0))) == 0.0
    finally:
        e.close()


def test_ark232_with_tracers():
    """SURVEY 8a row V2 with tracers: tmx_step(ARK232) with 2 tracers -- StepImplicitTermsExplicitly followed by the implicit
    column update of the tracers on every stored column -- against the reference's state / tracers after step 3 and the
    oracle, bit for bit; and the single call against the oracle on a state whose duplicated nodes differ."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    d = gu.load("steps_ark232_tracers_ne2_L5_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=True, ntracers=2)
    st, tr = gu.expand_compact(d, "step1", g), gu.expand_compact_tracers(d, "step1", g)
    dt = float(d["cfg/dt"][0])
    o = Oracle(g, ninst=8); o.set_state(0, st); o.set_tracers(0, tr)
    e = Engine(g, n_instances=8); e.upload_state(0, st); e.upload_tracers(0, tr)
    try:
        for _ in range(2):
            assert o.step("ark232", dt) == 0
            e.step("ark232", dt)
        e.sync()
        got_s, got_t = e.download_state(0), e.download_tracers(0)
        assert max(gu.tracer_errors(got_t, o.get_tracers(0))) <= EXACT
        assert max(gu.prognostic_errors(got_s, o.get_state(0))) <= EXACT
        assert max(gu.tracer_errors(got_t, gu.expand_compact_tracers(d, "step3", g))) <= EXACT
        assert max(gu.prognostic_errors(got_s, gu.expand_compact(d, "step3", g))) <= EXACT
        # the call on its own, from a state before any DSS (duplicated element-edge columns hold different data: the
        # reference updates every node from its own column there, VerticalDynamicsFEM.cpp:541-542)
        o.copy_data(0, 1); o.h_step_explicit(0, 1, dt); o.copy_data(1, 2)
        e.upload_state(1, o.get_state(1)); e.upload_tracers(1, o.get_tracers(1))
        e.upload_state(2, o.get_state(2)); e.upload_tracers(2, o.get_tracers(2))
        o.v_step_implicit_terms_explicitly(1, 2, 0.3 * dt)
        e.v_step_implicit_terms_explicitly(1, 2, 0.3 * dt)
        e.sync()
        assert max(gu.tracer_errors(e.download_tracers(2), o.get_tracers(2))) <= EXACT
        assert max(gu.prognostic_errors(e.download_state(2), o.get_state(2))) <= EXACT
    finally:
        e.close()


@pytest.mark.parametrize("scheme", ["ars343", "strang"])
def test_uniform_diffusion_explicit_vertical_steps(scheme):
    """Whole steps of the supercell configuration through the production program (fused paths included)."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    d = gu.load("steps_udiff_%s_ne3_L6_p6.npz" % scheme)
    g, _ = gu.make_grid(3, 6, 6, case="smallplanet", ntracers=2)
    st, tr = gu.expand_compact(d, "initial", g), gu.expand_compact_tracers(d, "initial", g)
    o = Oracle(g, fully_explicit=True, uniform_diffusion=UDIFF); o.set_state(0, st); o.set_tracers(0, tr)
    e = Engine(g, fully_explicit=True, uniform_diffusion=UDIFF); e.upload_state(0, st); e.upload_tracers(0, tr)
    try:
        for step in (1, 2, 3):
            assert o.step(scheme, 1.0, first=(step == 1)) == 0
            e.step(scheme, 1.0, first=(step == 1))
            e.sync()
            gs, gt = e.download_state(0), e.download_tracers(0)
            assert max(gu.prognostic_errors(gs, o.get_state(0))) == 0.0, step
            assert max(gu.tracer_errors(gt, o.get_tracers(0))) == 0.0, step
            assert max(gu.prognostic_errors(gs, gu.expand_compact(d, "step%d" % step, g))) < TOL_STEPS, step
            assert max(gu.tracer_errors(gt, gu.expand_compact_tracers(d, "step%d" % step, g))) < 1e-11, step
    finally:
        e.close()


def test_explicit_vertical_without_diffusion_matches_oracle():
    """--explicitvertical alone (no uniform diffusion, no tracers): ARS343 and Strang steps against the C oracle."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    g, st = gu.make_grid(3, 6, 6, case="smallplanet")
    o = Oracle(g, fully_explicit=True); o.set_state(0, st)
    e = Engine(g, fully_explicit=True); e.upload_state(0, st)
    try:
        for step, scheme in enumerate(("ars343", "ars343", "strang", "strang")):
            assert o.step(scheme, 1.0, first=(step == 2)) == 0
            e.step(scheme, 1.0, first=(step == 2))
        e.sync()
        assert max(gu.prognostic_errors(e.download_state(0), o.get_state(0))) == 0.0
    finally:
        e.close()


def test_viscosity_order_2_steps():
    """Hyperviscosity order 2 (one viscosity pass): ARS343 steps vs the C oracle and the reference fixture."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    d = gu.load("steps_visc2_ne3_L5_p6.npz")
    g, _ = gu.make_grid(3, 5, 6)
    nu = (2.0e5, 2.0e5, 2.0e5)
    start = gu.expand_compact(d, "step1", g)
    o = Oracle(g, nu=nu, hypervis_order=2); o.set_state(0, start)
    e = Engine(g, nu=nu, hypervis_order=2); e.upload_state(0, start)
    try:
        for step in (2, 3):
            assert o.step_ars343(300.0) == 0
            e.step_ars343(300.0)
            e.sync()
            got = e.download_state(0)
            assert max(gu.prognostic_errors(got, o.get_state(0))) == 0.0
            assert max(gu.prognostic_errors(got, gu.expand_compact(d, "step%d" % step, g))) < TOL_STEPS
    finally:
        e.close()


def test_graph_replay_is_bit_identical(monkeypatch):
    """TMX_GRAPH=1: ARS343 and Strang steps replayed from a captured hipGraph equal the plain launch sequence bit for bit."""
    from tempestmodel_amd.engine import Engine
    g, st = gu.make_grid(4, 8, 6)
    res = []
    for graph in ("0", "1"):
        monkeypatch.setenv("TMX_GRAPH", graph)
        e = Engine(g)
        try:
            e.upload_state(0, st)
            for _ in range(4):
                e.step_ars343(300.0)
            e.sync()
            res.append(e.download_state(0))
        finally:
            e.close()
    for (na, ea), (nb, eb) in zip(res[0], res[1]):
        assert np.array_equal(na, nb) and np.array_equal(ea, eb)


@pytest.mark.parametrize("ne,L,npatch,ntr", [
    (2, 3, 6, 0),        # minimum level count the engine accepts
    (4, 5, 96, 0),       # one element per patch: every column sits on a patch boundary
    (6, 7, 24, 1),       # odd level count, 3x3 elements per patch, one tracer
    (3, 33, 6, 0),       # more levels than a 32-entry anything
])
def test_unusual_sizes_vs_oracle(ne, L, npatch, ntr):
    """Edge sizes: two ARS343 steps and one Strang step against the C oracle (developed start state from the oracle)."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    g, states = gu.make_grid(ne, L, npatch, ntracers=ntr)
    o = Oracle(g); o.set_state(0, states)
    if ntr:
        o.set_tracers(0, [g.initial_tracers[p] for p in range(npatch)])
    assert o.step_ars343(100.0) == 0
    e = Engine(g); e.upload_state(0, o.get_state(0))
    if ntr:
        e.upload_tracers(0, o.get_tracers(0))
    try:
        for _ in range(2):
            assert o.step_ars343(100.0) == 0
            e.step_ars343(100.0)
        assert o.step("strang", 100.0, first=True) == 0
        e.step("strang", 100.0, first=True)
        e.sync()
        assert max(gu.prognostic_errors(e.download_state(0), o.get_state(0))) == 0.0
        if ntr:
            assert max(gu.tracer_errors(e.download_tracers(0), o.get_tracers(0))) == 0.0
    finally:
        e.close()


def test_hundred_steps_ne4_L10():
    """North-star gate (fp64 state difference < 1e-10 after 100 ARS343 steps) against the oracle on identical inputs:
    the device reproduces it BIT FOR BIT."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    g, states = gu.make_grid(4, 10, 6)
    o = Oracle(g); o.set_state(0, states)
    for _ in range(2):                      # leave the balanced initial state on the CPU first
        assert o.step_ars343(200.0) == 0
    start = o.get_state(0)
    e = Engine(g); e.upload_state(0, start)
    try:
        for _ in range(100):
            assert o.step_ars343(200.0) == 0
            e.step_ars343(200.0)
        e.sync()
        errs = gu.prognostic_errors(e.download_state(0), o.get_state(0))
        print("100 steps:", errs)
        assert max(errs) <= EXACT, errs
    finally:
        e.close()


def test_hundred_steps_ne4_L10_vs_reference():
    """Same gate against the REAL reference: 100 ARS343 steps from its state after 2 steps (golden fixture), on the
    synthesised grid (inputs agree with the reference's to 1e-15, not bit for bit)."""
    from tempestmodel_amd.engine import Engine
    d = gu.load("steps100_ne4_L10_p6.npz")
    g, _ = gu.make_grid(4, 10, 6)
    e = Engine(g); e.upload_state(0, gu.expand_compact(d, "warm", g))
    try:
        for _ in range(100):
            e.step_ars343(200.0)
        e.sync()
        errs = gu.prognostic_errors(e.download_state(0), gu.expand_compact(d, "step100", g))
        print("100 steps vs reference:", errs)
        assert max(errs) < TOL_STEPS, errs
    finally:
        e.close()


@pytest.mark.parametrize("scheme", ["ars232", "ars222", "ars443", "ark232", "strang"])
def test_other_time_schemes(scheme):
    """SURVEY 8a rows T2/T3 on the GPU: tmx_step for every stepper of the path vs the reference golden state
    and the oracle (two steps from the reference's state after step 1)."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    d = gu.load("schemes_ne2_L4_p6.npz")
    g, _ = gu.make_grid(2, 4, 6)
    o = Oracle(g, ninst=10)
    e = Engine(g, n_instances=10)
    try:
        for setter in (o.set_state, e.upload_state):
            setter(0, gu.expand_compact(d, "%s_step1" % scheme, g))
            if scheme == "strang":
                setter(1, gu.expand_compact(d, "strang_step1_carry", g))
        assert Engine.scheme_instances(scheme) <= 10
        for _ in range(2):
            assert o.step(scheme, 400.0) == 0
            e.step(scheme, 400.0)
        e.sync()
        got = e.download_state(0)
        assert max(gu.prognostic_errors(got, o.get_state(0))) == 0.0
        assert max(gu.prognostic_errors(got, gu.expand_compact(d, "%s_step3" % scheme, g))) < TOL_STEPS
    finally:
        e.close()


STRANG_VARIANTS = [("strang_fe", 0.0, 50.0), ("strang_rk4", 0.0, 400.0), ("strang_ssp3", 0.0, 400.0), ("strang_ssprk53", 0.0, 400.0),
                   ("strang_rk4", 0.25, 400.0), ("strang", 0.5, 400.0)]


@pytest.mark.parametrize("scheme,oc,dt", STRANG_VARIANTS)
def test_strang_explicit_discretisations(scheme, oc, dt):
    """SURVEY 8a row T3, the rest of it: tmx_step with TimestepSchemeStrang's other ExplicitDiscretization values and a non-zero
    off-centering (tmx_set_strang_off_centering), two steps from the reference's state after step 1: bit for bit against
    the reference's state and carried instance after step 3, and the oracle."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    d = gu.load("strang_variants_ne2_L4_p6.npz")
    name = scheme + ("_oc%02d" % round(100 * oc) if oc else "")
    g, _ = gu.grid_from_fixture(d, override=True)
    o = Oracle(g, ninst=5)
    e = Engine(g, n_instances=5)
    try:
        assert Engine.scheme_instances(scheme) == 5
        for setter in (o.set_state, e.upload_state):
            setter(0, gu.expand_compact(d, "%s_step1" % name, g))
            setter(1, gu.expand_compact(d, "%s_step1_carry" % name, g))
        o.set_strang_off_centering(oc); e.set_strang_off_centering(oc)
        try:
            for _ in range(2):
                assert o.step(scheme, dt) == 0
                e.step(scheme, dt)
        finally:
            o.set_strang_off_centering(0.0)
        e.sync()
        for inst, tag in ((0, "step3"), (1, "step3_carry")):
            got = e.download_state(inst)
            assert max(gu.prognostic_errors(got, o.get_state(inst))) <= EXACT
            assert max(gu.prognostic_errors(got, gu.expand_compact(d, "%s_%s" % (name, tag), g))) <= EXACT
        with pytest.raises(Exception, match="outside"):
            e.set_strang_off_centering(1.5)
    finally:
        e.close()


def test_shallow_water_test2_strang():
    """BASELINE config 1 on the GPU: Williamson test 2 (ne8, Strang/KGU35, dt 200 s), 10 steps from the
    analytic initial state vs the reference's dumped state and the oracle."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    from test_oracle_vs_reference import _sw_setup, _sw_ref
    d = gu.load("sw2_ne8_strang.npz")
    g, states = _sw_setup()
    o = Oracle(g, ninst=5); o.set_state(0, states)
    e = Engine(g, n_instances=5)
    try:
        e.upload_state(0, states)
        for n in range(10):
            assert o.step("strang", 200.0, first=(n == 0)) == 0
            e.step("strang", 200.0, first=(n == 0))
        e.sync()
        got = e.download_state(0)
        assert max(gu.prognostic_errors(got, o.get_state(0))[:3]) == 0.0
        assert max(gu.prognostic_errors(got, _sw_ref(d, "step10", g))[:3]) < 1e-12
    finally:
        e.close()


@pytest.mark.parametrize("scheme", ["ars343", "ars232", "ars222", "ars443", "ark232", "strang"])
def test_stage_copies_left_out_and_fused_uv_update_are_bit_identical(monkeypatch, scheme):
    """Fully explicit vertical mode (BASELINE config 4's): the CopyData in front of the V.StepImplicit that does nothing is not
    performed -- the copy reads its source's slot (TMX_SHARE_COPIES=0: every copy made); TMX_VX_FUSED=1 runs V.StepExplicit's
    U,V update inside the kernel of the explicitly treated terms instead of its own launch; TMX_H_WALK_UDIFF chooses how much of the
    uniform diffusion and of V.StepExplicit the explicit stage's walk applies itself.  Same doubles, in instance 0
    and in every other instance a caller can ask for between steps (the work instance of StepAfterSubCycle aside)."""
    from tempestmodel_amd.engine import Engine
    d = gu.load("steps_udiff_ars343_ne3_L6_p6.npz")
    g, _ = gu.make_grid(3, 6, 6, case="smallplanet", ntracers=2)
    st, tr = gu.expand_compact(d, "initial", g), gu.expand_compact_tracers(d, "initial", g)
    ninst = Engine.scheme_instances(scheme)
    runs = []
    # (h_walk_udiff: 2 = default, the explicit stage's walk applies the horizontal uniform diffusion and V.StepExplicit's U,V part to its
    # results in registers; 1 = the diffusion alone, k_v_explicit_slide a pass of its own; 0 = k_uniform_diffusion a pass of its own too)
    for variant in ({}, {"TMX_SHARE_COPIES": "0"}, {"TMX_VX_FUSED": "1"}, {"probe": True}, {"TMX_H_WALK_UDIFF": "1"}, {"TMX_H_WALK_UDIFF": "0"},
                    {"TMX_H_WALK_UDIFF": "2", "TMX_UDV_SEPARATE": "1"}):
        monkeypatch.delenv("TMX_SHARE_COPIES", raising=False)
        monkeypatch.delenv("TMX_VX_FUSED", raising=False)
        monkeypatch.delenv("TMX_H_WALK_UDIFF", raising=False)
        monkeypatch.delenv("TMX_UDV_SEPARATE", raising=False)
        for k, v in variant.items():
            if k.startswith("TMX_"):
                monkeypatch.setenv(k, v)
        e = Engine(g, n_instances=ninst, fully_explicit=True, uniform_diffusion=UDIFF, flavour="experiments" if ("TMX_VX_FUSED" in variant or "TMX_UDV_SEPARATE" in variant) else None)
        try:
            e.upload_state(0, st); e.upload_tracers(0, tr)
            for step in (1, 2, 3):
                e.step(scheme, 1.0, first=(step == 1))
                if variant.get("probe"):
                    # a caller looks at (and so settles) stage instances between steps, and writes one back
                    for ix in range(1, ninst):
                        e.download_state(ix)
                    e.upload_state(ninst - 1, e.download_state(ninst - 1))
            e.sync()
            runs.append([(e.download_state(ix), e.download_tracers(ix)) for ix in range(ninst)])
        finally:
            e.close()
    work = {"ars343": {2}}.get(scheme, set(range(1, ninst)))      # compared beyond instance 0: ARS343's stage instances
    for other in runs[1:]:
        for ix in range(ninst):
            if ix != 0 and ix in work:
                continue
            assert max(gu.prognostic_errors(runs[0][ix][0], other[ix][0])) <= EXACT, (scheme, ix)
            assert max(gu.tracer_errors(runs[0][ix][1], other[ix][1])) <= EXACT, (scheme, ix)


def test_stage_copies_left_out_in_shallow_water_are_bit_identical(monkeypatch):
    """Shallow water (V is a stub): the same, ARS343 on Williamson test 2."""
    from tempestmodel_amd.engine import Engine
    from test_oracle_vs_reference import _sw_setup
    g, states = _sw_setup()
    out = []
    for share in ("1", "0"):
        monkeypatch.setenv("TMX_SHARE_COPIES", share)
        e = Engine(g, n_instances=7)
        try:
            e.upload_state(0, states)
            for n in range(4):
                e.step("ars343", 200.0, first=(n == 0))
            e.sync()
            out.append([e.download_state(ix) for ix in (0, 1, 3, 4, 5, 6)])
        finally:
            e.close()
    for a, b in zip(*out):
        assert max(gu.prognostic_errors(a, b)[:3]) <= EXACT


@pytest.mark.parametrize("ntr", [0, 2])
@pytest.mark.parametrize("scheme", ["ars343", "ars232", "ars222", "ars443", "ark232", "strang"])
def test_uv_slabs_shared_in_front_of_the_column_solve_are_bit_identical(monkeypatch, scheme, ntr):
    """Implicit vertical dynamics: the CopyData in front of V.StepImplicit is not made at all -- the solve rewrites rho*theta,
    W, rho of the copy, and its U,V slabs are read from the source's slot by the fused explicit stage and the tracer flux kernel
    (TMX_SHARE_COPIES=0: U,V copied as before).  Same doubles in instance 0 and in every stage instance a caller asks for
    between steps, without and with tracers."""
    from tempestmodel_amd.engine import Engine
    g, states = gu.make_grid(4, 8, 6, ntracers=ntr)
    tracers = [np.stack([(0.01 + 0.002 * t) * n[4] * (1.0 + 0.1 * np.sin(3.0 * t + n[0] / 20.0)) for t in range(ntr)]) for n, _ in states] if ntr else None
    ninst = Engine.scheme_instances(scheme)
    runs = []
    for variant in ({}, {"TMX_SHARE_COPIES": "0"}, {"probe": True}):
        monkeypatch.delenv("TMX_SHARE_COPIES", raising=False)
        for k, v in variant.items():
            if k.startswith("TMX_"):
                monkeypatch.setenv(k, v)
        e = Engine(g, n_instances=ninst)
        try:
            e.upload_state(0, states)
            if ntr:
                e.upload_tracers(0, tracers)
            for step in (1, 2, 3):
                e.step(scheme, 200.0, first=(step == 1))
                if variant.get("probe"):
                    for ix in range(1, ninst):
                        e.download_state(ix)
                    e.upload_state(ninst - 1, e.download_state(ninst - 1))
            e.sync()
            runs.append([(e.download_state(ix), e.download_tracers(ix) if ntr else None) for ix in range(ninst)])
        finally:
            e.close()
    work = {"ars343": {2}}.get(scheme, set(range(1, ninst)))
    for other in runs[1:]:
        for ix in range(ninst):
            if ix != 0 and ix in work:
                continue
            assert max(gu.prognostic_errors(runs[0][ix][0], other[ix][0])) <= EXACT, (scheme, ix)
            if ntr:
                assert max(gu.tracer_errors(runs[0][ix][1], other[ix][1])) <= EXACT, (scheme, ix)
