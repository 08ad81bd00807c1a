"""GPU parity tests (run with -m gpu on an MI355X): the HIP engine, called through the C ABI,
against the C oracle on identical inputs and against the golden vectors of the real reference."""
import os
import numpy as np
import pytest
import golden_util as gu

pytestmark = pytest.mark.gpu

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


@pytest.fixture(scope="module", params=["jw", "schar"])
def percall(request):
    """jw: baroclinic wave; schar: Schar mountain on the reduced-radius sphere with a Rayleigh layer (row H7)."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    if request.param == "jw":
        d = gu.load("percall_ne3_L5_p6.npz")
        g, _ = gu.grid_from_fixture(d, override=True)
    else:
        d = gu.load("percall_schar_ne3_L6_p6.npz")
        g, _ = gu.grid_from_fixture(d, override=True, case="schar")
    e = Engine(g)
    o = Oracle(g)
    yield d, g, e, o
    e.close()


def _cmp(e, o, ie, io, tol, what):
    e.sync()
    errs = gu.prognostic_errors(e.download_state(ie), o.get_state(io))
    assert max(errs) <= tol, (what, errs)
    return errs


def test_percall_chain_vs_oracle_and_reference(percall):
    d, g, e, o = percall
    npatch = len(g.patches)
    dt = float(d["cfg/dt"][0]); sdt = dt * gu.ARS343_GAMMA
    start = gu.full_states(d, "warm", npatch)
    o.set_state(0, start); e.upload_state(0, start)
    # upload/download round trip is exact on the prognostic slots
    assert max(gu.prognostic_errors(e.download_state(0), start)) == 0.0
    o.copy_data(0, 1); e.copy_data(0, 1)
    o.h_step_explicit(0, 1, sdt); e.h_step_explicit(0, 1, sdt)
    _cmp(e, o, 1, 1, TOL_CALL, "h_explicit")
    assert max(gu.prognostic_errors(e.download_state(1), gu.full_states(d, "h_explicit", npatch))) == 0.0
    o.v_step_explicit(0, 1, sdt); e.v_step_explicit(0, 1, sdt)
    _cmp(e, o, 1, 1, TOL_CALL, "v_explicit")
    assert max(gu.prognostic_errors(e.download_state(1), gu.full_states(d, "v_explicit", npatch))) == 0.0
    o.apply_dss(1); e.apply_dss(1)
    _cmp(e, o, 1, 1, TOL_CALL, "dss")
    assert max(gu.prognostic_errors(e.download_state(1), gu.full_states(d, "dss", npatch))) == 0.0
    o.copy_data(1, 2); e.copy_data(1, 2)
    assert o.v_step_implicit(2, 2, sdt) == 0
    e.v_step_implicit(2, 2, sdt)
    _cmp(e, o, 2, 2, EXACT, "v_implicit")
    assert max(gu.prognostic_errors(e.download_state(2), gu.full_states(d, "v_implicit", npatch))) == 0.0
    o.h_step_after_subcycle(2, 3, 4, dt); e.h_step_after_subcycle(2, 3, 4, dt)
    _cmp(e, o, 4, 4, EXACT, "hypervis work")
    _cmp(e, o, 3, 3, EXACT, "hypervis")
    assert max(gu.prognostic_errors(e.download_state(3), gu.full_states(d, "hypervis", npatch))) == 0.0
    cf = [0.25, -0.5, 1.75, 0.5, 0, 0, 0]
    o.linear_combine_data(cf, 3); e.linear_combine_data(cf, 3)
    _cmp(e, o, 3, 3, EXACT, "lincomb")


def test_download_fills_derived_slots(percall):
    """W on levels and U,V on interfaces as the reference leaves them after the H prologue."""
    d, g, e, o = percall
    npatch = len(g.patches)
    e.upload_state(5, gu.full_states(d, "warm", npatch))
    got = e.download_state(5)
    ref = gu.full_states(d, "h_explicit_in_after", npatch)
    for (gn, ge), (rn, re_) in zip(got, ref):
        assert np.max(np.abs(gn[3, 1:-1, 1:-1] - rn[3, 1:-1, 1:-1])) <= 1e-15 * np.max(np.abs(rn[3]))
        for c in (0, 1):
            assert np.max(np.abs(ge[c, 1:-1, 1:-1] - re_[c, 1:-1, 1:-1])) <= 1e-15 * np.max(np.abs(re_[c]))


def test_error_behaviour(percall):
    from tempestmodel_amd.engine import TempestError
    d, g, e, o = percall
    with pytest.raises(TempestError):
        e.h_step_explicit(1, 1, 1.0)          # reference: "initial and update must be distinct"
    with pytest.raises(TempestError):
        e.h_step_after_subcycle(1, 2, 1, 1.0)  # initial == working
    with pytest.raises(TempestError):
        e.copy_data(0, 99)
    with pytest.raises(TempestError):
        e.linear_combine_data([1.0], 3)
    with pytest.raises(TempestError):
        e.h_substep_after_subcycle(0, 1, 2, 1.0, 2)   # reference: "Invalid iSubStep"
    with pytest.raises(TempestError):
        e.interp_tracers((None, 1, 1), 0)              # no plan; the reference: "Unable to Interpolate with no tracers."


def test_error_behaviour_reference_state_and_interpolation():
    """Uniform diffusion without the reference state, interpolation requests the plan cannot serve."""
    from tempestmodel_amd.engine import Engine, TempestError
    d = gu.load("interp_ne4_L6_p6.npz")
    g, st, tr, pts = gu.interp_case(d)
    e = Engine(g)
    try:
        e.upload_state(0, st); e.upload_tracers(0, tr)
        plan = e.interp_create({k: v for k, v in pts.items() if k != "rll_from_abp"})
        with pytest.raises(TempestError):
            e.interp_state(plan, 0, 0, True, True, 6.37122e6)      # primitive conversion needs the RLL matrices
        with pytest.raises(TempestError):
            e.interp_state(plan, 0, 0, False, False, 6.37122e6)    # reference state never provided
        with pytest.raises(TempestError):
            e.interp_state(plan, 0, 3, True, False, 6.37122e6)
        assert np.isfinite(e.interp_state(plan, 0, 0, True, False, 6.37122e6)).all()
        e.interp_destroy(plan)
        bad = dict(pts); bad["node_a"] = pts["node_a"] + 1          # not the first node of an element
        with pytest.raises(TempestError):
            e.interp_create(bad)
    finally:
        e.close()


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


def test_tracer_percall_chain():
    """SURVEY rows H8 / V8 on the device: every tracer operation against the C oracle and the reference's own output
    (fixture percall_tracers_ne3_L6_p6.npz: JW state + two closed-form tracer densities, one with compact support so
    that both FilterNegativeTracers variants act)."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    d = gu.load("percall_tracers_ne3_L6_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=True, ntracers=2)
    dt = float(d["cfg/dt"][0]); sdt = dt * gu.ARS343_GAMMA
    e = Engine(g); o = Oracle(g)
    try:
        st, tr = gu.full_states(d, "warm", 6), gu.full_tracers(d, "warm", 6)
        o.set_state(0, st); o.set_tracers(0, tr); e.upload_state(0, st); e.upload_tracers(0, tr)
        assert max(gu.tracer_errors(e.download_tracers(0), tr)) == 0.0

        def cmp(ie, io, tag, tol):
            e.sync()
            got = e.download_tracers(ie)
            assert max(gu.tracer_errors(got, o.get_tracers(io))) == 0.0, tag
            assert max(gu.tracer_errors(got, gu.full_tracers(d, tag, 6))) == 0.0, tag
            assert max(gu.prognostic_errors(e.download_state(ie), o.get_state(io))) == 0.0, tag

        o.copy_data(0, 1); e.copy_data(0, 1)
        o.h_step_explicit(0, 1, sdt); e.h_step_explicit(0, 1, sdt); cmp(1, 1, "h_explicit", TOL_CALL)
        o.v_step_explicit(0, 1, sdt); e.v_step_explicit(0, 1, sdt); cmp(1, 1, "v_explicit", TOL_CALL)
        o.apply_dss(1); e.apply_dss(1); cmp(1, 1, "dss", TOL_CALL)
        o.copy_data(1, 2); e.copy_data(1, 2)
        assert o.v_step_implicit(2, 2, sdt) == 0
        e.v_step_implicit(2, 2, sdt); cmp(2, 2, "v_implicit", 1e-11)
        o.h_step_after_subcycle(2, 3, 4, dt); e.h_step_after_subcycle(2, 3, 4, dt)
        cmp(4, 4, "hypervis_work", 1e-11); cmp(3, 3, "hypervis", 1e-11)
        cf = [0.25, -0.5, 1.75, 0.5, 0, 0, 0]
        o.linear_combine_data(cf, 3); e.linear_combine_data(cf, 3); cmp(3, 3, "lincomb", 1e-11)
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


UDIFF = (1500.0, 500.0)


def test_uniform_diffusion_explicit_vertical_percall():
    """BASELINE config 4 (supercell) dynamics on the device: uniform diffusion in H.StepExplicit and the fully explicit
    V.StepExplicit with two tracers, call by call against the C oracle and the reference's own output."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    d = gu.load("percall_udiff_ne3_L6_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=True, case="smallplanet", ntracers=2)
    dt = float(d["cfg/dt"][0]); sdt = dt * gu.ARS343_GAMMA
    e = Engine(g, fully_explicit=True, uniform_diffusion=UDIFF); o = Oracle(g, fully_explicit=True, uniform_diffusion=UDIFF)
    try:
        st, tr = gu.full_states(d, "warm", 6), gu.full_tracers(d, "warm", 6)
        o.set_state(0, st); o.set_tracers(0, tr); e.upload_state(0, st); e.upload_tracers(0, tr)

        def cmp(ie, io, tag):
            e.sync()
            gs, gt = e.download_state(ie), e.download_tracers(ie)
            assert max(gu.prognostic_errors(gs, o.get_state(io))) == 0.0, tag
            assert max(gu.tracer_errors(gt, o.get_tracers(io))) == 0.0, tag
            assert max(gu.prognostic_errors(gs, gu.full_states(d, tag, 6))) == 0.0, tag
            assert max(gu.tracer_errors(gt, gu.full_tracers(d, tag, 6))) == 0.0, tag

        o.copy_data(0, 1); e.copy_data(0, 1)
        o.h_step_explicit(0, 1, sdt); e.h_step_explicit(0, 1, sdt); cmp(1, 1, "h_explicit")
        o.v_step_explicit(0, 1, sdt); e.v_step_explicit(0, 1, sdt); cmp(1, 1, "v_explicit")
        o.apply_dss(1); e.apply_dss(1); cmp(1, 1, "dss")
        o.copy_data(1, 2); e.copy_data(1, 2)
        assert o.v_step_implicit(2, 2, sdt) == 0
        e.v_step_implicit(2, 2, sdt); cmp(2, 2, "v_implicit")
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


def test_level_parallel_explicit_tracer_update_is_bit_identical_to_the_column_kernel(monkeypatch):
    """UpdateColumnTracers in the fully explicit mode: the level-parallel kernel (default) and the one-lane-per-column kernel
    (TMX_VT_COLUMN=1) give identical doubles, with and without uniform diffusion -- and so do the sliding-window form (the default:
    a thread walks a segment of its column) at every segment count, the LDS-tiled form it superseded and the archived ones."""
    from tempestmodel_amd.engine import Engine
    d = gu.load("steps_udiff_ars343_ne3_L6_p6.npz")
    g, _ = gu.make_grid(3, 6, 6, case="smallplanet", ntracers=2)
    st, tr = gu.expand_compact(d, "step1", g), gu.expand_compact_tracers(d, "step1", g)
    for ud in (UDIFF, None):
        out = []
        # LDS-tiled, one lane per column, untiled, by column segments of 4 / 8 / 10 levels (six levels here: a ragged last segment, a lone ragged one)
        for variant in (None, "TMX_VT_COLUMN", "TMX_VT_EXPLICIT_V1", "TMX_VT_WALK=0", "TMX_VT_WALK=4", "TMX_VT_WALK=5", "TMX_VT_WALK=8", "TMX_VT_WALK=10",
                        "TMX_VT_WALK=-1", "TMX_VT_WALK=-2", "TMX_VT_WALK=-3", "TMX_VT_WALK=-6"):      # (negative: a sliding window over each of n column segments)
            for name in ("TMX_VT_COLUMN", "TMX_VT_EXPLICIT_V1", "TMX_VT_WALK"):
                monkeypatch.delenv(name, raising=False)
            if variant:
                monkeypatch.setenv(variant.split("=")[0], variant.split("=")[1] if "=" in variant else "1")
            # (the untiled form and the register-held segments are archived kernels: experiments flavour of the library)
            archived = variant == "TMX_VT_EXPLICIT_V1" or (variant or "").startswith("TMX_VT_WALK=") and int(variant.split("=")[1]) > 0
            e = Engine(g, fully_explicit=True, uniform_diffusion=ud, flavour="experiments" if archived else None)
            try:
                e.upload_state(0, st); e.upload_tracers(0, tr)
                e.copy_data(0, 1)
                e.v_step_explicit(0, 1, 0.7)
                e.sync()
                out.append((e.download_state(1), e.download_tracers(1)))
            finally:
                e.close()
        for other in out[1:]:
            assert max(gu.tracer_errors(out[0][1], other[1])) <= EXACT
            assert max(gu.prognostic_errors(out[0][0], other[0])) <= EXACT
        assert any(np.abs(a - b).max() > 0 for a, b in zip(out[0][1], tr))        # the update did something


def test_row_parallel_implicit_tracer_update_is_bit_identical_to_the_column_kernel(monkeypatch):
    """UpdateColumnTracers in the implicit mode: the kernel with four row lanes per column (default) and the one-lane-per-column
    kernel (TMX_VT_ROWS=0) give identical doubles over whole ARS343 steps, and so does ARK232's all-columns form."""
    from tempestmodel_amd.engine import Engine
    d = gu.load("steps_tracers_ne3_L6_p6.npz")
    g, _ = gu.make_grid(3, 6, 6, ntracers=2)
    st, tr = gu.expand_compact(d, "step1", g), gu.expand_compact_tracers(d, "step1", g)
    for scheme, ninst in (("ars343", None), ("ark232", 8)):
        out = []
        for rows in (True, False):
            if rows:
                monkeypatch.delenv("TMX_VT_ROWS", raising=False)
            else:
                monkeypatch.setenv("TMX_VT_ROWS", "0")
            e = Engine(g) if ninst is None else Engine(g, n_instances=ninst)
            try:
                e.upload_state(0, st); e.upload_tracers(0, tr)
                for step in range(2):
                    e.step(scheme, 200.0, first=(step == 0))
                e.sync()
                out.append((e.download_state(0), e.download_tracers(0)))
            finally:
                e.close()
        assert max(gu.tracer_errors(out[0][1], out[1][1])) <= EXACT, scheme
        assert max(gu.prognostic_errors(out[0][0], out[1][0])) <= EXACT, scheme
        assert any(np.abs(a - b).max() > 0 for a, b in zip(out[0][1], tr))


@pytest.mark.parametrize("L,ud", [(6, True), (6, False), (23, True)])
def test_column_walking_vertical_kernels_are_bit_identical_to_the_level_parallel_ones(L, ud):
    """V.StepExplicit's U,V update, the explicitly evaluated implicit terms and the explicit tracer column update as level-parallel
    kernels (option value 0) and as column walks with sliding register windows (the default; -n: n segments per column, here also
    more segments than fit, ragged last segments and single-level segments): per call and over whole ARS343 / ARK232 steps in the
    fully explicit vertical mode, with and without uniform diffusion -- identical doubles."""
    from tempestmodel_amd.engine import Engine
    g, st = gu.make_grid(3, L, 6, case="smallplanet", ntracers=2)
    rng = np.random.default_rng(5)
    tr = [np.abs(1e-3 * P_node[4][None] * (1.0 + 0.1 * rng.standard_normal((2,) + P_node[4].shape))) for P_node, _ in st]
    for P, (n, e_), t in zip(g.patches, st, tr):
        P.geom["ref_node"] = n.copy(); P.geom["ref_redge"] = e_.copy(); P.geom["ref_tracers"] = 0.9 * t
    # a rough state: the penalty terms and the diffusion stencils see sign changes and large gradients
    rough = []
    for n, e_ in st:
        n = n.copy(); e_ = e_.copy()
        n[0] += rng.uniform(-20.0, 20.0, n[0].shape); n[1] += rng.uniform(-20.0, 20.0, n[1].shape)
        n[2] *= 1.0 + 0.01 * rng.standard_normal(n[2].shape); n[4] *= 1.0 + 0.01 * rng.standard_normal(n[4].shape)
        e_[3] = rng.uniform(-3.0, 3.0, e_[3].shape)
        rough.append((n, e_))
    res = []
    for seg in (0, -1000, -1, -2, -3, -5, -64):
        e = Engine(g, n_instances=8, fully_explicit=True, uniform_diffusion=UDIFF if ud else None,
                   options={"vx_walk": seg, "vite_walk": seg, "vt_explicit_walk": seg})
        try:
            e.upload_state(0, rough); e.upload_tracers(0, tr)
            e.copy_data(0, 1); e.v_step_explicit(0, 1, 0.7)
            e.copy_data(0, 2); e.v_step_implicit_terms_explicitly(0, 2, 0.7)
            for k in range(2):
                e.step("ars343", 0.5, first=(k == 0))
            e.step("ark232", 0.5)
            e.sync()
            res.append([(e.download_state(ix), e.download_tracers(ix)) for ix in (0, 1, 2)])
        finally:
            e.close()
    for other in res[1:]:
        for (sa, ta), (sb, tb) in zip(res[0], other):
            assert max(gu.prognostic_errors(sa, sb)) == 0.0 and max(gu.tracer_errors(ta, tb)) == 0.0


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


def test_substep_after_subcycle_equals_step_after_subcycle():
    """HorizontalDynamicsFEM::SubStepAfterSubCycle (the two halves of the hyperviscosity step, DSS done by the caller)
    reproduces StepAfterSubCycle bit for bit, with and without tracers."""
    from tempestmodel_amd.engine import Engine
    for ntr in (0, 2):
        g, st = gu.make_grid(3, 6, 6, ntracers=ntr)
        e = Engine(g)
        try:
            e.upload_state(0, st)
            if ntr:
                e.upload_tracers(0, [g.initial_tracers[p] for p in range(6)])
            e.step_ars343(200.0)                      # W != 0
            e.h_step_after_subcycle(0, 1, 2, 200.0)
            assert e.h_substep_after_subcycle(0, 3, 4, 200.0, 0) == 4
            e.apply_dss(4)
            assert e.h_substep_after_subcycle(0, 3, 4, 200.0, 1) == 3
            e.apply_dss(3)
            e.sync()
            for a, b in ((1, 3), (2, 4)):
                for (na, ea), (nb, eb) in zip(e.download_state(a), e.download_state(b)):
                    assert np.array_equal(na[[0, 1, 2, 4]], nb[[0, 1, 2, 4]]) and np.array_equal(ea[3], eb[3])
                if ntr:
                    for ta, tb in zip(e.download_tracers(a), e.download_tracers(b)):
                        assert np.array_equal(ta, tb)
        finally:
            e.close()


def test_output_interpolation():
    """Grid::ReduceInterpolate on the device (tmx_interp_state / tmx_interp_tracers) against the C
    oracle and the reference's own output, every flag combination."""
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    d = gu.load("interp_ne4_L6_p6.npz")
    g, st, tr, pts = gu.interp_case(d)
    a = float(d["interp/earth_radius"][0])
    o = Oracle(g); o.set_state(0, st); o.set_tracers(0, tr)
    e = Engine(g)
    try:
        e.upload_state(0, st); e.upload_tracers(0, tr)
        e.set_reference_state()
        plan = e.interp_create(pts)
        for inc in (0, 1):
            for prim in (0, 1):
                got = e.interp_state(plan, 0, 0, bool(inc), bool(prim), a)
                assert gu.interp_rel(got, o.interpolate_state(0, pts, 0, bool(inc), bool(prim), a)) < 1e-14, (inc, prim)
                assert gu.interp_rel(got, d["interp/state_ref%d_prim%d" % (inc, prim)]) < 1e-14, (inc, prim)
        for only in (1, 2):
            got = e.interp_state(plan, 0, only, True, False, a)
            assert gu.interp_rel(got, o.interpolate_state(0, pts, only, True, False, a)) < 1e-14
        assert gu.interp_rel(e.interp_state(plan, 0, 1, True, False, a), d["interp/state_nodes_only"]) < 1e-14
        assert gu.interp_rel(e.interp_tracers(plan, 0), d["interp/tracers"]) < 1e-14
        e.interp_destroy(plan)
        late = e.interp_create(pts)          # a plan that outlives its engine is released without touching the engine
    finally:
        e.close()
    e.interp_destroy(late)


def test_restart_image_pack_and_unpack():
    """SURVEY 8(f) item 4, second half: the DataContainer image OutputManagerComposite writes per patch, transposed on the
    device.  Interior nodes of the image hold exactly what tmx_download_state / tmx_download_tracers deliver (prognostic and
    derived arrays, the reference's values of test_download_fills_derived_slots), halo ring zero, patch index in front;
    unpacking the image into another instance reproduces the instance bit for bit, and a run continued from it is identical."""
    from tempestmodel_amd.engine import Engine, TempestError
    d = gu.load("steps_tracers_ne3_L6_p6.npz")
    g, _ = gu.make_grid(3, 6, 6, ntracers=2)
    st, tr = gu.expand_compact(d, "step1", g), gu.expand_compact_tracers(d, "step1", g)
    e = Engine(g)
    try:
        e.upload_state(0, st); e.upload_tracers(0, tr)
        e.step_ars343(200.0); e.sync()
        want_s, want_t = e.download_state(0), e.download_tracers(0)
        L = g.L
        images = []
        for P, (wn, we), wt in zip(g.patches, want_s, want_t):
            img = e.pack_active_state(P.index, 0)
            images.append(img)
            nn = P.na * P.nb
            assert img.size == 8 * (1 + 5 * nn * L + 5 * nn * (L + 1) + 2 * nn * L)
            assert int(img[:4].view(np.int32)[0]) == P.index and not img[4:8].any()
            body = img[8:].view(np.float64)
            node = body[:5 * nn * L].reshape(5, P.na, P.nb, L)
            redge = body[5 * nn * L:5 * nn * (2 * L + 1)].reshape(5, P.na, P.nb, L + 1)
            trc = body[5 * nn * (2 * L + 1):].reshape(2, P.na, P.nb, L)
            assert np.array_equal(node[:, 1:-1, 1:-1], wn[:, 1:-1, 1:-1])
            assert np.array_equal(redge[[0, 1, 3]][:, 1:-1, 1:-1], we[[0, 1, 3]][:, 1:-1, 1:-1])
            assert np.array_equal(trc[:, 1:-1, 1:-1], wt[:, 1:-1, 1:-1])
            for a in (node, redge, trc):       # halo ring
                assert not a[:, 0].any() and not a[:, -1].any() and not a[:, :, 0].any() and not a[:, :, -1].any()
        for P, img in zip(g.patches, images):
            e.unpack_active_state(P.index, 3, img)
        got_s, got_t = e.download_state(3), e.download_tracers(3)
        for (an, ae), (bn, be) in zip(got_s, want_s):
            assert np.array_equal(an[:, 1:-1, 1:-1], bn[:, 1:-1, 1:-1]) and np.array_equal(ae[[0, 1, 3]][:, 1:-1, 1:-1], be[[0, 1, 3]][:, 1:-1, 1:-1])
        assert all(np.array_equal(a[:, 1:-1, 1:-1], b[:, 1:-1, 1:-1]) for a, b in zip(got_t, want_t))
        # continue from the image: a second engine that only ever saw the image
        e2 = Engine(g)
        try:
            for P, img in zip(g.patches, images):
                e2.unpack_active_state(P.index, 0, img)
            e.step_ars343(200.0); e2.step_ars343(200.0); e.sync(); e2.sync()
            assert max(gu.prognostic_errors(e2.download_state(0), e.download_state(0))) <= EXACT
            assert max(gu.tracer_errors(e2.download_tracers(0), e.download_tracers(0))) <= EXACT
        finally:
            e2.close()
        with pytest.raises(TempestError, match="image is of patch"):
            e.unpack_active_state(1, 0, images[0])
        with pytest.raises(TempestError, match="bytes given"):
            e.unpack_active_state(0, 0, images[0][:-8])
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


def test_hundred_steps_ne15_L30():
    """North-star gate at BASELINE config 2 (ne15 np4 L30, dt 300 s): fp64 state difference < 1e-10 after 100 ARS343
    steps, every variable.  Where the real reference binary travelled with the snapshot (oracle/_ref/tmx_ref_dump,
    built in the build container) it runs here, on the GPU box's host, and hands over ITS OWN inputs (geometry,
    operators, halo transforms, state after 2 steps): the device then reproduces the reference's state after 100
    further steps bit for bit.  Without the binary the run is checked against the C oracle on the synthesised grid."""
    import os, subprocess, tempfile
    import tmxd
    from tempestmodel_amd.engine import Engine
    from oracle_lib import Oracle
    ne, L, dt = 15, 30, 300.0
    # the reference with the netlib LAPACK of OpenBLAS: its band LU is evaluated with FMA kernels on every host, whereas
    # MKL 2021.4 takes a non-FMA level-2 path on the AMD host of the GPU box (tools/host_lapack_probe.sh; on the Intel
    # build host the two builds are bit-identical, tests/golden/lapack_spread.json)
    ref_bin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "tmx_ref_dump_lapack2")
    ref_end = None
    if os.path.exists(ref_bin):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "ref.tmxd")
            env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
            subprocess.run([ref_bin, "--mode", "steps", "--ne", str(ne), "--levels", str(L), "--dt", str(dt), "--warm", "2",
                            "--steps", "100", "--every", "100", "--geom", "1", "--out", out], env=env, check=True,
                           stdout=subprocess.DEVNULL, timeout=1500)
            rd = tmxd.read(out)
        g, _ = gu.grid_from_fixture(rd, override=True)
        start = gu.full_states(rd, "warm", 6)
        ref_end = gu.full_states(rd, "step100", 6)
        del rd
    else:
        g, states = gu.make_grid(ne, L, 6)
        o0 = Oracle(g); o0.set_state(0, states)
        for _ in range(2):
            assert o0.step_ars343(dt) == 0
        start = o0.get_state(0)
    o = Oracle(g); o.set_state(0, start)
    e = Engine(g); e.upload_state(0, start)
    try:
        for _ in range(100):
            e.step_ars343(dt)
        e.sync()
        got = e.download_state(0)
        for _ in range(100):
            assert o.step_ars343(dt) == 0
        errs_o = gu.prognostic_errors(got, o.get_state(0))
        print("ne15 L30, 100 steps vs C oracle:", errs_o)
        assert max(errs_o) <= EXACT, errs_o
        if ref_end is not None:
            errs_r = gu.prognostic_errors(got, ref_end)
            print("ne15 L30, 100 steps vs real reference (its own inputs):", errs_r)
            assert max(errs_r) < TOL_STEPS, errs_r          # the gate
            assert max(errs_r) <= EXACT, errs_r               # and in fact bit for bit
    finally:
        e.close()


def test_bench_workload_vs_live_reference():
    """The workload bench.py times (BASELINE config 3's grid: ne30 np4 L30, 24 patches, dt 150 s) at its FULL size against
    the real reference run live on the GPU box's host: the reference hands over its own geometry, operators, halo
    transforms and its state after 2 steps; three ARS343 steps later the device state equals the reference's bit for bit.
    Needs oracle/_ref (built in the build container; it travels with the snapshot)."""
    import os, subprocess, tempfile
    import tmxd
    from tempestmodel_amd.engine import Engine
    ref_bin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "tmx_ref_dump_lapack2")
    gu.need_live_reference(ref_bin)
    ne, L, dt, npatch = 30, 30, 150.0, 24
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "ref.tmxd")
        env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
        subprocess.run([ref_bin, "--mode", "steps", "--ne", str(ne), "--levels", str(L), "--dt", str(dt), "--patches", str(npatch),
                        "--warm", "2", "--steps", "3", "--every", "3", "--geom", "1", "--out", out], env=env, check=True,
                       stdout=subprocess.DEVNULL, timeout=1500)
        rd = tmxd.read(out)
    g, _ = gu.grid_from_fixture(rd, override=True)
    assert len(g.patches) == npatch
    start = gu.full_states(rd, "warm", npatch)
    ref_end = gu.full_states(rd, "step3", npatch)
    del rd
    e = Engine(g)
    try:
        e.upload_state(0, start)
        for _ in range(3):
            e.step_ars343(dt)
        e.sync()
        errs = gu.prognostic_errors(e.download_state(0), ref_end)
        print("ne30 L30 24 patches, 3 steps vs the live reference:", errs)
        assert max(errs) <= EXACT, errs
    finally:
        e.close()


LIVE_CASES = [
    # name, dump-driver arguments, synthesiser case, tracers, engine options, stepper, dt, steps
    ("jw_tracers_p24", ["--ne", "8", "--levels", "12", "--tracers", "2"], "jw", 2, {}, "ars343", 200.0, 3),
    ("jw_ark232_oddL_p24", ["--ne", "6", "--levels", "9", "--scheme", "ark232", "--tracers", "1"], "jw", 1, {}, "ark232", 200.0, 3),
    ("jw_ars443_p24", ["--ne", "10", "--levels", "16", "--scheme", "ars443"], "jw", 0, {}, "ars443", 150.0, 2),
    ("jw_ars232_visc2_p24", ["--ne", "6", "--levels", "10", "--scheme", "ars232", "--hypervis", "2"], "jw", 0,
     {"nu": (2.0e5, 2.0e5, 2.0e5), "hypervis_order": 2}, "ars232", 200.0, 3),
    ("smallplanet_udiff_p24", ["--ne", "6", "--levels", "8", "--planet", "120", "--udiff", "1500", "500", "--explicitvertical", "--tracers", "2"],
     "smallplanet", 2, {"fully_explicit": True, "uniform_diffusion": (1500.0, 500.0)}, "ars343", 1.0, 3),
    ("jw_p96", ["--ne", "8", "--levels", "6", "--patches", "96"], "jw", 0, {}, "ars343", 200.0, 2),
    ("jw_tracers_L60_p24", ["--ne", "4", "--levels", "60", "--tracers", "3"], "jw", 3, {}, "ars343", 200.0, 2),
    ("jw_ars232_tracers_oddL_p24", ["--ne", "6", "--levels", "7", "--scheme", "ars232", "--tracers", "2"], "jw", 2, {}, "ars232", 150.0, 2),
    # Schar mountain on the reduced-radius sphere: topography (terrain-following metric with non-zero slopes) and a Rayleigh layer
    ("schar_p24", ["--case", "schar", "--ne", "6", "--levels", "10"], "schar", 0, {}, "ars343", 0.5, 3),
    ("schar_ars222_p6", ["--case", "schar", "--ne", "4", "--levels", "8", "--patches", "6", "--scheme", "ars222"], "schar", 0, {}, "ars222", 0.5, 3),
]


@pytest.mark.parametrize("name,args,case,ntr,opts,scheme,dt,nsteps", LIVE_CASES, ids=[c[0] for c in LIVE_CASES])
def test_live_reference_sweep(name, args, case, ntr, opts, scheme, dt, nsteps):
    """Configurations no committed fixture holds, against the real reference run live on the GPU box's host (its own geometry,
    operators and halo transforms; state after 2 steps): sub-panel patch layouts (24 and 96 patches) with tracers, odd level
    counts, the other steppers, order-2 viscosity, the supercell dynamics.  Identical doubles after 2-3 further steps.
    Needs oracle/_ref."""
    import os, subprocess, tempfile
    import tmxd
    from tempestmodel_amd.engine import Engine
    ref_bin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "tmx_ref_dump_lapack2")
    gu.need_live_reference(ref_bin)
    npatch = int(args[args.index("--patches") + 1]) if "--patches" in args else 24
    full = list(args) + ([] if "--patches" in args else ["--patches", "24"])
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "ref.tmxd")
        env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
        subprocess.run([ref_bin, "--mode", "steps", "--dt", str(dt), "--warm", "2", "--steps", str(nsteps), "--every", str(nsteps),
                        "--geom", "1", "--out", out] + full, env=env, check=True, stdout=subprocess.DEVNULL, timeout=1500)
        rd = tmxd.read(out)
    g, _ = gu.grid_from_fixture(rd, override=True, case=case, ntracers=ntr)
    assert len(g.patches) == npatch
    for P in g.patches:
        for nm in ("ref_node", "ref_redge", "ref_tracers"):
            if "p%d/%s" % (P.index, nm) in rd:
                P.geom[nm] = rd["p%d/%s" % (P.index, nm)]
    e = Engine(g, n_instances=10, **opts)
    try:
        e.upload_state(0, gu.full_states(rd, "warm", npatch))
        if ntr:
            e.upload_tracers(0, gu.full_tracers(rd, "warm", npatch))
        for _ in range(nsteps):
            e.step(scheme, dt)
        e.sync()
        errs = gu.prognostic_errors(e.download_state(0), gu.full_states(rd, "step%d" % nsteps, npatch))
        terr = gu.tracer_errors(e.download_tracers(0), gu.full_tracers(rd, "step%d" % nsteps, npatch)) if ntr else [0.0]
        print(name, "vs the live reference:", errs, terr)
        assert max(errs) <= EXACT and max(terr) <= EXACT, (errs, terr)
    finally:
        e.close()


def test_fused_column_kernels_are_bit_identical_to_split_kernels():
    """k_vi_fused (one wavefront per 64 columns), k_vi_pair (assembly and elimination in two wavefronts), k_vi_group (one
    column per 16-lane group; what small per-GPU grids run) and k_vi_assemble + k_vi_solve (TMX_VI_MODE=split) perform the
    same arithmetic: results are bit-identical.  The pair kernel also with two pairs per workgroup on an ODD number of
    column groups (60 columns per wavefront -> 17 groups: the last workgroup's second pair has no columns of its own and must
    still keep the barrier rhythm) and with ragged last groups."""
    import os
    from tempestmodel_amd.engine import Engine
    d = gu.load("steps_ne4_L6_p24.npz")
    g, _ = gu.make_grid(4, 6, 24)
    start = gu.expand_compact(d, "step2", g)
    out = []
    for env in ({"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "0"}, {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1"}, {"TMX_VI_GROUP": "1"},
                {"TMX_VI_MODE": "split"}, {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "0", "TMX_VI_SPARSE": "0"},
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_PAIR_WG": "2", "TMX_VI_CPW": "60"},
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_PAIR_WG": "1", "TMX_VI_CPW": "37"},
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_SPLIT_BACK": "1"},
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_SPLIT_BACK": "1", "TMX_VI_BACK_SUB": "2", "TMX_VI_CPW": "60"},
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_SPLIT_BACK": "1", "TMX_VI_BACK_SUB": "4", "TMX_VI_PAIR_WG": "1"},
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_STAGGER_NS": "5000"}):
        os.environ.update(env)
        try:
            e = Engine(g, flavour="experiments")      # (the one-wavefront, split and separate-back-substitution kernels live there)
            e.upload_state(1, start)
            e.copy_data(1, 2)
            e.v_step_implicit(1, 2, 87.0)
            e.sync()
            out.append(e.download_state(2))
            e.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    for other in out[1:]:
        for (an, ae), (bn, be) in zip(out[0], other):
            assert np.array_equal(an[[0, 1, 2, 4]], bn[[0, 1, 2, 4]])
            assert np.array_equal(ae[3], be[3])


def test_column_solve_with_lane_divergent_pivots():
    """Neighbouring columns of a smooth state choose the same pivot rows, so the production column kernel normally runs its
    wave-uniform tails only.  Here every column gets its own random scaling of rho*theta, W and rho (x 0.5 .. 2 per level,
    W up to +-30 m/s): the 64 columns of a wavefront now pivot differently (asserted through the kernel's own statistics)
    and the kernel takes its lane-divergent tails.  The result must still be the C oracle's dgbsv restatement bit for bit,
    and equal to the one-wavefront kernel, whose per-lane select path is an independent implementation of the interchange."""
    import os
    from oracle_lib import Oracle
    from tempestmodel_amd.engine import Engine
    d = gu.load("steps_ne4_L6_p24.npz")
    g, _ = gu.make_grid(4, 6, 24)
    start = gu.expand_compact(d, "step2", g)
    rng = np.random.default_rng(7)
    rough = []
    for node, edge in start:
        node = node.copy(); edge = edge.copy()
        node[2] *= rng.uniform(0.5, 2.0, node[2].shape)
        node[4] *= rng.uniform(0.5, 2.0, node[4].shape)
        edge[3] = rng.uniform(-30.0, 30.0, edge[3].shape)
        rough.append((node, edge))
    o = Oracle(g); o.set_state(1, rough); o.set_state(2, rough)
    assert o.v_step_implicit(1, 2, 87.0) == 0
    want = o.get_state(2)
    out = []
    for env in ({"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1"}, {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "0"},
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_PAIR_WG": "2", "TMX_VI_CPW": "60"}):
        os.environ.update(env)
        try:
            e = Engine(g, flavour="experiments" if env["TMX_VI_PAIR"] == "0" else None)
            e.upload_state(1, rough)
            e.copy_data(1, 2)
            if env["TMX_VI_PAIR"] == "1":
                e.pivot_stats(True)
            e.v_step_implicit(1, 2, 87.0)
            e.sync()
            if env["TMX_VI_PAIR"] == "1":
                uni, tot = e.pivot_stats(False)
                print("pivot steps on the wave-uniform path: %d of %d" % (uni, tot))
                assert 0 < tot and uni < tot, (uni, tot)         # some steps did take the lane-divergent tails
            out.append(e.download_state(2))
            e.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    for got in out:
        errs = gu.prognostic_errors(got, want)
        assert max(errs) <= EXACT, errs
        # the duplicated seam nodes inside a patch as well (the solve scatters its result to them)
        for (an, ae), (bn, be) in zip(got, out[0]):
            assert np.array_equal(an[[0, 1, 2, 4]], bn[[0, 1, 2, 4]])
            assert np.array_equal(ae[3], be[3])


def test_closed_form_metric_is_bit_identical_to_stored_arrays():
    """tmx_set_patch_metric_factors: the kernels' in-register evaluation of the terrain-following metric
    (GridPatchCSGLL.cpp:441-566) is accepted only after a bit-for-bit comparison with the stored arrays, so
    whole steps and the per-call kernels agree bit for bit with the engine streaming the arrays; factors that
    do not reproduce the arrays are refused (stored arrays stay in use)."""
    import os
    from tempestmodel_amd.engine import Engine
    INFO_CLOSED = 6
    d = gu.load("steps_ne4_L6_p24.npz")
    g, _ = gu.make_grid(4, 6, 24)
    start = gu.expand_compact(d, "step2", g)
    out = []
    for mode in ("closed", "stored"):
        os.environ["TMX_METRIC"] = mode
        try:
            e = Engine(g)
        finally:
            os.environ.pop("TMX_METRIC", None)
        assert e.info(INFO_CLOSED) == (1 if mode == "closed" else 0)
        e.upload_state(0, start)
        res = []
        e.copy_data(0, 1); e.h_step_explicit(0, 1, 40.0); e.sync(); res.append(e.download_state(1))
        e.v_step_explicit(0, 1, 40.0); e.sync(); res.append(e.download_state(1))
        e.copy_data(0, 2); e.v_step_implicit(0, 2, 87.0); e.sync(); res.append(e.download_state(2))
        e.copy_data(0, 3); e.v_step_implicit_terms_explicitly(0, 3, 40.0); e.sync(); res.append(e.download_state(3))
        for _ in range(3):
            e.step_ars343(200.0)
        e.sync(); res.append(e.download_state(0))
        out.append(res)
        e.close()
    for ra, rb in zip(*out):
        for (an, ae), (bn, be) in zip(ra, rb):
            assert np.array_equal(an[[0, 1, 2, 4]], bn[[0, 1, 2, 4]])
            assert np.array_equal(ae[3], be[3])
    # perturbed factors must be refused
    keep = g.patches[0].X.copy()
    try:
        g.patches[0].X = keep * (1.0 + 1e-15)
        e = Engine(g)
        assert e.info(INFO_CLOSED) == 0
        e.close()
    finally:
        g.patches[0].X = keep


def _reference_call_sequence(e, dt):
    """TimestepSchemeARS343::Step spelled out with one ABI call per reference call
    (TimestepSchemeARS343.cpp:150-236); the tableau is taken from the oracle's restatement."""
    import ctypes as C
    from oracle_lib import lib as olib
    # tableau via the C oracle is not exported; restate the U-form combos numerically from the ABI-independent formulae
    gm = 0.4358665215084590
    b1 = -1.5 * gm * gm + 4.0 * gm - 0.25; b2 = 1.5 * gm * gm - 5.0 * gm + 1.25
    a42 = a43 = 0.5529291480359398
    a31 = (1.0 - 4.5 * gm + 1.5 * gm * gm) * a42 + (2.75 - 10.5 * gm + 3.75 * gm * gm) * a43 - 3.5 + 13 * gm - 4.5 * gm * gm
    a32 = (-1.0 + 4.5 * gm - 1.5 * gm * gm) * a42 + (-2.75 + 10.5 * gm - 3.75 * gm * gm) * a43 + 4.0 - 12.5 * gm + 4.5 * gm * gm
    a41 = 1.0 - a42 - a43
    I = [[gm, 0, 0, 0], [0.5 * (1.0 - gm), gm, 0, 0], [b1, b2, gm, 0], [b1, b2, gm, 0]]
    E = [[gm, 0, 0, 0], [a31, a32, 0, 0], [a41, a42, a43, 0], [0, b1, b2, gm]]
    u2 = [0.0] * 7; u3 = [0.0] * 7; u4 = [0.0] * 7
    u2[0] = 1.0 - E[1][0] / E[0][0]; u2[1] = E[1][0] / E[0][0] - I[1][0] / I[0][0]; u2[2] = I[1][0] / I[0][0]
    u3[0] = 1.0 - E[2][0] / E[0][0]; u3[1] = E[2][0] / E[0][0] - I[2][0] / I[0][0]; u3[2] = I[2][0] / I[0][0]
    u3[3] = E[2][1] / E[1][1] - I[2][1] / I[1][1]; u3[4] = I[2][1] / I[1][1]
    u37 = -E[2][1] / E[1][1]
    u4[0] = 1.0 - E[3][0] / E[0][0]; u4[1] = E[3][0] / E[0][0] - I[3][0] / I[0][0]; u4[2] = I[3][0] / I[0][0]
    u4[3] = E[3][1] / E[1][1] - I[3][1] / I[1][1]; u4[4] = I[3][1] / I[1][1]
    u4[5] = E[3][2] / E[2][2] - I[3][2] / I[2][2]; u4[6] = I[3][2] / I[2][2]
    u47 = -E[3][1] / E[1][1]; u48 = -E[3][2] / E[2][2]
    for i in range(3):
        u3[i] += u37 * u2[i]
    for i in range(3):
        u4[i] += u47 * u2[i] + u48 * u3[i]
    u4[3] += u48 * u3[3]; u4[4] += u48 * u3[4]
    de = [E[i][i] for i in range(4)]
    e.copy_data(0, 1)
    e.h_step_explicit(0, 1, de[0] * dt); e.v_step_explicit(0, 1, de[0] * dt); e.apply_dss(1)
    e.copy_data(1, 2); e.v_step_implicit(2, 2, gm * dt)
    e.linear_combine_data(u2, 3)
    e.h_step_explicit(2, 3, de[1] * dt); e.v_step_explicit(2, 3, de[1] * dt); e.apply_dss(3)
    e.copy_data(3, 4); e.v_step_implicit(4, 4, gm * dt)
    e.linear_combine_data(u3, 5)
    e.h_step_explicit(4, 5, de[2] * dt); e.v_step_explicit(4, 5, de[2] * dt); e.apply_dss(5)
    e.copy_data(5, 6); e.v_step_implicit(6, 6, gm * dt)
    e.linear_combine_data(u4, 1)
    e.h_step_explicit(6, 1, de[3] * dt); e.v_step_explicit(6, 1, de[3] * dt); e.apply_dss(1)
    e.copy_data(1, 0)
    e.h_step_after_subcycle(1, 0, 2, dt)


def test_production_step_is_bit_identical_to_the_reference_call_sequence():
    """tmx_step_ars343 folds copies and V.StepExplicit into neighbouring kernels; the state it produces must
    equal, bit for bit, the one produced by issuing every reference call separately through the ABI."""
    from tempestmodel_amd.engine import Engine
    d = gu.load("steps_ne4_L6_p24.npz")
    g, _ = gu.make_grid(4, 6, 24)
    start = gu.expand_compact(d, "step1", g)
    a = Engine(g); b = Engine(g)
    try:
        a.upload_state(0, start); b.upload_state(0, start)
        for _ in range(2):
            a.step_ars343(200.0)
            _reference_call_sequence(b, 200.0)
        a.sync(); b.sync()
        for (an, ae), (bn, be) in zip(a.download_state(0), b.download_state(0)):
            assert np.array_equal(an[[0, 1, 2, 4]], bn[[0, 1, 2, 4]])
            assert np.array_equal(ae[3], be[3])
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("n_ranks,ntr", [(2, 0), (3, 0), (8, 0), (4, 2)])
def test_multirank_device_path_loopback(n_ranks, ntr):
    """The N>1 device path on ONE GPU: n engines (ranks 0..n-1 of the same 24-patch grid) in one process,
    transport replaced by device copies in wire order (tmx_debug_dss_loopback).  Two full ARS343 steps
    issued call by call on every 'rank' must reproduce the single-engine result bit for bit."""
    from tempestmodel_amd.engine import Engine
    d = gu.load("steps_ne4_L6_p24.npz")
    g, _ = gu.make_grid(4, 6, 24, ntracers=ntr)
    start = gu.expand_compact(d, "step1", g)
    single = Engine(g)
    ranks = [Engine(g, rank=r, n_ranks=n_ranks) for r in range(n_ranks)]
    if ntr:        # tracer slabs travel in the same exchange
        for e in [single] + ranks:
            e.upload_tracers(0, [g.initial_tracers[p] for p in range(24)])

    class Lockstep:
        """Issues every call on all rank engines; DSS goes through the loopback exchange."""
        def __getattr__(self, name):
            if name == "apply_dss":
                return lambda ix: Engine.dss_loopback(ranks, ix)
            return lambda *a: [getattr(e, name)(*a) for e in ranks]

    try:
        single.upload_state(0, start)
        for e in ranks:
            e.upload_state(0, start)
        ls = Lockstep()
        # StepAfterSubCycle contains its DSS calls, so the lockstep sequence covers two explicit stages,
        # an implicit solve and a linear combination -- every kernel that sees ghost columns.
        gm = gu.ARS343_GAMMA
        dt = 200.0
        for eng in (single, ls):
            eng.copy_data(0, 1)
            eng.h_step_explicit(0, 1, gm * dt); eng.v_step_explicit(0, 1, gm * dt); eng.apply_dss(1)
            eng.copy_data(1, 2); eng.v_step_implicit(2, 2, gm * dt)
            eng.linear_combine_data([0.3, -0.2, 0.9, 0, 0, 0, 0], 3)
            eng.h_step_explicit(2, 3, 0.6 * dt); eng.v_step_explicit(2, 3, 0.6 * dt); eng.apply_dss(3)
        single.sync()
        ref = single.download_state(3)
        for e in ranks:
            e.sync()
            got = e.download_state(3)
            for p in e.local_patches:
                dn = np.max(np.abs(got[p][0][[0, 1, 2, 4]] - ref[p][0][[0, 1, 2, 4]]) / np.max(np.abs(ref[p][0][[0, 1, 2, 4]]), axis=(1, 2, 3), keepdims=True))
                de = np.max(np.abs(got[p][1][3] - ref[p][1][3])) / max(np.max(np.abs(ref[p][1][3])), 1e-300)
                assert dn <= EXACT and de <= EXACT, (e.rank, p, dn, de)
            if ntr:
                tg, tr = e.download_tracers(3), single.download_tracers(3)
                for p in e.local_patches:
                    assert np.array_equal(tg[p][:, 1:-1, 1:-1], tr[p][:, 1:-1, 1:-1]), (e.rank, p)
        assert sum(len(e.local_patches) for e in ranks) == 24
        assert all(e.info(4) > 0 for e in ranks)       # every rank really has ghost columns
    finally:
        single.close()
        for e in ranks:
            e.close()


INFO_EARLY_TILES, INFO_LATE_TILES = 7, 8


@pytest.mark.parametrize("n_ranks,scheme,ntr", [(4, "ars343", 0), (8, "ars343", 0), (3, "strang", 0), (4, "ars343", 2), (3, "ark232", 2)])
def test_multirank_production_steps_loopback(n_ranks, scheme, ntr):
    """The production call (tmx_step: fused program, exchanges inside) on n rank engines, one host thread each, with the
    wire replaced by device copies (tmx_debug_loopback_group): three whole steps are bit-identical to one engine.  The
    stages run boundary tiles first (asserted: early and late tile lists exist), also with tracers, whose flux update,
    column update and hyperviscosity then run over the two tile lists as well."""
    import threading
    from tempestmodel_amd.engine import Engine
    # ne12 on 24 patches (6 x 6 elements each): the smallest layout on which every rank of a 3-, 4- or 8-rank split owns
    # both kinds of tiles -- at ne4 / ne8 every 64-column tile touches a rank boundary and the stages are not split at all
    g, start = gu.make_grid(12, 6, 24, ntracers=ntr)
    tr = [g.initial_tracers[p] for p in range(24)] if ntr else None
    ni = max(7, Engine.scheme_instances(scheme))
    single = Engine(g, n_instances=ni)
    ranks = [Engine(g, rank=r, n_ranks=n_ranks, n_instances=ni) for r in range(n_ranks)]
    try:
        assert all(e.info(INFO_EARLY_TILES) > 0 and e.info(INFO_LATE_TILES) > 0 for e in ranks)
        if ntr:
            for e in [single] + ranks:
                e.upload_tracers(0, tr)
        single.upload_state(0, start)
        for k in range(3):
            single.step(scheme, 200.0, first=(k == 0))
        single.sync()
        ref = single.download_state(0)
        for e in ranks:
            e.upload_state(0, start)
        Engine.loopback_group(ranks)
        errors = []

        def run(e):
            try:
                for k in range(3):
                    e.step(scheme, 200.0, first=(k == 0))
                e.sync()
            except Exception as ex:        # a failing member must not leave the others at the barrier silently
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
            if ntr:
                gott, reft = e.download_tracers(0), single.download_tracers(0)
                for p in e.local_patches:
                    assert np.array_equal(gott[p][:, 1:-1, 1:-1], reft[p][:, 1:-1, 1:-1]), (e.rank, p)
    finally:
        single.close()
        for e in ranks:
            e.close()


def test_multirank_shallow_water_loopback():
    """BASELINE config 1's equation set on 3 rank engines (24 patches, loopback wire): the shallow-water stage runs its
    boundary tiles first as well; five Strang steps are bit-identical to one engine."""
    import threading
    from tempestmodel_amd.cubed_sphere import CubedSphereGrid, ShallowWaterTest2
    from tempestmodel_amd.engine import Engine
    g = CubedSphereGrid(12, 1, 1.0, shallow_water=True, ppd=2)      # ne12: every rank has early and late tiles
    start = g.evaluate_test_case(ShallowWaterTest2())
    single = Engine(g, n_instances=5)
    ranks = [Engine(g, rank=r, n_ranks=3, n_instances=5) for r in range(3)]
    try:
        assert all(e.info(INFO_EARLY_TILES) > 0 and e.info(INFO_LATE_TILES) > 0 for e in ranks)
        for e in [single] + ranks:
            e.upload_state(0, start)
        for k in range(5):
            single.step("strang", 200.0, first=(k == 0))
        single.sync()
        ref = single.download_state(0)
        Engine.loopback_group(ranks)
        errors = []

        def run(e):
            try:
                for k in range(5):
                    e.step("strang", 200.0, first=(k == 0))
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
                assert np.array_equal(got[p][0][[0, 1, 2], 1:-1, 1:-1], ref[p][0][[0, 1, 2], 1:-1, 1:-1]), (e.rank, p)
    finally:
        single.close()
        for e in ranks:
            e.close()


def test_rccl_transport_selftest():
    """Library resolution + communicator + grouped send/recv on the engine's stream (1 rank, to self)."""
    from tempestmodel_amd.engine import Engine
    g, _ = gu.make_grid(2, 4, 6)
    e = Engine(g)
    try:
        e.comm_init(Engine.comm_unique_id())
        e.comm_selftest()
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


def test_shallow_water_24_patches_vs_live_reference():
    """BASELINE config 1's case (Williamson test 2, the stock SWTest2) on 24 patches against the real reference run live on the
    GPU box's host, on the reference's own 2-D geometry: four Strang steps from the analytic initial state (the first with
    fFirstStep), state and carried instance equal the reference's bit for bit.  Needs oracle/_ref."""
    import os, subprocess, tempfile
    import tmxd
    from tempestmodel_amd.cubed_sphere import CubedSphereGrid
    from tempestmodel_amd.engine import Engine
    ref_bin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "tmx_ref_dump_lapack2")
    gu.need_live_reference(ref_bin)
    ne, npatch, dt, nsteps = 8, 24, 200.0, 4
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "ref.tmxd")
        env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
        subprocess.run([ref_bin, "--mode", "steps", "--case", "sw2", "--scheme", "strang", "--ne", str(ne), "--patches", str(npatch), "--dt", str(dt),
                        "--steps", str(nsteps), "--every", str(nsteps), "--geom", "1", "--out", out], env=env, check=True,
                       stdout=subprocess.DEVNULL, timeout=600)
        rd = tmxd.read(out)
    from tempestmodel_amd.cubed_sphere import ShallowWaterTest2
    g = CubedSphereGrid(ne, 1, 1.0, ppd=2, shallow_water=True)
    g.evaluate_test_case(ShallowWaterTest2())          # geometry of the synthesiser, replaced by the reference's below
    for P in g.patches:
        for nm in list(P.geom):
            if "p%d/%s" % (P.index, nm) in rd and np.shape(P.geom[nm]) == rd["p%d/%s" % (P.index, nm)].shape:
                P.geom[nm] = rd["p%d/%s" % (P.index, nm)]
        P.delta_a = float(rd["p%d/element_delta_a" % P.index][0]); P.delta_b = float(rd["p%d/element_delta_b" % P.index][0])
    g.dx = rd["op/dx_basis_1d"]; g.stiff = rd["op/stiffness_1d"]
    gu.override_halo_transforms(g, rd)

    def full(tag):
        out = []
        for P in g.patches:
            n = np.zeros((5, P.na, P.nb, 1)); e_ = np.zeros((5, P.na, P.nb, 2))
            n[[0, 1, 2]] = rd["state/%s/p%d/node" % (tag, P.index)]
            out.append((n, e_))
        return out
    e = Engine(g, n_instances=5)
    try:
        e.upload_state(0, full("initial"))
        for n in range(nsteps):
            e.step("strang", dt, first=(n == 0))
        e.sync()
        for inst, tag in ((0, "step%d" % nsteps), (1, "step%d_carry" % nsteps)):
            errs = gu.prognostic_errors(e.download_state(inst), full(tag))[:3]
            print("shallow water ne8 24 patches, instance %d vs the live reference:" % inst, errs)
            assert max(errs) <= EXACT, errs
    finally:
        e.close()


@pytest.mark.parametrize("scheme", ["ars343", "strang"])
def test_multirank_supercell_configuration_loopback(scheme):
    """The supercell configuration (uniform diffusion relative to the per-rank reference state, fully explicit vertical
    mode, two tracers) on 4 rank engines through the loopback wire: three production steps bit-identical to one engine."""
    import threading
    from tempestmodel_amd.engine import Engine
    g, start = gu.make_grid(12, 6, 24, case="smallplanet", ntracers=2)      # ne12: every rank has early and late tiles
    tr = [g.initial_tracers[p] for p in range(24)]
    kw = dict(fully_explicit=True, uniform_diffusion=UDIFF)
    single = Engine(g, **kw)
    ranks = [Engine(g, rank=r, n_ranks=4, **kw) for r in range(4)]
    try:
        # boundary-first stages are on for this configuration too (tracers, uniform diffusion, explicit vertical mode)
        assert all(e.info(INFO_EARLY_TILES) > 0 and e.info(INFO_LATE_TILES) > 0 for e in ranks)
        for e in [single] + ranks:
            e.upload_state(0, start); e.upload_tracers(0, tr)
        for k in range(3):
            single.step(scheme, 1.0, first=(k == 0))
        single.sync()
        ref, reft = single.download_state(0), single.download_tracers(0)
        Engine.loopback_group(ranks)
        errors = []

        def run(e):
            try:
                for k in range(3):
                    e.step(scheme, 1.0, first=(k == 0))
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
            got, gott = e.download_state(0), e.download_tracers(0)
            for p in e.local_patches:
                assert np.array_equal(got[p][0][[0, 1, 2, 4], 1:-1, 1:-1], ref[p][0][[0, 1, 2, 4], 1:-1, 1:-1]), (e.rank, p)
                assert np.array_equal(got[p][1][3, 1:-1, 1:-1], ref[p][1][3, 1:-1, 1:-1]), (e.rank, p)
                assert np.array_equal(gott[p][:, 1:-1, 1:-1], reft[p][:, 1:-1, 1:-1]), (e.rank, p)
    finally:
        single.close()
        for e in ranks:
            e.close()


# ---- BASELINE config 4: the stock DCMIP2016 supercell with Kessler microphysics -------------------------------

def _supercell_case():
    d = gu.load("supercell_ne2_L40_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=True, case="supercell")      # the reference's own (flat) geometry
    for P in g.patches:
        for nm in ("ref_node", "ref_redge", "ref_tracers"):
            P.geom[nm] = d["p%d/%s" % (P.index, nm)]
    return d, g


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


def test_supercell_kessler_24_patches_vs_live_reference():
    """BASELINE config 4's case on 24 patches (ne4 np4 L40: the stock SupercellTest state from the Fortran initialiser,
    --explicitvertical, uniform diffusion, three tracers, KesslerPhysics::Perform after every step) against the real reference
    run live on the GPU box's host (about a minute of host time: the initialiser is slow): three steps from its state after one
    step, state and tracers bit for bit.  Needs oracle/_ref."""
    import os, subprocess, tempfile
    import tmxd
    from tempestmodel_amd.engine import Engine
    ref_bin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "tmx_ref_dump_lapack2")
    gu.need_live_reference(ref_bin)
    ne, L, npatch, dt, nsteps = 4, 40, 24, 0.5, 3
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "ref.tmxd")
        env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
        subprocess.run([ref_bin, "--case", "supercell", "--explicitvertical", "--physics", "kessler", "--mode", "steps", "--ne", str(ne),
                        "--levels", str(L), "--dt", str(dt), "--patches", str(npatch), "--warm", "1", "--steps", str(nsteps),
                        "--every", str(nsteps), "--geom", "1", "--out", out], env=env, check=True, stdout=subprocess.DEVNULL, timeout=1500)
        rd = tmxd.read(out)
    g, _ = gu.grid_from_fixture(rd, override=True, case="supercell")
    for P in g.patches:
        for nm in ("ref_node", "ref_redge", "ref_tracers"):
            P.geom[nm] = rd["p%d/%s" % (P.index, nm)]
    e = Engine(g, fully_explicit=True, uniform_diffusion=(1500.0, 500.0), nu=(0.0, 0.0, 0.0))
    try:
        e.set_level_heights()
        e.upload_state(0, gu.full_states(rd, "warm", npatch)); e.upload_tracers(0, gu.full_tracers(rd, "warm", npatch))
        for _ in range(nsteps):
            e.step_ars343(dt)
            e.kessler(0, dt)
        e.sync()
        errs = gu.prognostic_errors(e.download_state(0), gu.full_states(rd, "step%d" % nsteps, npatch))
        terr = gu.tracer_errors(e.download_tracers(0), gu.full_tracers(rd, "step%d" % nsteps, npatch))
        print("supercell + Kessler ne4 L40 24 patches, %d steps vs the live reference:" % nsteps, errs, terr)
        assert max(errs) <= EXACT and max(terr) <= EXACT, (errs, terr)
    finally:
        e.close()


# ---- BASELINE config 5 shape: the stock Held-Suarez test at L60 ------------------------------------------------

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


def test_held_suarez_L60_24_patches_vs_live_reference():
    """BASELINE config 5's shape on 24 patches (ne16 np4 L60: 1.5 M points) against the real reference run live on the GPU
    box's host: the stock HeldSuarezTest state after 2 steps with forcing, its own geometry; 3 further steps of dynamics +
    HeldSuarezPhysics::Perform on the device equal the reference's bit for bit (about 4.4 M evaluations each of the forcing's
    pow / exp / log on developed data).  Needs oracle/_ref."""
    import os, subprocess, tempfile
    import tmxd
    from tempestmodel_amd.engine import Engine
    ref_bin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "tmx_ref_dump_lapack2")
    gu.need_live_reference(ref_bin)
    ne, L, dt, npatch = 16, 60, 200.0, 24
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "ref.tmxd")
        env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
        subprocess.run([ref_bin, "--case", "heldsuarez", "--physics", "hs", "--mode", "steps", "--ne", str(ne), "--levels", str(L),
                        "--dt", str(dt), "--patches", str(npatch), "--warm", "2", "--steps", "3", "--every", "3", "--geom", "1",
                        "--out", out], env=env, check=True, stdout=subprocess.DEVNULL, timeout=1500)
        rd = tmxd.read(out)
    g, _ = gu.grid_from_fixture(rd, override=True, case="heldsuarez")
    for P in g.patches:
        P.lat = rd["p%d/hs_latitude" % P.index]
    start = [(rd["state/warm/p%d/node" % p], rd["state/warm/p%d/redge" % p]) for p in range(npatch)]
    ref_end = gu.full_states(rd, "step3", npatch)
    del rd
    e = Engine(g)
    try:
        e.set_physics_inputs(None)
        e.upload_state(0, start)
        for _ in range(3):
            e.step_ars343(dt)
            e.held_suarez(0, dt)
        e.sync()
        errs = gu.prognostic_errors(e.download_state(0), ref_end)
        print("Held-Suarez ne16 L60 24 patches, 3 steps + forcing vs the live reference:", errs)
        assert max(errs) <= EXACT, errs
    finally:
        e.close()


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


def test_band_lu_without_fused_multiply_add_flavour():
    """A reference linked to a BLAS without fused multiply-adds (plain netlib, MKL on some AMD hosts) evaluates the band LU's
    updates as multiply + subtract; the oracle restates that with orc_set_lu_fma(0) and the product with its second library
    flavour (libtempest_mi355x_nofma.so, TMX_LU_FMA=0).  The column solve of that flavour equals the oracle's in that mode
    bit for bit -- and differs from the default flavour, so the switch is not a no-op.  Runs in a child process (the
    library is chosen when it is first loaded)."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "tempestmodel_amd", "libtempest_mi355x_nofma.so")):
        pytest.fail("libtempest_mi355x_nofma.so is missing: build() makes it (make -C tempestmodel_amd/csrc nofma)")
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np
        import golden_util as gu
        from oracle_lib import Oracle, lib
        from tempestmodel_amd.engine import Engine
        d = gu.load("steps_ne4_L6_p24.npz")
        g, _ = gu.make_grid(4, 6, 24)
        start = gu.expand_compact(d, "step2", g)
        res = {}
        for fma in (1, 0):
            lib().orc_set_lu_fma(fma)
            o = Oracle(g); o.set_state(1, start); o.set_state(2, start)
            assert o.v_step_implicit(1, 2, 87.0) == 0
            res[fma] = o.get_state(2)
        lib().orc_set_lu_fma(1)
        e = Engine(g)
        e.upload_state(1, start); e.copy_data(1, 2); e.v_step_implicit(1, 2, 87.0); e.sync()
        got = e.download_state(2)
        e.close()
        print("RESULT", max(gu.prognostic_errors(got, res[0])), max(gu.prognostic_errors(got, res[1])))
    """ % (root, root))
    env = dict(os.environ, TMX_LU_FMA="0")
    env.pop("TMX_LIBRARY", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert r.returncode == 0 and line, r.stderr[-2000:]
    err_nofma, err_fma = (float(v) for v in line[0].split()[1:])
    print("multiply + subtract flavour vs oracle without / with fused multiply-adds:", err_nofma, err_fma)
    assert err_nofma <= EXACT and err_fma > EXACT


def test_nofma_flavour_vs_live_mkl_reference():
    """Pins the second library flavour (libtempest_mi355x_nofma.so: band-LU updates as multiply + subtract) to a REAL reference:
    the MKL-linked oracle/_ref/tmx_ref_dump run live on this host.  MKL 2021.4 dispatches a non-FMA level-2 path on the AMD
    hosts of the GPU boxes (tools/host_lapack_probe.sh), which is the host class the flavour exists for.  Whether THIS host is
    one is measured, not assumed: the MKL build and the OpenBLAS-netlib build (FMA kernels on every host) of the reference
    run the same 100 steps at ne4 L10; identical checksums mean MKL uses FMAs here, the default flavour is the matching one
    (test_hundred_steps_ne15_L30 covers it) and this test skips, saying so.  Otherwise BASELINE config 2 (ne15 np4 L30,
    dt 300 s) runs 100 ARS343 steps in the MKL reference, which hands over its own geometry and its state after 2 steps, and
    the nofma flavour must reproduce the reference's final state bit for bit (child process: the library is chosen at load)."""
    import subprocess, sys, tempfile, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mkl_bin = os.path.join(root, "oracle", "_ref", "tmx_ref_dump")
    ob_bin = os.path.join(root, "oracle", "_ref", "tmx_ref_dump_lapack2")
    gu.need_live_reference(mkl_bin, ob_bin)
    if not os.path.exists(os.path.join(root, "tempestmodel_amd", "libtempest_mi355x_nofma.so")):
        pytest.fail("libtempest_mi355x_nofma.so is missing: build() makes it (make -C tempestmodel_amd/csrc nofma)")
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    sums = []
    for b in (mkl_bin, ob_bin):
        out = subprocess.run([b, "--mode", "steps", "--ne", "4", "--levels", "10", "--dt", "200", "--warm", "2", "--steps", "100",
                              "--every", "100", "--geom", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             universal_newlines=True, timeout=900).stdout
        sums.append([l for l in out.splitlines() if "checksum step100" in l])
    cpu = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")
    print("host CPU:", cpu, "| MKL reference == OpenBLAS-netlib reference after 100 steps at ne4 L10:", sums[0] == sums[1])
    assert sums[0] and sums[1], "the reference binaries printed no checksum"
    if sums[0] == sums[1]:
        pytest.skip("MKL evaluates the band LU with fused multiply-adds on this host (%s): the default flavour is the matching one "
                    "(test_hundred_steps_ne15_L30); the multiply + subtract flavour has no live reference here" % cpu)
    with tempfile.TemporaryDirectory() as td:
        ref = os.path.join(td, "ref.tmxd")
        subprocess.run([mkl_bin, "--mode", "steps", "--ne", "15", "--levels", "30", "--dt", "300", "--warm", "2", "--steps", "100",
                        "--every", "100", "--geom", "1", "--out", ref], env=env, check=True, stdout=subprocess.DEVNULL, timeout=1500)
        code = textwrap.dedent("""
            import os, sys
            sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
            import golden_util as gu, tmxd
            from tempestmodel_amd.engine import Engine, LIB_PATH
            rd = tmxd.read(%r)
            g, _ = gu.grid_from_fixture(rd, override=True)
            start = gu.full_states(rd, "warm", 6); end = gu.full_states(rd, "step100", 6)
            e = Engine(g); e.upload_state(0, start)
            for _ in range(100):
                e.step_ars343(300.0)
            e.sync()
            got = e.download_state(0); e.close()
            print("RESULT", " ".join("%%.3e" %% v for v in gu.prognostic_errors(got, end)))
        """ % (root, root, ref))
        res = {}
        for flavour, fma in (("nofma", "0"), ("default", "1")):
            cenv = dict(os.environ, TMX_LU_FMA=fma)
            cenv.pop("TMX_LIBRARY", None)
            r = subprocess.run([sys.executable, "-c", code], env=cenv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            assert r.returncode == 0 and line, r.stderr[-2000:]
            res[flavour] = [float(v) for v in line[0].split()[1:]]
    print("ne15 L30, 100 steps against the live MKL reference: nofma flavour", res["nofma"], "default flavour", res["default"])
    # (the default flavour against THIS reference shows the reference's own LAPACK-to-LAPACK spread, 1.9e-10 in W after 100 steps
    #  on the EPYC hosts -- tests/golden/lapack_spread.json -- which is why the flavour exists; printed above, not asserted)
    assert max(res["nofma"]) <= EXACT, res               # the multiply + subtract flavour IS this host's reference, bit for bit
    assert max(res["default"]) > EXACT, res              # and the switch is not a no-op on such a host


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


def test_config4_full_size_properties(monkeypatch):
    """BASELINE config 4's shape at FULL size (ne30 np4 L40 on 24 patches, planet / 120, uniform diffusion, fully explicit
    vertical dynamics, three tracers, Kessler microphysics after every step; the moist column of tools/config4_timing.py,
    because the stock Fortran initialiser needs > 25 min of host time at this size).  No fixture can hold this size, so the
    checks are the size-independent ones: two steps stay finite; 8 rank engines through the loopback wire (boundary-first
    stages on) end bit-identical to one engine; the hipGraph replay (TMX_GRAPH=1) equals the queued launches."""
    from tempestmodel_amd.engine import Engine
    ne, L, dt = 30, 40, 0.5
    g, states = gu.make_grid(ne, L, 24, ztop=20000.0, case="supercell")
    tracers = []
    for P, (n, e_) in zip(g.patches, states):
        z = P.geom["z_levels"]; rho = n[4]
        qv = 0.014 * np.exp(-z / 2500.0); qc = 2.0e-4 * np.exp(-((z - 3000.0) / 1500.0) ** 2); qr = 1.0e-4 * np.exp(-((z - 2000.0) / 1500.0) ** 2)
        tracers.append(np.stack([rho * qv, rho * qc, rho * qr]))
        P.geom["ref_node"] = n.copy(); P.geom["ref_redge"] = e_.copy(); P.geom["ref_tracers"] = tracers[-1].copy()
    kw = dict(fully_explicit=True, uniform_diffusion=(1500.0, 500.0), nu=(0.0, 0.0, 0.0))

    def step(e, k):
        e.step_ars343(dt); e.kessler(0, dt)

    def run_single():
        e = Engine(g, **kw)
        try:
            e.set_level_heights()
            e.upload_state(0, states); e.upload_tracers(0, tracers)
            for k in range(2):
                step(e, k)
            e.sync()
            return e.download_state(0), e.download_tracers(0)
        finally:
            e.close()

    ref, reft = run_single()
    assert all(np.isfinite(n).all() and np.isfinite(e_[3]).all() for n, e_ in ref) and all(np.isfinite(t).all() for t in reft)
    assert any(np.max(np.abs(t[2])) > 0.0 for t in reft)                       # rain is there: the microphysics ran its branches
    monkeypatch.setenv("TMX_GRAPH", "1")
    got, gott = run_single()
    monkeypatch.delenv("TMX_GRAPH")
    for p in range(24):
        assert np.array_equal(got[p][0], ref[p][0]) and np.array_equal(got[p][1], ref[p][1]) and np.array_equal(gott[p], reft[p]), p
    ranks = [Engine(g, rank=r, n_ranks=8, **kw) for r in range(8)]
    try:
        assert all(e.info(INFO_EARLY_TILES) > 0 and e.info(INFO_LATE_TILES) > 0 for e in ranks)
        for e in ranks:
            e.set_level_heights()
            e.upload_state(0, states); e.upload_tracers(0, tracers)
        Engine.loopback_group(ranks)
        _rank_engines_step(ranks, step, 2)
        Engine.loopback_dissolve(ranks[0])
        for e in ranks:
            got, gott = e.download_state(0), e.download_tracers(0)
            for p in e.local_patches:
                assert np.array_equal(got[p][0][[0, 1, 2, 4], 1:-1, 1:-1], ref[p][0][[0, 1, 2, 4], 1:-1, 1:-1]), (e.rank, p)
                assert np.array_equal(got[p][1][3, 1:-1, 1:-1], ref[p][1][3, 1:-1, 1:-1]), (e.rank, p)
                assert np.array_equal(gott[p][:, 1:-1, 1:-1], reft[p][:, 1:-1, 1:-1]), (e.rank, p)
    finally:
        for e in ranks:
            e.close()


def test_config5_full_size_properties(monkeypatch):
    """BASELINE config 5 at FULL size (Held-Suarez, ne120 np4 L60 on 24 patches: 86 400 elements, 414.7 M DOF, ~35 GiB of HBM on
    one GPU), forcing after every step.  Size-independent checks: two steps stay finite, the hipGraph replay equals the queued
    launches, and 8 rank engines through the loopback wire end bit-identical to one engine.  The state is synthesised once
    (about a minute of host time) and shared; TMX_TEST_SKIP_C5=1 skips the test on a box without the memory or the time."""
    from tempestmodel_amd.cubed_sphere import CubedSphereGrid
    from tempestmodel_amd.engine import Engine
    if os.environ.get("TMX_TEST_SKIP_C5") == "1":
        pytest.skip("TMX_TEST_SKIP_C5=1")
    ne, L, dt = 120, 60, 37.5
    g, states = gu.make_grid(ne, L, 24, case="heldsuarez")

    def step(e, k):
        e.step_ars343(dt); e.held_suarez(0, dt)

    def run_single():
        e = Engine(g)
        try:
            e.set_physics_inputs(None)
            e.upload_state(0, states)
            for k in range(2):
                step(e, k)
            e.sync()
            return e.download_state(0)
        finally:
            e.close()

    ref = run_single()
    assert all(np.isfinite(n).all() and np.isfinite(e_[3]).all() for n, e_ in ref)
    monkeypatch.setenv("TMX_GRAPH", "1")
    got = run_single()
    monkeypatch.delenv("TMX_GRAPH")
    for p in range(24):
        assert np.array_equal(got[p][0], ref[p][0]) and np.array_equal(got[p][1], ref[p][1]), p
    del got
    ranks = [Engine(g, rank=r, n_ranks=8) for r in range(8)]
    try:
        assert all(e.info(INFO_EARLY_TILES) > 0 and e.info(INFO_LATE_TILES) > 0 for e in ranks)
        for e in ranks:
            e.set_physics_inputs(None)
            e.upload_state(0, states)
        Engine.loopback_group(ranks)
        _rank_engines_step(ranks, step, 2)
        Engine.loopback_dissolve(ranks[0])
        for e in ranks:
            got = e.download_state(0)
            for p in e.local_patches:
                assert np.array_equal(got[p][0][[0, 1, 2, 4], 1:-1, 1:-1], ref[p][0][[0, 1, 2, 4], 1:-1, 1:-1]), (e.rank, p)
                assert np.array_equal(got[p][1][3, 1:-1, 1:-1], ref[p][1][3, 1:-1, 1:-1]), (e.rank, p)
    finally:
        for e in ranks:
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



@pytest.mark.parametrize("scheme", ["ars343", "ars232", "ars222", "ars443", "ark232", "strang"])
def test_stage_copies_left_out_and_fused_uv_update_are_bit_identical(monkeypatch, scheme):
    """Fully explicit vertical mode (BASELINE config 4's): the CopyData in front of the V.StepImplicit that does nothing is not
    performed -- the copy reads its source's slot (TMX_SHARE_COPIES=0: every copy made); TMX_VX_FUSED=1 runs V.StepExplicit's
    U,V update inside the kernel of the explicitly treated terms instead of its own launch.  Same doubles, in instance 0
    and in every other instance a caller can ask for between steps (the work instance of StepAfterSubCycle aside)."""
    from tempestmodel_amd.engine import Engine
    d = gu.load("steps_udiff_ars343_ne3_L6_p6.npz")
    g, _ = gu.make_grid(3, 6, 6, case="smallplanet", ntracers=2)
    st, tr = gu.expand_compact(d, "initial", g), gu.expand_compact_tracers(d, "initial", g)
    ninst = Engine.scheme_instances(scheme)
    runs = []
    for variant in ({}, {"TMX_SHARE_COPIES": "0"}, {"TMX_VX_FUSED": "1"}, {"probe": True}):
        monkeypatch.delenv("TMX_SHARE_COPIES", raising=False)
        monkeypatch.delenv("TMX_VX_FUSED", raising=False)
        for k, v in variant.items():
            if k.startswith("TMX_"):
                monkeypatch.setenv(k, v)
        e = Engine(g, n_instances=ninst, fully_explicit=True, uniform_diffusion=UDIFF, flavour="experiments" if "TMX_VX_FUSED" in variant else None)
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
