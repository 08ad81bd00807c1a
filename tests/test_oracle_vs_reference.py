"""The C oracle (oracle/tmx_oracle.c) against golden vectors dumped from the real reference.

Tolerances are relative to the max abs value of each variable.  H/V explicit reproduce the reference
bit for bit on identical inputs -- including the implicit solve: the reference's band LU is LAPACK's dgbsv on top of
an FMA BLAS (MKL and OpenBLAS agree bit for bit), and the oracle's restatement fuses the same three update statements."""
import json
import os
import numpy as np
import pytest
import golden_util as gu
from oracle_lib import Oracle, dgbsv

ZERO = gu.exact_tolerance()      # 0.0 where the host's libm is the one the fixtures were made with (golden_util.exact_tolerance)


@pytest.fixture(scope="module", params=["jw", "schar"])
def percall(request):
    """jw: baroclinic wave (ne3 L5); schar: Schar mountain on the reduced-radius sphere with a Rayleigh layer
    (ne3 L6; SURVEY row H7 -- StepAfterSubCycle ends with ApplyRayleighFriction there)."""
    if request.param == "jw":
        d = gu.load("percall_ne3_L5_p6.npz")
        g, _ = gu.grid_from_fixture(d, override=True)
    else:
        d = gu.load("percall_schar_ne3_L6_p6.npz")
        g, _ = gu.grid_from_fixture(d, override=True, case="schar")
        assert g.has_rayleigh_friction and float(np.max(g.patches[0].geom["rayleigh_node"])) > 0.0
    return d, g


def test_percall_chain(percall):
    d, g = percall
    npatch = len(g.patches)
    dt = float(d["cfg/dt"][0]); sdt = dt * gu.ARS343_GAMMA
    o = Oracle(g)
    o.set_state(0, gu.full_states(d, "warm", npatch))
    o.copy_data(0, 1)
    o.h_step_explicit(0, 1, sdt)
    # the prologue mutates the input instance (W->levels, U,V->interfaces)
    for (gn, ge), (rn, re_) in zip(o.get_state(0), gu.full_states(d, "h_explicit_in_after", npatch)):
        assert np.array_equal(gn[:, 1:-1, 1:-1], rn[:, 1:-1, 1:-1])
        assert np.array_equal(ge[:, 1:-1, 1:-1], re_[:, 1:-1, 1:-1])
    assert max(gu.prognostic_errors(o.get_state(1), gu.full_states(d, "h_explicit", npatch))) <= ZERO
    o.v_step_explicit(0, 1, sdt)
    assert max(gu.prognostic_errors(o.get_state(1), gu.full_states(d, "v_explicit", npatch))) <= ZERO
    o.apply_dss(1)
    assert max(gu.prognostic_errors(o.get_state(1), gu.full_states(d, "dss", npatch))) <= ZERO
    o.copy_data(1, 2)
    assert o.v_step_implicit(2, 2, sdt) == 0
    assert max(gu.prognostic_errors(o.get_state(2), gu.full_states(d, "v_implicit", npatch))) <= ZERO
    o.h_step_after_subcycle(2, 3, 4, dt)
    assert max(gu.prognostic_errors(o.get_state(4), gu.full_states(d, "hypervis_work", npatch))) <= ZERO
    assert max(gu.prognostic_errors(o.get_state(3), gu.full_states(d, "hypervis", npatch))) <= ZERO
    o.linear_combine_data([0.25, -0.5, 1.75, 0.5, 0, 0, 0], 3)
    assert max(gu.prognostic_errors(o.get_state(3), gu.full_states(d, "lincomb", npatch))) <= ZERO


def test_dss_fills_halo_like_exchange(percall):
    d, g = percall
    npatch = len(g.patches)
    o = Oracle(g)
    o.set_state(1, gu.full_states(d, "v_explicit", npatch))
    o.apply_dss(1)
    ref = gu.full_states(d, "dss", npatch)
    for P, (gn, ge), (rn, re_) in zip(g.patches, o.get_state(1), ref):
        mask = np.ones((P.na, P.nb), dtype=bool)
        for (i, j) in ((0, 0), (0, P.nb - 1), (P.na - 1, 0), (P.na - 1, P.nb - 1)):
            mask[i, j] = False        # cube-corner halo nodes carry no data in the reference
        for c in (0, 1, 2, 4):
            s = np.max(np.abs(rn[c]))
            assert np.max(np.abs(gn[c] - rn[c])[mask]) < 1e-14 * s
        s = np.max(np.abs(re_[3]))
        assert np.max(np.abs(ge[3] - re_[3])[mask]) <= 1e-14 * s


@pytest.mark.parametrize("fixture,ne,L,npatch,dt,tags", [
    ("steps_ne4_L6_p24.npz", 4, 6, 24, 200.0, ["step1", "step2", "step3"]),
    ("steps_ne3_L5_p6.npz", 3, 5, 6, 300.0, ["step1", "step2", "step5", "step10"]),
    ("steps_schar_ne3_L6_p6.npz", 3, 6, 6, 0.5, ["step1", "step2", "step3"]),
])
def test_ars343_steps_from_developed_state(fixture, ne, L, npatch, dt, tags):
    """Full IMEX steps.  Started from the reference's state after step 1: the very first step from the
    analytically balanced initial state is excluded because sign(xi_dot ~ rounding noise) makes the
    reference itself discontinuous there ('first-step sensitivity')."""
    d = gu.load(fixture)
    g, _ = gu.make_grid(ne, L, npatch, case="schar" if "schar" in fixture else "jw")
    o = Oracle(g)
    o.set_state(0, gu.expand_compact(d, tags[0], g))
    step = int(tags[0][4:])
    for tag in tags[1:]:
        while step < int(tag[4:]):
            assert o.step_ars343(dt) == 0
            step += 1
        errs = gu.prognostic_errors(o.get_state(0), gu.expand_compact(d, tag, g))
        assert max(errs) < 1e-10, (tag, errs)
    cs = g.checksum(o.get_state(0))
    ref = d["checksum/" + tags[-1]]
    assert abs(cs[0] - ref[0]) < 1e-11 * abs(ref[0])
    assert abs(cs[4] - ref[4]) < 1e-12 * abs(ref[4])


def test_viscosity_order_2_steps_vs_reference():
    """HorizontalDynamicsFEM with hyperviscosity order 2 (one viscosity pass, nu = 2e5): ARS343 steps vs the reference."""
    d = gu.load("steps_visc2_ne3_L5_p6.npz")
    g, _ = gu.make_grid(3, 5, 6)
    o = Oracle(g, nu=(2.0e5, 2.0e5, 2.0e5), hypervis_order=2)
    o.set_state(0, gu.expand_compact(d, "step1", g))
    for step in (2, 3):
        assert o.step_ars343(300.0) == 0
        assert max(gu.prognostic_errors(o.get_state(0), gu.expand_compact(d, "step%d" % step, g))) < 1e-11


UDIFF = (1500.0, 500.0)      # K_scalar, K_vector of test/dcmip2016/SupercellTest.cpp:107-113


def test_uniform_diffusion_explicit_vertical_percall():
    """BASELINE config 4 without its Fortran physics: uniform diffusion of state minus
    reference state in H.StepExplicit (vector / scalar Laplacians, tracer fluxes) and the fully explicit
    V.StepExplicit (Evaluate, -dt F, explicit tracer update, vertical diffusion of U,V, rho*theta, W, tracers) on the
    reference's own inputs -- bit for bit; V.StepImplicit does nothing."""
    d = gu.load("percall_udiff_ne3_L6_p6.npz")
    assert int(d["cfg/fully_explicit"][0]) == 1 and int(d["cfg/uniform_diffusion"][0]) == 1
    g, _ = gu.grid_from_fixture(d, override=True, case="smallplanet", ntracers=2)
    dt = float(d["cfg/dt"][0]); sdt = dt * gu.ARS343_GAMMA
    o = Oracle(g, fully_explicit=True, uniform_diffusion=UDIFF)
    o.set_state(0, gu.full_states(d, "warm", 6)); o.set_tracers(0, gu.full_tracers(d, "warm", 6))
    o.copy_data(0, 1)
    o.h_step_explicit(0, 1, sdt)
    assert max(gu.prognostic_errors(o.get_state(1), gu.full_states(d, "h_explicit", 6))) <= ZERO
    assert max(gu.tracer_errors(o.get_tracers(1), gu.full_tracers(d, "h_explicit", 6))) <= ZERO
    o.v_step_explicit(0, 1, sdt)
    assert max(gu.prognostic_errors(o.get_state(1), gu.full_states(d, "v_explicit", 6))) <= ZERO
    assert max(gu.tracer_errors(o.get_tracers(1), gu.full_tracers(d, "v_explicit", 6))) <= ZERO
    o.apply_dss(1)
    o.copy_data(1, 2)
    assert o.v_step_implicit(2, 2, sdt) == 0
    assert max(gu.prognostic_errors(o.get_state(2), gu.full_states(d, "v_implicit", 6))) <= ZERO
    # the diffusion terms matter at this size: without them the same call is off by far more than rounding
    o2 = Oracle(g, fully_explicit=True)
    o2.set_state(0, gu.full_states(d, "warm", 6)); o2.set_tracers(0, gu.full_tracers(d, "warm", 6))
    o2.copy_data(0, 1); o2.h_step_explicit(0, 1, sdt); o2.v_step_explicit(0, 1, sdt)
    assert max(gu.prognostic_errors(o2.get_state(1), gu.full_states(d, "v_explicit", 6))) > 1e-8


@pytest.mark.parametrize("scheme", ["ars343", "strang"])
def test_uniform_diffusion_explicit_vertical_steps(scheme):
    """Three whole steps of the supercell configuration (ARS343, and Strang incl. its FilterNegativeTracers after the
    carry-over combination) against the reference."""
    d = gu.load("steps_udiff_%s_ne3_L6_p6.npz" % scheme)
    g, synth = gu.make_grid(3, 6, 6, case="smallplanet", ntracers=2)
    # the host-side synthesiser reproduces the reference's initial state and tracers of this configuration
    assert max(gu.prognostic_errors(synth, gu.expand_compact(d, "initial", g))) < 1e-13
    assert max(gu.tracer_errors([g.initial_tracers[p] for p in range(6)], gu.expand_compact_tracers(d, "initial", g))) < 1e-13
    o = Oracle(g, fully_explicit=True, uniform_diffusion=UDIFF)
    o.set_state(0, gu.expand_compact(d, "initial", g)); o.set_tracers(0, gu.expand_compact_tracers(d, "initial", g))
    for step in (1, 2, 3):
        assert o.step(scheme, 1.0, first=(step == 1)) == 0
        assert max(gu.prognostic_errors(o.get_state(0), gu.expand_compact(d, "step%d" % step, g))) < 1e-12
        assert max(gu.tracer_errors(o.get_tracers(0), gu.expand_compact_tracers(d, "step%d" % step, g))) < 1e-13


def test_tracer_percall_chain():
    """tracer flux divergence + element filter in H.StepExplicit, tracer DSS, the implicit column
    tracer update (tridiagonal dgbtrf/dgbtrs) + column filter in V.StepImplicit, tracer hyperdiffusion + filter in
    StepAfterSubCycle, tracer stage algebra -- each against the reference's own output on identical inputs."""
    d = gu.load("percall_tracers_ne3_L6_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=True, ntracers=2)
    dt = float(d["cfg/dt"][0]); sdt = dt * gu.ARS343_GAMMA
    o = Oracle(g)
    o.set_state(0, gu.full_states(d, "warm", 6)); o.set_tracers(0, gu.full_tracers(d, "warm", 6))
    o.copy_data(0, 1)
    o.h_step_explicit(0, 1, sdt)
    assert max(gu.tracer_errors(o.get_tracers(1), gu.full_tracers(d, "h_explicit", 6))) <= ZERO
    o.v_step_explicit(0, 1, sdt)
    assert max(gu.tracer_errors(o.get_tracers(1), gu.full_tracers(d, "v_explicit", 6))) <= ZERO
    o.apply_dss(1)
    assert max(gu.tracer_errors(o.get_tracers(1), gu.full_tracers(d, "dss", 6))) <= ZERO
    assert max(gu.prognostic_errors(o.get_state(1), gu.full_states(d, "dss", 6))) <= ZERO
    o.copy_data(1, 2)
    assert o.v_step_implicit(2, 2, sdt) == 0
    assert max(gu.tracer_errors(o.get_tracers(2), gu.full_tracers(d, "v_implicit", 6))) <= ZERO
    o.h_step_after_subcycle(2, 3, 4, dt)
    assert max(gu.tracer_errors(o.get_tracers(4), gu.full_tracers(d, "hypervis_work", 6))) <= ZERO
    assert max(gu.tracer_errors(o.get_tracers(3), gu.full_tracers(d, "hypervis", 6))) <= ZERO
    o.linear_combine_data([0.25, -0.5, 1.75, 0.5, 0, 0, 0], 3)
    assert max(gu.tracer_errors(o.get_tracers(3), gu.full_tracers(d, "lincomb", 6))) <= ZERO
    # the filters were exercised: the reference's own state has clipped zeros next to positive values
    t = gu.full_tracers(d, "h_explicit", 6)[0][1, 1:-1, 1:-1]
    assert (t == 0.0).any() and (t > 0.0).any()


def test_tracer_steps_vs_reference():
    """Whole ARS343 steps with two tracers from the reference's state after step 1."""
    d = gu.load("steps_tracers_ne3_L6_p6.npz")
    g, _ = gu.make_grid(3, 6, 6, ntracers=2)
    o = Oracle(g)
    o.set_state(0, gu.expand_compact(d, "step1", g)); o.set_tracers(0, gu.expand_compact_tracers(d, "step1", g))
    for step in (2, 3, 4):
        assert o.step_ars343(200.0) == 0
        assert max(gu.tracer_errors(o.get_tracers(0), gu.expand_compact_tracers(d, "step%d" % step, g))) < 1e-12
        assert max(gu.prognostic_errors(o.get_state(0), gu.expand_compact(d, "step%d" % step, g))) < 1e-10


def test_ark232_with_tracers_vs_reference():
    """SURVEY 8a row V2 with tracers: ARK232's first implicit stage is StepImplicitTermsExplicitly, which ends with the
    implicit column update of the tracers (VerticalDynamicsFEM.cpp:600-608).  Two steps from the reference's state and tracers
    after step 1 against those after step 3, on the reference's own geometry: bit for bit."""
    d = gu.load("steps_ark232_tracers_ne2_L5_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=True, ntracers=2)
    o = Oracle(g, ninst=8)
    o.set_state(0, gu.expand_compact(d, "step1", g)); o.set_tracers(0, gu.expand_compact_tracers(d, "step1", g))
    for _ in range(2):
        assert o.step("ark232", float(d["cfg/dt"][0])) == 0
    assert max(gu.tracer_errors(o.get_tracers(0), gu.expand_compact_tracers(d, "step3", g))) <= ZERO
    assert max(gu.prognostic_errors(o.get_state(0), gu.expand_compact(d, "step3", g))) <= ZERO


def test_output_interpolation_vs_reference():
    """Grid::ReduceInterpolate (the interpolation behind OutputManagerReference) for the state with
    every flag combination and for the tracers -- bit for bit, except (U,V) in primitive form where the probed 2x2
    matrix of CoVecTransRLLFromABP rounds differently (1 ulp)."""
    d = gu.load("interp_ne4_L6_p6.npz")
    g, st, tr, pts = gu.interp_case(d)
    o = Oracle(g)
    o.set_state(0, st); o.set_tracers(0, tr)
    a = float(d["interp/earth_radius"][0])
    for inc in (0, 1):
        for prim in (0, 1):
            got = o.interpolate_state(0, pts, 0, bool(inc), bool(prim), a)
            want = d["interp/state_ref%d_prim%d" % (inc, prim)]
            assert gu.interp_rel(got, want) < (1e-15 if prim else 1e-300), (inc, prim)
    got = o.interpolate_state(0, pts, 1, True, False, a)
    assert gu.interp_rel(got, d["interp/state_nodes_only"]) <= ZERO and np.all(got[3] == 0.0)
    assert gu.interp_rel(o.interpolate_tracers(0, pts), d["interp/tracers"]) <= ZERO
    # the sample includes a GLL node (panel centre) and a point on a panel edge; every value is finite and non-trivial
    assert np.isfinite(d["interp/state_ref1_prim1"]).all() and np.abs(d["interp/state_ref1_prim1"][2]).min() > 0.0


def test_held_suarez_physics_vs_reference():
    """HeldSuarezPhysics::Perform.  Bit for bit on identical inputs (same libm on the CPU)."""
    d = gu.load("physics_hs_ne2_L30_p6.npz")
    g, _ = gu.make_grid(2, 30, 6)
    for P in g.patches:
        P.geom["hs_surface_pressure"] = d["p%d/hs_surface_pressure" % P.index]
        assert np.max(np.abs(P.lat - d["p%d/hs_latitude" % P.index])[1:-1, 1:-1]) < 1e-15
    o = Oracle(g)
    before, after = gu.full_states(d, "hs_before", 6), gu.full_states(d, "hs_after", 6)
    o.set_state(0, before)
    o.held_suarez(0, 200.0)
    assert max(gu.prognostic_errors(o.get_state(0), after)) < 1e-15
    ch = gu.prognostic_errors(before, after)
    assert ch[0] > 1e-5 and ch[2] > 1e-7 and ch[3] == 0.0 and ch[4] == 0.0     # friction and heating both acted


def test_hundred_reference_steps_ne4_L10():
    """North-star gate on the CPU side: the C oracle stays within 1e-10 of the real reference over 100 ARS343
    steps (ne4 L10 dt 200 s, from the reference's state after 2 steps)."""
    d = gu.load("steps100_ne4_L10_p6.npz")
    g, _ = gu.make_grid(4, 10, 6)
    o = Oracle(g)
    o.set_state(0, gu.expand_compact(d, "warm", g))
    for _ in range(100):
        assert o.step_ars343(200.0) == 0
    errs = gu.prognostic_errors(o.get_state(0), gu.expand_compact(d, "step100", g))
    print("oracle vs reference after 100 steps:", errs)
    assert max(errs) < 1e-10, errs


def test_known_answers_are_the_survey_values():
    ka = json.load(open(os.path.join(gu.GOLDEN, "known_answers.json")))["jw_ne4_L10_dt200_ars343"]
    # reference stdout, ne4 L10 ARS343 dt=200s
    assert ka["initial"][0] == 7.536453292760709e+26
    assert ka["step3"][0] == 7.536840678700595e+26
    assert ka["step3"][1] == -2.629408997242810e+20
    assert ka["step3"][3] == 3.205179299367944e+22


def test_dgbsv_against_lapack():
    """orc_dgbsv restates LAPACK dgbsv (partial pivoting); pinned against scipy's LAPACK."""
    from scipy.linalg import lapack
    rng = np.random.default_rng(7)
    for n, kl, ku in ((18, 4, 4), (93, 4, 4), (12, 1, 1), (33, 4, 4)):
        ldab = 2 * kl + ku + 1
        A = np.zeros((n, n))
        for i in range(n):
            for j in range(max(0, i - kl), min(n, i + ku + 1)):
                A[i, j] = rng.standard_normal() * (1e3 if (i + j) % 3 == 0 else 1.0)
        A[np.arange(n), np.arange(n)] *= 1e-2       # weak diagonal: forces row interchanges
        b = rng.standard_normal(n)
        ab = np.zeros((n, ldab))                     # ab[j][kl+ku+i-j]
        for j in range(n):
            for i in range(max(0, j - ku), min(n, j + kl + 1)):
                ab[j, kl + ku + i - j] = A[i, j]
        info, x, ipiv = dgbsv(ab, b, kl, ku)
        assert info == 0
        lub, piv, xr, info2 = lapack.dgbsv(kl, ku, np.asfortranarray(ab.T), b)
        assert info2 == 0
        assert np.array_equal(ipiv, piv)
        assert np.max(np.abs(x - xr)) <= 1e-12 * np.max(np.abs(xr))
        assert np.sum(ipiv != np.arange(n)) > 0


def _scheme_start(d, scheme, g, o_or_e, setter):
    start = gu.expand_compact(d, "%s_step1" % scheme, g)
    setter(0, start)
    if scheme == "strang":
        setter(1, gu.expand_compact(d, "strang_step1_carry", g))
    return start


@pytest.mark.parametrize("scheme", ["ars232", "ars222", "ars443", "ark232", "strang"])
def test_other_time_schemes_vs_reference(scheme):
    """the remaining steppers of the path, two steps from the reference's state after
    its first step (ne2, L4, dt 400 s) against the reference's state after step 3."""
    d = gu.load("schemes_ne2_L4_p6.npz")
    g, _ = gu.make_grid(2, 4, 6)
    o = Oracle(g, ninst=10)
    _scheme_start(d, scheme, g, o, o.set_state)
    for _ in range(2):
        assert o.step(scheme, 400.0) == 0
    errs = gu.prognostic_errors(o.get_state(0), gu.expand_compact(d, "%s_step3" % scheme, g))
    assert max(errs) < 1e-10, (scheme, errs)


STRANG_VARIANTS = [("strang_fe", 0.0, 50.0), ("strang_rk4", 0.0, 400.0), ("strang_ssp3", 0.0, 400.0), ("strang_ssprk53", 0.0, 400.0),
                   ("strang_rk4", 0.25, 400.0), ("strang", 0.5, 400.0)]


@pytest.mark.parametrize("scheme,oc,dt", STRANG_VARIANTS)
def test_strang_explicit_discretisations_vs_reference(scheme, oc, dt):
    """SURVEY 8a row T3, the rest of it: TimestepSchemeStrang with ForwardEuler / RungeKutta4 / RungeKuttaSSP3 /
    RungeKuttaSSPRK53 (TimestepSchemeStrang.cpp:484-648) and with a non-zero off-centering (:651-660): two steps from the
    reference's state (and carried instance 1) after its first step, against the reference's state after step 3."""
    d = gu.load("strang_variants_ne2_L4_p6.npz")
    name = scheme + ("_oc%02d" % round(100 * oc) if oc else "")
    g, _ = gu.grid_from_fixture(d, override=True)          # the reference's own geometry
    o = Oracle(g, ninst=5)
    o.set_state(0, gu.expand_compact(d, "%s_step1" % name, g))
    o.set_state(1, gu.expand_compact(d, "%s_step1_carry" % name, g))
    ref = gu.expand_compact(d, "%s_step3" % name, g)
    assert all(np.isfinite(n).all() for n, _ in ref)
    o.set_strang_off_centering(oc)
    try:
        for _ in range(2):
            assert o.step(scheme, dt) == 0
    finally:
        o.set_strang_off_centering(0.0)
    assert max(gu.prognostic_errors(o.get_state(0), ref)) <= ZERO
    assert max(gu.prognostic_errors(o.get_state(1), gu.expand_compact(d, "%s_step3_carry" % name, g))) <= ZERO


def _sw_setup():
    from tempestmodel_amd.cubed_sphere import CubedSphereGrid, ShallowWaterTest2
    g = CubedSphereGrid(8, 1, 1.0, shallow_water=True)
    states = g.evaluate_test_case(ShallowWaterTest2())
    return g, states


def _sw_ref(d, tag, g):
    out = []
    for P in g.patches:
        n = np.zeros((5, P.na, P.nb, 1)); e = np.zeros((5, P.na, P.nb, 2))
        n[[0, 1, 2], 1:-1, 1:-1] = d["state/%s/p%d/node" % (tag, P.index)]
        out.append((n, e))
    return out


def test_shallow_water_test2_strang_vs_reference():
    """BASELINE config 1: Williamson test 2, ne8, Strang/KGU35, dt 200 s, 10 steps
    from the analytic initial state, against the reference's state dumps and its printed checksums."""
    d = gu.load("sw2_ne8_strang.npz")
    g, states = _sw_setup()
    assert max(gu.prognostic_errors(states, _sw_ref(d, "initial", g))[:3]) < 1e-14
    o = Oracle(g, ninst=5)
    o.set_state(0, states)
    for n in range(10):
        assert o.step("strang", 200.0, first=(n == 0)) == 0
        if n == 0:
            assert max(gu.prognostic_errors(o.get_state(0), _sw_ref(d, "step1", g))[:3]) < 1e-13
    got = o.get_state(0)
    assert max(gu.prognostic_errors(got, _sw_ref(d, "step10", g))[:3]) < 1e-12
    cs = g.checksum(got)
    ka = json.load(open(os.path.join(gu.GOLDEN, "known_answers.json")))["sw2_ne8_dt200_strang"]
    assert ka["initial"][0] == 7.114413185251165e+22 and ka["initial"][2] == 1.205365998142204e+18
    assert ka["step10"][0] == 7.114410762138457e+22 and ka["step10"][2] == 1.205365998142180e+18
    assert abs(cs[0] - ka["step10"][0]) < 1e-12 * abs(ka["step10"][0])
    assert abs(cs[2] - ka["step10"][2]) < 1e-13 * abs(ka["step10"][2])


def test_ten_steps_bit_for_bit_with_reference_inputs():
    """Whole ARS343 steps: with the reference's own geometry, operators and halo transforms (per-call fixture, same
    ne3 L5 grid) the oracle reproduces the reference's state after 1, 2, 5 and 10 steps BIT FOR BIT."""
    dg = gu.load("percall_ne3_L5_p6.npz")
    d = gu.load("steps_ne3_L5_p6.npz")
    g, _ = gu.grid_from_fixture(dg, override=True)
    o = Oracle(g)
    o.set_state(0, gu.expand_compact(d, "step1", g))
    step = 1
    for tag in ("step2", "step5", "step10"):
        while step < int(tag[4:]):
            assert o.step_ars343(300.0) == 0
            step += 1
        assert max(gu.prognostic_errors(o.get_state(0), gu.expand_compact(d, tag, g))) <= ZERO, tag


def test_lapack_spread_of_the_reference_is_zero():
    """tests/golden/lapack_spread.json (tests/analysis/lapack_spread.py): the reference linked with MKL and with the
    netlib LAPACK of OpenBLAS ends 100 ARS343 steps in bit-identical states at ne4 L10 and ne15 L30 (sha256 of the
    prognostic state), so 'which LAPACK' is not a source of spread -- both evaluate dgbtf2 / dgbtrs with fused
    multiply-adds, as the oracle and the device do.  The oracle's remaining distance to those runs is input synthesis
    (cubed_sphere.py vs the reference's grid objects), within the 1e-10 gate for every variable but W at ne15 L30,
    which tests/test_gpu_live_reference.py::test_hundred_steps_ne15_L30 therefore runs on the reference's own inputs."""
    import json, os
    sp = json.load(open(os.path.join(gu.GOLDEN, "lapack_spread.json")))
    for case in ("ne4_L10_dt200", "ne15_L30_dt300"):
        c = sp[case]
        assert c["mkl"]["sha256_prognostic"] == c["openblas_netlib"]["sha256_prognostic"]
        assert max(c["spread_ref_vs_ref"]) == 0.0
        assert max(c["oracle_vs_mkl"][i] for i in (0, 1, 2, 4)) < 1e-10


def test_lu_fma_switch():
    """orc_set_lu_fma(0) = a BLAS without fused multiply-add: the implicit step then differs from the reference dumps
    by rounding (1e-14), which is what pins the fused form as the reference's arithmetic."""
    from oracle_lib import lib
    d = gu.load("percall_ne3_L5_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=True)
    sdt = float(d["cfg/dt"][0]) * gu.ARS343_GAMMA
    out = {}
    try:
        for fma in (1, 0):
            lib().orc_set_lu_fma(fma)
            o = Oracle(g)
            o.set_state(2, gu.full_states(d, "dss", 6))
            assert o.v_step_implicit(2, 2, sdt) == 0
            out[fma] = max(gu.prognostic_errors(o.get_state(2), gu.full_states(d, "v_implicit", 6)))
    finally:
        lib().orc_set_lu_fma(1)
    assert out[1] == 0.0 and 0.0 < out[0] < 1e-12, out


# ---- BASELINE config 4: DCMIP2016 supercell with Kessler microphysics -------------------------------------------

def test_kessler_column_known_answers():
    """orc_kessler_column against SUBROUTINE KESSLER as the reference's build compiles it (amdflang -O3, mixed single /
    double precision): 3 x 96 columns (12, 40, 60 levels; dry, cloudy, raining; dt 0.5 .. 150 s so that the rain
    sub-cycling acts), bit for bit."""
    from oracle_lib import kessler_column
    d = gu.load("kessler_columns.npz")
    split = 0
    for gi in range(3):
        inp, out, dts, precl = d["g%d/in" % gi], d["g%d/out" % gi], d["g%d/dt" % gi], d["g%d/precl" % gi]
        for c in range(inp.shape[1]):
            th, qv, qc, qr, pr = kessler_column(*inp[:6, c], dts[c], inp[6, c])
            for got, ref in zip((th, qv, qc, qr), out[:, c]):
                assert np.array_equal(got, ref), (gi, c)
            assert pr == precl[c]
            split += int(np.any(out[3, c] != inp[3, c]))
    assert split > 100          # rain changed in most columns


def _supercell():
    d = gu.load("supercell_ne2_L40_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=True, case="supercell")      # the reference's own (flat) geometry
    assert g.L == 40 and abs(g.ztop - 20000.0) < 1e-9
    for P in g.patches:
        for nm in ("ref_node", "ref_redge", "ref_tracers"):
            P.geom[nm] = d["p%d/%s" % (P.index, nm)]
        # flat topography: the reference's level heights are one column, and the synthesised ones equal them
        zl = np.broadcast_to(d["p0/kessler_z_levels"][1, 1], P.geom["z_levels"].shape)
        assert np.array_equal(P.geom["z_levels"], zl)
    return d, g


def test_kessler_physics_on_the_supercell_state():
    """KesslerPhysics::Perform on the stock supercell state after 2 steps (nearly a no-op there: no cloud yet) and on a
    moistened copy where condensation, autoconversion, accretion, sedimentation and evaporation all act: bit for bit."""
    d, g = _supercell()
    dt = float(d["cfg/dt"][0])
    for before, after in (("warm", "kessler_after"), ("kessler_moist_before", "kessler_moist_after")):
        o = Oracle(g, fully_explicit=True, uniform_diffusion=UDIFF)
        o.set_state(0, gu.expand_compact(d, before, g)); o.set_tracers(0, gu.expand_compact_tracers(d, before, g))
        prect = [np.zeros((P.na, P.nb)) for P in g.patches]
        o.kessler(0, dt, [P.geom["z_levels"] for P in g.patches], prect)
        want = gu.expand_compact(d, after, g)
        assert max(gu.prognostic_errors(o.get_state(0), want)) <= ZERO, after
        assert max(gu.tracer_errors(o.get_tracers(0), gu.expand_compact_tracers(d, after, g))) <= ZERO, after
    # accumulated precipitation after both calls (UserData2D "PRECT", KesslerPhysics.cpp:234): second call only rains
    got = max(float(np.max(p_)) for p_ in prect)
    assert got > 1e-6
    for P, a in zip(g.patches, prect):
        assert np.array_equal(a[1:-1, 1:-1], d["p%d/kessler_prect" % P.index][0][1:-1, 1:-1])
    changed = gu.prognostic_errors(gu.expand_compact(d, "kessler_moist_after", g), gu.expand_compact(d, "kessler_moist_before", g))
    assert changed[2] > 1e-3 and changed[4] > 1e-6          # rho*theta and rho moved


def test_supercell_steps_with_kessler():
    """The stock supercell end to end: 3 ARS343 steps (uniform diffusion, fully explicit vertical dynamics, nu = 0),
    KesslerPhysics::Perform after every step as Model::Go does (Model.cpp:470-481), from the reference's state after 2
    such steps, on the reference's own grid: bit for bit."""
    d, g = _supercell()
    dt = float(d["cfg/dt"][0])
    assert float(d["cfg/nu"][0]) == 0.0 and int(d["cfg/fully_explicit"][0]) == 1 and int(d["cfg/ntracers"][0]) == 3
    o = Oracle(g, nu=(0.0, 0.0, 0.0), fully_explicit=True, uniform_diffusion=UDIFF)
    o.set_state(0, gu.expand_compact(d, "warm", g)); o.set_tracers(0, gu.expand_compact_tracers(d, "warm", g))
    zl = [P.geom["z_levels"] for P in g.patches]
    for _ in range(3):
        assert o.step_ars343(dt) == 0
        o.kessler(0, dt, zl)
    errs = gu.prognostic_errors(o.get_state(0), gu.expand_compact(d, "step3", g))
    terr = gu.tracer_errors(o.get_tracers(0), gu.expand_compact_tracers(d, "step3", g))
    assert max(errs) <= ZERO, errs
    assert max(terr) <= ZERO, terr


# ---- BASELINE config 5 shape: Held-Suarez at L60 ---------------------------------------------------------------

def test_held_suarez_steps_L60():
    """The stock HeldSuarezTest (ztop 35 km, Rayleigh layer of strength zero, ARS343 dt 200 s) at ne3 L60: 10 steps of
    dynamics + HeldSuarezPhysics::Perform after every step from the reference's state after 2 such steps."""
    d = gu.load("heldsuarez_ne3_L60_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=True, case="heldsuarez")     # the reference's own (flat) geometry
    assert g.L == 60 and abs(g.ztop - 35000.0) < 1e-9
    for P in g.patches:
        assert np.max(np.abs(P.lat - d["p%d/hs_latitude" % P.index])[1:-1, 1:-1]) < 1e-15
        P.lat = d["p%d/hs_latitude" % P.index]          # the reference's own latitudes (the forcing takes sin / cos of them)
    dt = float(d["cfg/dt"][0])
    o = Oracle(g)
    o.set_state(0, gu.expand_compact(d, "warm", g))
    for _ in range(10):
        assert o.step_ars343(dt) == 0
        o.held_suarez(0, dt)
    errs = gu.prognostic_errors(o.get_state(0), gu.expand_compact(d, "step10", g))
    assert max(errs) <= ZERO, errs          # bit for bit
