"""The column kernels against one another, bit for bit: two-wavefront / one-wavefront / lane-group column solve, lane-divergent pivots, the tracer column kernels, the column walks of the fully explicit vertical mode."""
import os
import numpy as np
import pytest
import golden_util as gu
from parity_common import EXACT, TOL_CALL, TOL_STEPS, _cmp, UDIFF, _supercell_case, _rank_engines_step, INFO_EARLY_TILES, INFO_LATE_TILES      # noqa: F401

pytestmark = pytest.mark.gpu


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
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_STAGGER_NS": "5000"},
                # one / two assembly wavefronts per column group (round 5: the second producer evaluates every other block row without the carry)
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_PAIR_WG": "1", "TMX_VI_PRODUCERS": "1"},
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_PAIR_WG": "1", "TMX_VI_PRODUCERS": "2"},
                # ring of two / three block rows (two is what grids of more than ~38 levels get: the LDS then holds two workgroups per CU)
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_PAIR_WG": "2", "TMX_VI_RING_DEPTH": "2"},
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_PAIR_WG": "1", "TMX_VI_PRODUCERS": "1", "TMX_VI_RING_DEPTH": "2"},
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_PAIR_WG": "2", "TMX_VI_RING_DEPTH": "3"}):
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
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_PRODUCERS": "1"}, {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_PRODUCERS": "2"},
                {"TMX_VI_GROUP": "0", "TMX_VI_PAIR": "1", "TMX_VI_PAIR_WG": "2", "TMX_VI_RING_DEPTH": "2"},
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
