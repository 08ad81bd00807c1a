"""The host grid / test-case synthesiser against dumps of the reference's own Grid objects."""
import numpy as np
import golden_util as gu


def _relc(a, b):
    s = np.max(np.abs(b))
    return np.max(np.abs(a - b)) / (s if s > 0 else 1.0)


def test_operators_and_gll_matrices():
    d = gu.load("percall_ne3_L5_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=False)
    assert _relc(g.dx, d["op/dx_basis_1d"]) < 1e-15
    assert _relc(g.stiff, d["op/stiffness_1d"]) < 1e-15
    assert _relc(g.gllw, d["op/gll_weights_1d"]) < 1e-15
    for nm, (c, b, e) in g.ops.items():
        assert _relc(c, d["op/%s/coeff" % nm]) < 1e-14, nm
        assert np.array_equal(b, d["op/%s/begin" % nm]), nm
        assert np.array_equal(e, d["op/%s/end" % nm]), nm
    for nm in ("reta_levels", "reta_interfaces"):
        assert np.array_equal(getattr(g, nm), d["grid/" + nm])


def test_metric_terms_per_component():
    d = gu.load("percall_ne3_L5_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=False)
    for P in g.patches:
        pn = "p%d/" % P.index
        assert np.array_equal(P.neighbor_panels, d[pn + "neighbor_panels"])
        for nm, arr in P.geom.items():
            if pn + nm not in d:
                continue
            ref = d[pn + nm]
            if ref.ndim == 4 or (ref.ndim == 3 and ref.shape[-1] == 2):
                for c in range(ref.shape[-1]):
                    assert _relc(arr[..., c], ref[..., c]) < 5e-12, (nm, c)
            else:
                assert _relc(arr, ref) < 5e-14, nm


def test_halo_transforms_match_covec_panel_trans():
    d = gu.load("percall_ne3_L5_p6.npz")
    g, _ = gu.grid_from_fixture(d, override=False)
    checked = 0
    for P in g.patches:
        h = P.halo
        for e in range(4):
            key = "halo_trans/p%d/e%d" % (P.index, e)
            if key not in d:
                continue
            M = d[key]
            for m in range(len(h["i"])):
                i, j = h["i"][m], h["j"][m]
                on = (e == 0 and i == P.na - 1) or (e == 1 and j == P.nb - 1) or (e == 2 and i == 0) or (e == 3 and j == 0)
                if not on or h["src_patch"][m] < 0 or h["src_panel"][m] != d[key + "_srcpanel"][0]:
                    continue
                s = j if e in (0, 2) else i
                assert np.max(np.abs(M[s] - h["trans"][m])) < 1e-14
                checked += 1
    assert checked == 6 * 4 * 12


def test_initial_state_and_checksum():
    d = gu.load("steps_ne4_L6_p24.npz")
    g, states = gu.make_grid(4, 6, 24)
    ref = gu.expand_compact(d, "initial", g)
    errs = gu.prognostic_errors(states, ref)
    assert max(errs) < 1e-13, errs
    cs = g.checksum(states)
    ref_cs = d["checksum/initial"]
    for c in (0, 1, 2, 4):
        assert abs(cs[c] - ref_cs[c]) <= 1e-12 * abs(ref_cs[0] if c == 1 else ref_cs[c])
