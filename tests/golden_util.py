"""Helpers shared by the parity tests: fixture loading and grid construction."""
import os
import sys
import numpy as np
from tempestmodel_amd.cubed_sphere import (CubedSphereGrid, BaroclinicWaveJW, BaroclinicWaveJWSmallPlanet, ScharMountainSphere,
                                           FlatPlanetCase)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# TMX_PARITY_REPORT=<file>: every error vector computed by prognostic_errors / tracer_errors is appended to <file>
# with the calling test and line, so that one run shows which comparisons are bit-exact and which are not.
_REPORT = os.environ.get("TMX_PARITY_REPORT")


def _report(kind, errs):
    if not _REPORT:
        return
    import inspect
    fr = inspect.stack()[2]
    with open(_REPORT, "a") as f:
        f.write("%s:%d %s %s %s\n" % (os.path.basename(fr.filename), fr.lineno, fr.function, kind, " ".join("%.2e" % v for v in errs)))
ARS343_GAMMA = 0.4358665215084590


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k.replace("__", "/"): z[k] for k in z.files}


def make_grid(ne, L, npatch, ztop=30000.0, pert="exp", case="jw", ntracers=0):
    """case: "jw" Jablonowski-Williamson baroclinic wave, "schar" Schar mountain on the reduced-radius sphere
    (Rayleigh layer; ScharMountainSphereTest.cpp defaults), "smallplanet" the JW state on a planet reduced by 120 without
    rotation (the supercell configuration: uniform diffusion, --explicitvertical)."""
    ppd = int(round((npatch / 6) ** 0.5))
    g = CubedSphereGrid(ne, L, ztop, ppd=ppd)
    if case == "jw":
        test = BaroclinicWaveJW(0.0, ztop, pert, ntracers=ntracers)
    elif case == "smallplanet":
        test = BaroclinicWaveJWSmallPlanet(ztop, pert, ntracers=ntracers)
    elif case == "supercell":       # SupercellTest.cpp: X = 120, no rotation, uniform diffusion, reference length / X
        test = FlatPlanetCase(ztop, X=120.0, rotating=False, ntracers=3, udiff=(1500.0, 500.0), scale_reference_length=1.0 / 120.0)
    elif case == "heldsuarez":      # HeldSuarezTest.cpp: HasRayleighFriction() with strength 0.0 x profile
        test = FlatPlanetCase(ztop, rayleigh=True)
    else:
        test = ScharMountainSphere(ztop=ztop)
    states = g.evaluate_test_case(test)
    return g, states


def grid_from_fixture(d, override=True, case="jw", ntracers=0):
    """Grid for a fixture; with ``override`` the geometry / operator arrays are replaced by the
    reference's own (bit-identical inputs for per-call parity)."""
    ne = int(d["cfg/ne"][0]); L = int(d["cfg/levels"][0]); npatch = int(d["cfg/npatch"][0])
    ztop = float(d["grid/ztop"][0]) if "grid/ztop" in d else 30000.0
    g, states = make_grid(ne, L, npatch, ztop, case=case, ntracers=ntracers)
    if override:
        for P in g.patches:
            for nm in list(P.geom):
                k = "p%d/%s" % (P.index, nm)
                if k in d:
                    v = d[k]
                    have = np.shape(P.geom[nm])
                    if v.ndim >= 3 and v.shape != have and v.shape[2] == 1:
                        # flat topography: the fixture keeps ONE level of the level-independent 3-D arrays
                        v = np.broadcast_to(v, have).copy()
                    P.geom[nm] = v
        for nm in g.ops:
            g.ops[nm] = (d["op/%s/coeff" % nm], d["op/%s/begin" % nm], d["op/%s/end" % nm])
        g.dx = d["op/dx_basis_1d"]
        g.stiff = d["op/stiffness_1d"]
        override_halo_transforms(g, d)
        for P in g.patches:     # GridPatchGLL::GetElementDeltaA / B: per patch, not exactly pi / (2 ne)
            if "p%d/element_delta_a" % P.index in d:
                P.delta_a = float(d["p%d/element_delta_a" % P.index][0])
                P.delta_b = float(d["p%d/element_delta_b" % P.index][0]) if "p%d/element_delta_b" % P.index in d else P.delta_a
    return g, states


def override_halo_transforms(g, d):
    """Replace the synthesised covector panel transforms of the halo ring by the reference's own (probed from
    CubedSphereTrans::CoVecPanelTrans by the dump driver, ``halo_trans/p*/e*``): the synthesiser derives them from the 3-D
    embedding, which agrees to 1e-15 but not bit for bit, and the DSS of (U,V) on panel edges sees the difference."""
    n = 0
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
                h["trans"][m] = M[j if e in (0, 2) else i]
                n += 1
    return n


def full_states(d, tag, npatch):
    return [(d["state/%s/p%d/node" % (tag, p)], d["state/%s/p%d/redge" % (tag, p)]) for p in range(npatch)]


def full_tracers(d, tag, npatch):
    return [d["state/%s/p%d/tracers" % (tag, p)] for p in range(npatch)]


def expand_compact_tracers(d, tag, grid):
    """Compact fixture tracers (interior nodes) -> full [nt][na][nb][L] arrays, halo zero."""
    out = []
    for P in grid.patches:
        c = d["state/%s/p%d/tracers" % (tag, P.index)]
        t = np.zeros((c.shape[0], P.na, P.nb, grid.L))
        t[:, 1:-1, 1:-1] = c
        out.append(t)
    return out


def tracer_errors(got, ref):
    """Max abs difference per tracer over interior nodes, relative to the max abs value of that tracer in ``ref``."""
    nt = ref[0].shape[0]
    errs = []
    for c in range(nt):
        num = max(float(np.max(np.abs(a[c, 1:-1, 1:-1] - b[c, 1:-1, 1:-1]))) for a, b in zip(got, ref))
        den = max(float(np.max(np.abs(b[c, 1:-1, 1:-1]))) for b in ref)
        errs.append(num / den if den > 0 else num)
    _report("tracers", errs)
    return errs


def expand_compact(d, tag, grid):
    """Compact fixture state (interior prognostic slots) -> full reference-layout arrays, halo and
    scratch slots zero."""
    out = []
    L = grid.L
    for P in grid.patches:
        n = np.zeros((5, P.na, P.nb, L)); e = np.zeros((5, P.na, P.nb, L + 1))
        n[[0, 1, 2, 4], 1:-1, 1:-1] = d["state/%s/p%d/node" % (tag, P.index)]
        e[3, 1:-1, 1:-1] = d["state/%s/p%d/redge" % (tag, P.index)]
        k0 = "state/%s/p%d/redge0" % (tag, P.index)
        if k0 in d:     # interface-level-0 scratch entries of rho*theta and rho (what HeldSuarezPhysics reads as "surface")
            e[[2, 4], 1:-1, 1:-1, 0] = d[k0]
        out.append((n, e))
    return out


def prognostic_errors(got, ref, interior=True):
    """Max abs difference per variable (U,V,rhotheta,W,rho) over the prognostic slots, relative to the
    max abs value of that variable in ``ref``."""
    errs = []
    for c in range(5):
        m = 0.0; s = 0.0
        loc = 1 if c == 3 else 0
        for (gn, ge), (rn, re_) in zip(got, ref):
            a = (ge if loc else gn)[c]; b = (re_ if loc else rn)[c]
            if interior:
                a = a[1:-1, 1:-1]; b = b[1:-1, 1:-1]
            m = max(m, float(np.max(np.abs(a - b)))); s = max(s, float(np.max(np.abs(b))))
        errs.append(m / s if s > 0 else m)
    _report("state", errs)
    return errs


def interp_case(d):
    """Grid, state, tracers and sample-point description of the output-interpolation fixture (interp_ne4_L6_p6.npz)."""
    g, _ = make_grid(4, 6, 6, ntracers=2)
    for P in g.patches:
        for nm in ("deriv_r_redge", "ref_node", "ref_redge"):
            P.geom[nm] = d["p%d/%s" % (P.index, nm)]
    pts = {k.split("/")[1]: v for k, v in d.items() if k.startswith("interp/")}
    return g, full_states(d, "interp_in", 6), full_tracers(d, "interp_in", 6), pts


def interp_rel(x, y):
    return max(float(np.max(np.abs(x[c] - y[c])) / max(np.max(np.abs(y[c])), 1e-300)) for c in range(x.shape[0]))


_EXACT_TOL = None
_EXACT_WHY = ""


def exact_tolerance():
    """Tolerance of the comparisons that are bit for bit on this project's hosts: 0.0 where the host's libm is the one
    tempestmodel_amd/csrc/tmx_refmath.h restates (glibc >= 2.28, x86-64 FMA variant: the build container and the GPU boxes),
    checked by compiling tests/native/refmath_check.c and running it on 4e5 arguments.  Only when that probe RAN and found
    another libm (no FMA units, another libm, another architecture) do the assertions fall back to the north-star tolerance
    1e-10 -- the device, the C oracle and the golden vectors still agree to rounding error there.  A probe that could not run
    (no gcc, compile error) is an error, not a reason to relax: set TMX_ALLOW_INEXACT=1 to accept 1e-10 on such a host.
    TMX_REQUIRE_EXACT=1 fails instead of relaxing in either case.  The tolerance in use is printed in the pytest header
    (tests/conftest.py) and by exact_tolerance_report()."""
    global _EXACT_TOL, _EXACT_WHY
    if _EXACT_TOL is None:
        import subprocess, tempfile
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        ran, ok, why = False, False, ""
        try:
            with tempfile.TemporaryDirectory() as td:
                exe = os.path.join(td, "refmath_check")
                subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(root, "tempestmodel_amd", "csrc"), "-o", exe,
                                os.path.join(root, "tests", "native", "refmath_check.c"), "-lm"], check=True,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                rc = subprocess.run([exe, "400000"], stdout=subprocess.DEVNULL).returncode
                ran, ok = True, rc == 0
                if not ok:
                    why = "this host's libm is not the one tmx_refmath.h restates (refmath_check exit code %d)" % rc
        except Exception as ex:      # the probe itself failed: no gcc, compile error
            why = "the libm probe could not run (%s: %s)" % (type(ex).__name__, str(ex)[:200])
        require = os.environ.get("TMX_REQUIRE_EXACT", "") not in ("", "0")
        if ok:
            _EXACT_TOL, _EXACT_WHY = 0.0, "bit for bit (host libm == tmx_refmath.h, checked on 4e5 arguments)"
        elif require:
            raise RuntimeError("TMX_REQUIRE_EXACT=1: exact comparisons are not possible on this host: " + why)
        elif ran or os.environ.get("TMX_ALLOW_INEXACT", "") not in ("", "0"):
            _EXACT_TOL, _EXACT_WHY = 1e-10, "RELAXED to 1e-10: " + why
            sys.stderr.write("golden_util: %s: exact comparisons use 1e-10\n" % why)
        else:
            raise RuntimeError("golden_util.exact_tolerance: " + why + "; fix the host or set TMX_ALLOW_INEXACT=1 to compare with 1e-10")
    return _EXACT_TOL


def exact_tolerance_report():
    """One line for test reports: the tolerance the `exact` assertions use on this host, and why."""
    tol = exact_tolerance()
    return "exact-comparison tolerance = %g (%s)" % (tol, _EXACT_WHY)



def live_reference_required():
    """Whether a missing oracle/_ref binary fails a test instead of skipping it.  TMX_REQUIRE_LIVE_REF=1 / 0 decides; unset, the live
    reference is REQUIRED wherever a GPU is present -- the strongest parity tests (the device against the real reference run on the same
    host) are `-m gpu` tests, oracle/_ref travels to the GPU box with the snapshot, and a run there that silently skipped them would
    stay green without its proof.  Set TMX_REQUIRE_LIVE_REF=0 on a GPU host that has no /root/reference to build oracle/_ref from."""
    v = os.environ.get("TMX_REQUIRE_LIVE_REF", "")
    if v != "":
        return v != "0"
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def need_live_reference(*paths, what=None):
    """Fail (where the live reference is required) or skip when one of the oracle/_ref binaries is missing."""
    import pytest
    missing = [p for p in paths if not os.path.exists(p)]
    if not missing:
        return
    msg = "%s not built / did not travel with the snapshot (oracle/_ref is built from /root/reference in the build container%s)" % (
        ", ".join(os.path.relpath(m, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) for m in missing), "; " + what if what else "")
    if live_reference_required():
        pytest.fail(msg + " -- required on a GPU host (TMX_REQUIRE_LIVE_REF=0 turns this into a skip)")
    pytest.skip(msg)
