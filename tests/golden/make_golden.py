#!/usr/bin/env python3
"""Regenerate the golden fixtures from the REAL reference (oracle/_ref/tmx_ref_dump, built by
`make -C oracle ref` from /root/reference -- only possible in the build container).

The fixtures are data only: inputs and outputs of the reference's hot-path calls.

  percall_ne3_L5_p6.npz   geometry, operators, halo transforms and the state after each single
                          hot-path call (H explicit, V explicit, DSS, V implicit, hyperviscosity,
                          linear combination), starting from the state after 2 real ARS343 steps
  steps_ne4_L6_p24.npz    24-patch layout: prognostic interior state after 0..3 ARS343 steps
  steps_ne3_L5_p6.npz     6-patch layout: prognostic interior state after 0,1,2,5,10 steps
  schemes_ne2_L4_p6.npz   ARS232 / ARS222 / ARS443 / ARK232 / Strang: state after steps 1 and 3 (ne2, L4)
  steps_ark232_tracers_ne2_L5_p6.npz   ARK232 with 2 tracers (StepImplicitTermsExplicitly + tracer column update): steps 1 and 3
  strang_variants_ne2_L4_p6.npz   Strang with ForwardEuler / RK4 / SSP3 / SSPRK53 and with off-centering: steps 1 and 3
  sw2_ne8_strang.npz      Williamson shallow-water test 2 (BASELINE config 1): state after 0, 1, 10 Strang steps
  percall_schar_ne3_L6_p6.npz / steps_schar_ne3_L6_p6.npz   Schar mountain, reduced-radius sphere, Rayleigh layer (row H7)
  percall_tracers_ne3_L6_p6.npz / steps_tracers_ne3_L6_p6.npz   JW + 2 tracer densities (rows H8, V8)
  steps_visc2_ne3_L5_p6.npz   three ARS343 steps with hyperviscosity order 2
  percall_udiff_ne3_L6_p6.npz / steps_udiff_{ars343,strang}_ne3_L6_p6.npz   the supercell configuration without its
                          Fortran physics: planet / 120, no rotation, uniform diffusion 1500 / 500 m^2/s,
                          --explicitvertical, 2 tracers (BASELINE config 4)
  interp_ne4_L6_p6.npz     Grid::ReduceInterpolate (OutputManagerReference's interpolation): sample points,
                          per-point coefficients, state and outputs for the flag combinations
  physics_hs_ne2_L30_p6.npz   HeldSuarezPhysics::Perform input/output
  steps100_ne4_L10_p6.npz north-star gate: state after 2 warm-up steps and after 100 further ARS343 steps (ne4, L10)
  supercell_ne2_L40_p6.npz  BASELINE config 4, the stock DCMIP2016 supercell (test/dcmip2016/SupercellTest.cpp + Kessler
                          microphysics, Fortran halves compiled with amdflang -O3): reference state, state after 2 steps
                          (each followed by KesslerPhysics::Perform), one more Perform call on it, a Perform call on a
                          moistened copy (test input: all Kessler branches act), and the state after 3 further steps
  kessler_columns.npz     SUBROUTINE KESSLER (interface/kessler.f90, compiled) on 3 x 96 columns: inputs and outputs
  heldsuarez_ne3_L60_p6.npz  BASELINE config 5 shape, the stock HeldSuarezTest at L60: state after 2 steps and after 10
                          further steps, HeldSuarezPhysics::Perform after every step
  lapack_spread.json      (tests/analysis/lapack_spread.py) reference-vs-reference spread over two LAPACKs
  known_answers.json      area-weighted checksums printed by the reference
"""
import json
import os
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tmxd  # noqa: E402

DUMP = os.path.join(ROOT, "oracle", "_ref", "tmx_ref_dump")
ENV = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL")


def run(args, out):
    txt = subprocess.run([DUMP] + args + ["--out", out], env=ENV, check=True,
                         stdout=subprocess.PIPE, universal_newlines=True).stdout
    sums = {}
    for line in txt.splitlines():
        if line.startswith("checksum "):
            f = line.split()
            sums[f[1]] = [float(v) for v in f[2:]]
    return tmxd.read(out), sums


def compact_states(d, npatch, surface_slots=False):
    """Keep only prognostic slots on interior nodes: node comps (U,V,rhotheta,rho), redge comp W.  surface_slots: also the
    interface-level-0 scratch entries of rho*theta and rho (redge comps 2, 4), which HeldSuarezPhysics reads."""
    out = {}
    for k, v in d.items():
        if not k.startswith("state/"):
            continue
        if surface_slots and k.endswith("/redge"):
            out[k + "0"] = v[[2, 4], 1:-1, 1:-1, 0]
        if k.endswith("/node"):
            out[k] = v[[0, 1, 2, 4], 1:-1, 1:-1]
        elif k.endswith("/tracers"):
            out[k] = v[:, 1:-1, 1:-1]
        else:
            out[k] = v[3, 1:-1, 1:-1]
    return out


def flat_geometry(d):
    """Geometry, operators and halo transforms of a dump made with --geom 1 on a FLAT grid, with the level-independent
    3-D metric arrays cut to one level (golden_util.grid_from_fixture broadcasts them back): the reference's own inputs
    at a few hundred KB, so that the fixture can be compared bit for bit."""
    keep3 = ("jacobian", "jacobian_redge", "contra_metric_a", "contra_metric_b", "contra_metric_xi", "contra_metric_a_redge",
             "contra_metric_b_redge", "contra_metric_xi_redge", "deriv_r_node", "deriv_r_redge", "element_area_node")
    keep2 = ("jacobian2d", "contra_metric_2d_a", "contra_metric_2d_b", "coriolis_f", "topography", "topography_deriv",
             "element_delta_a", "element_delta_b")
    out = {}
    for k, v in d.items():
        if k.startswith(("grid/", "op/", "halo_trans/", "phys/")):
            out[k] = v
        elif k[0] == "p" and "/" in k:
            nm = k.split("/")[1]
            if nm in keep2:
                out[k] = v
            elif nm in keep3:
                assert all(np.array_equal(v[:, :, 0], v[:, :, kk]) for kk in range(v.shape[2])), k
                out[k] = v[:, :, :1].copy()
    return out


def save(name, rec):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k.replace("/", "__"): v for k, v in rec.items()})
    print(name, "%.2f MB" % (os.path.getsize(path) / 1e6))


def gen_supercell(tmp):
    # BASELINE config 4: the stock supercell with Kessler physics after every step (ne2, L40, dt 0.5 s)
    sc = ["--case", "supercell", "--explicitvertical", "--physics", "kessler", "--ne", "2", "--levels", "40", "--dt", "0.5"]
    d, _ = run(sc + ["--mode", "kessler", "--moisten", "1.5", "--warm", "2", "--geom", "1"], tmp)
    rec = {k: v for k, v in d.items() if k.startswith("cfg/") or k.startswith("phys/") or k.startswith("grid/")}
    rec.update(flat_geometry(d))
    cs = compact_states(d, 6)
    rec.update({k.replace("kessler_before", "warm"): v for k, v in cs.items() if k.split("/")[1].startswith("kessler")})
    for p in range(6):
        for nm in ("ref_node", "ref_redge", "ref_tracers", "kessler_prect"):
            rec["p%d/%s" % (p, nm)] = d["p%d/%s" % (p, nm)]
    rec["p0/kessler_z_levels"] = d["p0/kessler_z_levels"]
    for p in range(6):      # flat topography: one column of heights serves every node
        assert np.array_equal(d["p%d/kessler_z_levels" % p], np.broadcast_to(d["p0/kessler_z_levels"][1, 1], d["p%d/kessler_z_levels" % p].shape))
    d2, _ = run(sc + ["--mode", "steps", "--warm", "2", "--steps", "3", "--every", "3", "--geom", "0"], tmp)
    cs2 = compact_states(d2, 6)
    for k, v in cs2.items():
        if k.split("/")[1] == "warm":
            assert np.array_equal(v, rec[k]), k         # the two runs share their first two steps
        if k.split("/")[1] == "step3":
            rec[k] = v
    rec["checksum/step3"] = d2["checksum/step3"]
    save("supercell_ne2_L40_p6.npz", rec)


def gen_kessler_columns():
    # SUBROUTINE KESSLER alone (the compiled Fortran, oracle/_ref/libkessler_ref.so) on synthetic columns
    import ctypes as C
    kl = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libkessler_ref.so"))
    PD = C.POINTER(C.c_double)
    rng = np.random.default_rng(20161)
    rec = {}
    for gi, nz in enumerate((12, 40, 60)):
        ncol = 96
        inp = np.zeros((7, ncol, nz)); out = np.zeros((4, ncol, nz)); dts = np.zeros(ncol); precl = np.zeros(ncol)
        for c in range(ncol):
            dz = rng.uniform(100, 800, nz); z = np.cumsum(dz) - dz / 2
            pk = np.exp(-z / rng.uniform(7000, 9000)) ** 0.2875
            theta = rng.uniform(285, 305) + z * rng.uniform(0.002, 0.006) + rng.normal(0, 0.5, nz)
            rho = rng.uniform(1.0, 1.2) * np.exp(-z / 8000)
            qv = np.clip(rng.uniform(0, 0.02, nz) * np.exp(-z / 3000), 0, None)
            mode = c % 4
            qc = np.where(rng.random(nz) < 0.5, rng.uniform(0, 0.004, nz), 0.0) if mode > 0 else np.zeros(nz)
            qr = np.where(rng.random(nz) < 0.5, rng.uniform(0, 0.01, nz), 0.0) if mode > 1 else np.zeros(nz)
            dts[c] = float(rng.choice([0.5, 1.0, 5.0, 30.0, 150.0]))
            inp[:, c] = (theta, qv, qc, qr, rho, pk, z)
            a = [np.ascontiguousarray(x) for x in (theta, qv, qc, qr, rho, pk, z)]
            pr = C.c_double(0); dtc = C.c_double(dts[c]); n = C.c_int(nz)
            kl.kessler_(*[x.ctypes.data_as(PD) for x in a[:6]], C.byref(dtc), a[6].ctypes.data_as(PD), C.byref(n), C.byref(pr))
            out[:, c] = a[:4]; precl[c] = pr.value
        rec["g%d/in" % gi] = inp; rec["g%d/out" % gi] = out; rec["g%d/dt" % gi] = dts; rec["g%d/precl" % gi] = precl
    save("kessler_columns.npz", rec)


def gen_heldsuarez(tmp):
    # BASELINE config 5 shape: the stock Held-Suarez test at L60 (ne3, dt 200 s), forcing after every step
    hs = ["--case", "heldsuarez", "--physics", "hs", "--ne", "3", "--levels", "60", "--dt", "200"]
    d, _ = run(hs + ["--mode", "steps", "--warm", "2", "--steps", "10", "--every", "10", "--geom", "1"], tmp)
    rec = {k: v for k, v in d.items() if k.startswith("cfg/") or k.startswith("phys/") or k.startswith("checksum/") or "hs_" in k}
    rec.update(flat_geometry(d))
    cs = compact_states(d, 6, surface_slots=True)
    rec.update({k: v for k, v in cs.items() if k.split("/")[1] in ("warm", "step10")})
    save("heldsuarez_ne3_L60_p6.npz", rec)




STRANG_VARIANTS = (("strang_fe", 0.0, 50.0), ("strang_rk4", 0.0, 400.0), ("strang_ssp3", 0.0, 400.0), ("strang_ssprk53", 0.0, 400.0),
                   ("strang_rk4", 0.25, 400.0), ("strang", 0.5, 400.0))


def gen_strang_variants(tmp):
    # SURVEY 8a row T3, the other ExplicitDiscretization values of TimestepSchemeStrang (+ a non-zero off-centering): state and
    # the carried instance 1 after steps 1 and 3 (ne2, L4; forward Euler at a time step it is stable with)
    rec = {}
    for n, (scheme, oc, dt) in enumerate(STRANG_VARIANTS):
        d, _ = run(["--mode", "steps", "--ne", "2", "--levels", "4", "--dt", str(dt), "--steps", "3", "--scheme", scheme,
                    "--offcentering", str(oc), "--geom", "1" if n == 0 else "0"], tmp)
        if n == 0:      # the reference's own geometry, once: the comparison is bit for bit
            drop = ("lon", "lat", "z_levels", "z_interfaces", "element_area_node", "element_area_redge", "a_nodes", "b_nodes")
            rec.update({k: v for k, v in d.items() if not k.startswith(("state/", "checksum/")) and k.split("/")[-1] not in drop})
        name = scheme + ("_oc%02d" % round(100 * oc) if oc else "")
        for k, v in compact_states(d, 6).items():
            if k.split("/")[1].split("_")[0] in ("step1", "step3"):
                rec[k.replace("state/", "state/%s_" % name, 1)] = v
    save("strang_variants_ne2_L4_p6.npz", rec)


def gen_ark232_tracers(tmp):
    # SURVEY 8a row V2 with tracers: ARK232 (its first implicit stage is StepImplicitTermsExplicitly, which ends with the implicit
    # column update of the tracers) on the JW state with 2 tracers, ne2 L5: state + tracers after steps 1 and 3, own geometry
    d, _ = run(["--mode", "steps", "--ne", "2", "--levels", "5", "--dt", "300", "--steps", "3", "--scheme", "ark232",
                "--tracers", "2", "--geom", "1"], tmp)
    drop = ("lon", "lat", "z_levels", "z_interfaces", "element_area_redge", "a_nodes", "b_nodes")     # the filters need the areas
    rec = {k: v for k, v in d.items() if not k.startswith(("state/", "checksum/")) and k.split("/")[-1] not in drop}
    rec.update({k: v for k, v in compact_states(d, 6).items() if k.split("/")[1] in ("step1", "step3")})
    save("steps_ark232_tracers_ne2_L5_p6.npz", rec)


def main():
    tmp = "/tmp/tmx_golden.tmxd"
    if len(sys.argv) > 1:       # regenerate only the named fixtures: supercell, kessler, heldsuarez, strang
        for nm in sys.argv[1:]:
            {"supercell": lambda: gen_supercell(tmp), "kessler": gen_kessler_columns, "heldsuarez": lambda: gen_heldsuarez(tmp),
             "strang": lambda: gen_strang_variants(tmp), "ark232_tracers": lambda: gen_ark232_tracers(tmp)}[nm]()
        os.remove(tmp) if os.path.exists(tmp) else None
        return
    drop = ("lon", "lat", "z_levels", "z_interfaces", "element_area_node", "element_area_redge",
            "a_nodes", "b_nodes")
    d, _ = run(["--mode", "percall", "--ne", "3", "--levels", "5", "--dt", "300", "--warm", "2"], tmp)
    d = {k: v for k, v in d.items() if k.split("/")[-1] not in drop and not k.startswith("state/initial")}
    save("percall_ne3_L5_p6.npz", d)

    d, sums = run(["--mode", "steps", "--ne", "4", "--levels", "6", "--dt", "200", "--steps", "3",
                   "--patches", "24", "--geom", "0"], tmp)
    rec = {k: v for k, v in d.items() if k.startswith("cfg/") or k.startswith("checksum/")}
    rec.update(compact_states(d, 24))
    save("steps_ne4_L6_p24.npz", rec)

    d, sums2 = run(["--mode", "steps", "--ne", "3", "--levels", "5", "--dt", "300", "--steps", "10",
                    "--geom", "0"], tmp)
    rec = {k: v for k, v in d.items() if k.startswith("cfg/") or k.startswith("checksum/")}
    cs = compact_states(d, 6)
    keep = ("initial", "step1", "step2", "step5", "step10")
    rec.update({k: v for k, v in cs.items() if k.split("/")[1] in keep})
    save("steps_ne3_L5_p6.npz", rec)

    # the other steppers of the path: state after steps 1 and 3, tiny grid
    rec = {}
    for scheme in ("ars232", "ars222", "ars443", "ark232", "strang"):
        d, _ = run(["--mode", "steps", "--ne", "2", "--levels", "4", "--dt", "400", "--steps", "3",
                    "--scheme", scheme, "--geom", "0"], tmp)
        cs = compact_states(d, 6)
        for k, v in cs.items():
            tag = k.split("/")[1]
            if tag.split("_")[0] in ("step1", "step3"):
                rec[k.replace("state/", "state/%s_" % scheme, 1)] = v
        rec["cfg/dt"] = d["cfg/dt"]
    save("schemes_ne2_L4_p6.npz", rec)

    # configuration 1 of BASELINE.json: Williamson shallow-water test 2, ne8, default Strang/KGU35 stepper
    d, sw_sums = run(["--case", "sw2", "--mode", "steps", "--ne", "8", "--dt", "200", "--steps", "10",
                      "--scheme", "strang", "--geom", "1"], tmp)
    rec = {k: v for k, v in d.items() if k.startswith("cfg/") or k.startswith("checksum/")}
    for k, v in d.items():
        if k.startswith("state/") and k.endswith("/node") and k.split("/")[1] in ("initial", "step1", "step10"):
            rec[k] = v[:, 1:-1, 1:-1]
        if k in ("p0/jacobian2d", "p4/jacobian2d", "p0/coriolis_f", "p4/contra_metric_2d_a", "p0/jacobian", "p0/topography"):
            rec[k] = v
    save("sw2_ne8_strang.npz", rec)

    # tracer transport.  JW state + two closed-form tracer densities (JWWithTracers in ref_dump.cpp)
    d, _ = run(["--tracers", "2", "--mode", "percall", "--ne", "3", "--levels", "6", "--dt", "200", "--warm", "2"], tmp)
    drop_tr = tuple(x for x in drop if x != "element_area_node") + ("ref_node", "ref_redge")   # the filters need the areas
    d = {k: v for k, v in d.items() if k.split("/")[-1] not in drop_tr and not k.startswith("state/initial")}
    save("percall_tracers_ne3_L6_p6.npz", d)
    d, _ = run(["--tracers", "2", "--mode", "steps", "--ne", "3", "--levels", "6", "--dt", "200", "--steps", "4",
                "--geom", "0"], tmp)
    rec = {k: v for k, v in d.items() if k.startswith("cfg/") or k.startswith("checksum/")}
    rec.update(compact_states(d, 6))
    save("steps_tracers_ne3_L6_p6.npz", rec)

    # BASELINE config 4 (DCMIP2016 supercell) without the Fortran physics: uniform diffusion + fully explicit vertical
    ud = ["--planet", "120", "--udiff", "1500", "500", "--explicitvertical", "--tracers", "2", "--ne", "3", "--levels", "6", "--dt", "1"]
    d, _ = run(ud + ["--mode", "percall", "--warm", "1"], tmp)
    drop_ud = tuple(x for x in drop if x != "element_area_node")      # keeps ref_node / ref_redge / ref_tracers
    d = {k: v for k, v in d.items() if k.split("/")[-1] not in drop_ud and not k.startswith("state/initial")
         and not k.startswith("state/hypervis") and not k.startswith("state/lincomb")}
    save("percall_udiff_ne3_L6_p6.npz", d)
    for sch in ("ars343", "strang"):
        d, _ = run(ud + ["--mode", "steps", "--scheme", sch, "--steps", "3", "--geom", "0"], tmp)
        rec = {k: v for k, v in d.items() if k.startswith("cfg/") or k.startswith("checksum/")}
        rec.update(compact_states(d, 6))
        save("steps_udiff_%s_ne3_L6_p6.npz" % sch, rec)

    # hyperviscosity order 2 (plain viscosity, one pass; HorizontalDynamicsFEM.cpp:2672-2686), nu = 2e5 m^2/s
    d, _ = run(["--hypervis", "2", "--mode", "steps", "--ne", "3", "--levels", "5", "--dt", "300", "--steps", "3", "--geom", "0"], tmp)
    rec = {k: v for k, v in d.items() if k.startswith("cfg/") or k.startswith("checksum/")}
    rec.update(compact_states(d, 6))
    save("steps_visc2_ne3_L5_p6.npz", rec)

    # output interpolation
    d, _ = run(["--mode", "interp", "--warm", "2", "--ne", "4", "--levels", "6", "--dt", "200", "--tracers", "2"], tmp)
    keep_geom = ("deriv_r_redge", "ref_node", "ref_redge")
    d = {k: v for k, v in d.items() if k.startswith("cfg/") or k.startswith("interp/") or k.startswith("state/interp_in")
         or k.split("/")[-1] in keep_geom}
    save("interp_ne4_L6_p6.npz", d)

    # column physics.  HeldSuarezPhysics::Perform on the JW state after 2 steps (ne2, L30: the
    # boundary-layer friction only reaches the lowest levels)
    d, _ = run(["--mode", "physics", "--ne", "2", "--levels", "30", "--dt", "200", "--warm", "2", "--geom", "0"], tmp)
    rec = {k: v for k, v in d.items() if k.startswith("cfg/") or k.startswith("phys/") or "hs_" in k}
    save("physics_hs_ne2_L30_p6.npz", rec)

    # north-star gate: 100 reference ARS343 steps at ne4 L10 from the developed state after 2 steps
    d, _ = run(["--mode", "steps", "--ne", "4", "--levels", "10", "--dt", "200", "--warm", "2", "--steps", "100",
                "--every", "100", "--geom", "0"], tmp)
    rec = {k: v for k, v in d.items() if k.startswith("cfg/") or k.startswith("checksum/")}
    cs = compact_states(d, 6)
    rec.update({k: v for k, v in cs.items() if k.split("/")[1] in ("warm", "step100")})
    save("steps100_ne4_L10_p6.npz", rec)

    # Rayleigh friction.  Schar mountain on the reduced-radius sphere (the stock cubed-sphere test whose
    # Rayleigh strength is not identically zero), per-call chain (StepAfterSubCycle applies the friction) and 3 steps
    d, _ = run(["--case", "schar", "--mode", "percall", "--ne", "3", "--levels", "6", "--dt", "0.5", "--warm", "2"], tmp)
    d = {k: v for k, v in d.items() if k.split("/")[-1] not in drop and not k.startswith("state/initial")}
    save("percall_schar_ne3_L6_p6.npz", d)
    d, _ = run(["--case", "schar", "--mode", "steps", "--ne", "3", "--levels", "6", "--dt", "0.5", "--steps", "3",
                "--geom", "0"], tmp)
    rec = {k: v for k, v in d.items() if k.startswith("cfg/") or k.startswith("checksum/")}
    rec.update(compact_states(d, 6))
    save("steps_schar_ne3_L6_p6.npz", rec)

    gen_supercell(tmp)
    gen_kessler_columns()
    gen_heldsuarez(tmp)
    gen_strang_variants(tmp)
    gen_ark232_tracers(tmp)

    # reference stdout known answers (same command as SURVEY.md section 8c)
    _, ka = run(["--mode", "steps", "--ne", "4", "--levels", "10", "--dt", "200", "--steps", "3",
                 "--geom", "0"], tmp)
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump({"jw_ne4_L10_dt200_ars343": ka,
                   "jw_ne4_L6_p24_dt200_ars343": sums,
                   "jw_ne3_L5_dt300_ars343": sums2,
                   "sw2_ne8_dt200_strang": sw_sums}, f, indent=1)
    os.remove(tmp)


if __name__ == "__main__":
    main()
