"""CPU-side tests: the C-ABI library loads and exports every symbol the header declares (no compute
call is made without a GPU), and the host logic of the engine (layout, DSS groups, unique columns,
multi-rank exchange plan) is consistent -- exercised on 'plan only' engines (cfg.device = -2)."""
import ctypes
import os
import re
import numpy as np
import pytest
import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "tempest_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tmx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from tempestmodel_amd.engine import load_library
    syms = _declared_symbols()
    assert len(syms) >= 25
    for flavour in (None, "experiments"):      # the production library and the one with the archived experiments compiled in: one ABI
        lib = load_library(flavour)
        for s in syms:
            assert hasattr(lib, s), "libtempest_mi355x%s.so does not export %s" % ("_exp" if flavour else "", s)


def test_no_cpu_fallback():
    """Without a GPU the product refuses to run; it never routes to the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tempestmodel_amd.engine import Engine, TempestError
    g, _ = gu.make_grid(2, 4, 6)
    with pytest.raises(TempestError) as ei:
        Engine(g)
    assert "no CPU path" in str(ei.value)
    src = open(os.path.join(ROOT, "tempestmodel_amd", "engine.py")).read()
    assert "oracle" not in src.replace("no CPU fallback", "")


def test_unsupported_configurations_are_rejected():
    from tempestmodel_amd import engine as eng
    lib = eng.load_library()
    cfg = eng.TmxConfig()
    cfg.abi_version = eng.TMX_ABI_VERSION
    cfg.horizontal_order, cfg.vertical_order, cfg.levels = 4, 1, 10
    cfg.n_patches, cfg.n_instances, cfg.hypervis_order = 6, 7, 4
    cfg.n_ranks, cfg.device = 1, -2
    h = ctypes.c_void_p()
    for field, bad in (("horizontal_order", 5), ("vertical_order", 2), ("n_tracers", 17), ("n_tracers", -1), ("hypervis_order", 3), ("abi_version", 99)):
        c2 = eng.TmxConfig.from_buffer_copy(cfg)
        setattr(c2, field, bad)
        assert lib.tmx_create(ctypes.byref(c2), ctypes.byref(h)) != 0
        assert len(lib.tmx_last_error()) > 0
    # uniform diffusion needs the fully explicit vertical mode and ztop; with both it is accepted
    c2 = eng.TmxConfig.from_buffer_copy(cfg); c2.uniform_diffusion = 1; c2.ztop = 2.0e4
    assert lib.tmx_create(ctypes.byref(c2), ctypes.byref(h)) != 0 and b"fully explicit" in lib.tmx_last_error()
    c2.fully_explicit = 1; c2.ztop = 0.0
    assert lib.tmx_create(ctypes.byref(c2), ctypes.byref(h)) != 0
    c2.ztop = 2.0e4
    assert lib.tmx_create(ctypes.byref(c2), ctypes.byref(h)) == 0
    lib.tmx_destroy(h)
    assert lib.tmx_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    lib.tmx_destroy(h)


@pytest.mark.parametrize("ne,L,npatch", [(3, 5, 6), (4, 6, 24)])
def test_layout_groups_and_unique_columns(ne, L, npatch):
    from tempestmodel_amd.engine import Engine
    g, _ = gu.make_grid(ne, L, npatch)
    e = Engine(g, device=-2)
    NS, ncol, nuniq, ngroups, nx = e.plan(3)
    assert ncol == 6 * ne * ne * 16 and NS % 64 == 0 and NS >= ncol
    ppd = int(round((npatch / 6) ** 0.5)); epp = ne // ppd
    assert nuniq == npatch * (3 * epp + 1) ** 2          # unique columns per patch
    grp = e.plan(2).reshape(-1, 5)
    n = grp[:, 0]
    # every stored node is either interior to an element or in exactly one group
    members = grp[:, 1:][grp[:, 1:] >= 0]
    assert len(np.unique(members)) == len(members)
    assert len(members) == ncol * 12 // 16               # 12 of 16 nodes of an element lie on its boundary
    assert (n == 3).sum() == 8                            # the 8 cube corners
    assert set(np.unique(n)) <= {2, 3, 4}
    # unique global points: 6 (3 ne + 1)^2 - duplicates across panels ... check through Euler count
    n_points = (ncol - len(members)) + len(grp)
    assert n_points == 6 * (3 * ne) ** 2 + 2              # continuous GLL grid on the cubed sphere
    assert nx > 0
    e.close()


@pytest.mark.parametrize("ne,npatch,nr", [(4, 24, 3), (6, 54, 5), (8, 96, 7), (6, 54, 3), (4, 24, 7)])
def test_rank_plans_are_mutually_consistent(ne, npatch, nr):
    """What rank s sends to rank r, in wire order, is what r expects from s, in ghost order -- also on odd layouts (54 / 96 patches over
    3 / 5 / 7 ranks: uneven patch counts per rank, ranks that are not neighbours)."""
    from tempestmodel_amd.engine import Engine
    g, _ = gu.make_grid(ne, 6, npatch)
    es = [Engine(g, device=-2, rank=r, n_ranks=nr) for r in range(nr)]
    sends = [e.plan(0).reshape(-1, 4) for e in es]
    recvs = [e.plan(1).reshape(-1, 4) for e in es]
    pairs = 0
    for r in range(nr):
        for s in range(nr):
            if r == s:
                continue
            a = sends[s][sends[s][:, 3] == r][:, :3]       # what s sends to r, in wire order
            b = recvs[r][recvs[r][:, 3] == s][:, :3]       # what r expects from s, in ghost order
            assert np.array_equal(a, b)
            pairs += len(a) > 0
    assert pairs >= 2 * (nr - 1)      # every rank has a neighbour, both directions
    assert sum(len(e.local_patches) for e in es) == npatch
    for e in es:
        e.close()


def test_active_state_image_size_is_the_reference_container_size():
    """tmx_active_state_bytes = GridPatch::GetDataContainerActiveState().GetTotalByteSize() (GridPatch.cpp:359-361,387;
    DataContainer.cpp): an 8-byte patch-index chunk, the node and interface state arrays and the tracer array, halo ring
    included.  Host logic only (plan-only engine)."""
    from tempestmodel_amd.engine import Engine
    g, _ = gu.make_grid(3, 5, 6, ntracers=2)
    e = Engine(g, device=-2)
    try:
        for P in g.patches:
            nn = P.na * P.nb
            assert e.lib.tmx_active_state_bytes(e.h, P.index) == 8 * (1 + 5 * nn * 5 + 5 * nn * 6 + 2 * nn * 5)
        assert e.lib.tmx_active_state_bytes(e.h, 99) == -1
    finally:
        e.close()


def test_default_owner_leaves_no_rank_empty():
    """Patch -> rank map of bench.py / Engine: contiguous blocks, the remainder spread, a clear error past one patch per rank."""
    from tempestmodel_amd.engine import default_owner
    for n in range(1, 25):
        own = default_owner(24, n)
        assert own == sorted(own) and set(own) == set(range(n))
        counts = [own.count(r) for r in range(n)]
        assert max(counts) - min(counts) <= 1
    assert default_owner(24, 8) == [p // 3 for p in range(24)]
    with pytest.raises(ValueError, match="at least one patch"):
        default_owner(6, 8)


def test_refmath_is_glibc_bit_for_bit(tmp_path):
    """tempestmodel_amd/csrc/tmx_refmath.h (the exp / log the device evaluates in the Exner function) restates glibc's
    FMA-build exp / log; its CPU build must agree with this image's libm bit for bit: 2e7 arguments, half of them on
    the ranges the hot path uses (R/p0 * rho*theta in (0.05, 1.15), including the near-1 branch of log)."""
    import subprocess
    exe = str(tmp_path / "refmath_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "tempestmodel_amd", "csrc"), "-o", exe,
                    os.path.join(ROOT, "tests", "native", "refmath_check.c"), "-lm"], check=True)
    r = subprocess.run([exe, "20000000"], stdout=subprocess.PIPE, universal_newlines=True)
    assert r.returncode == 0, r.stdout
    assert "exp mismatches 0 log mismatches 0" in r.stdout


def test_peer_to_peer_halo_needs_a_device_engine_of_a_multi_rank_layout():
    """tmx_halo_p2p_* argument checking on the host (plan-only engines: no device is touched): the blob size is the documented
    handle + offsets record, and neither a plan-only nor a single-rank engine can export."""
    from tempestmodel_amd.engine import Engine, TempestError, load_library
    g, _ = gu.make_grid(4, 6, 24)
    e = Engine(g, device=-2, rank=1, n_ranks=3)
    try:
        lib = load_library()
        n = lib.tmx_halo_p2p_blob_bytes(e.h)
        assert n >= 64 + 8 + 8 + 4 * 4 and n % 4 == 0
        with pytest.raises(TempestError):
            e.halo_p2p_export()                      # plan-only: no device memory to export
        with pytest.raises(ValueError):
            e.halo_p2p_connect([b"\0" * n] * 2)      # one blob per rank
        with pytest.raises(TempestError):
            e.halo_p2p_connect([b"\0" * n] * 3)      # nothing exported yet
        with pytest.raises(TempestError):
            e.halo_p2p_enable(True)                  # not connected
    finally:
        e.close()


def test_stepper_programs_share_the_copies_in_front_of_a_stub_implicit_step():
    """Host logic of tmx_step, no device: which whole-instance copies a scheme's program makes per step, and which it
    replaces by letting the copy read its source's slot (tmx_debug_program_copies runs the program's own decision code
    without launches).  ARS343 in the fully explicit vertical mode and in shallow water: the three CopyData in front of
    V.StepImplicit (TimestepSchemeARS343.cpp:150-236) are shared, none is made, also in steady state (the slots shared at
    the end of a step are re-validated against the next step's program); with implicit vertical dynamics the U,V slabs of
    those copies are shared (the solve rewrites the rest).  Every scheme: sharing never adds copies, and copies + shared is what the program asks for."""
    from tempestmodel_amd.engine import Engine, load_library
    lib = load_library()
    lib.tmx_debug_program_copies.restype = ctypes.c_int
    lib.tmx_debug_program_copies.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_int)] * 2

    def run(scheme, mode, steps, share):
        c, s = ctypes.c_int(-1), ctypes.c_int(-1)
        assert lib.tmx_debug_program_copies(Engine.SCHEMES[scheme], mode, steps, share, ctypes.byref(c), ctypes.byref(s)) == 0
        return c.value, s.value

    for steps in (1, 2, 5):
        assert run("ars343", 1, steps, 1) == (0, 3)
        assert run("ars343", 1, steps, 0) == (3, 0)
        assert run("ars343", 2, steps, 1) == (0, 3)
        assert run("ars343", 2, steps, 0) == (3, 0)
    assert run("ars232", 1, 3, 1) == (2, 2)
    # implicit vertical dynamics: the column solve rewrites rho*theta, W, rho of the copy; its U,V slabs are shared instead of copied
    for steps in (1, 2, 5):
        assert run("ars343", 0, steps, 1) == (0, 3)
        assert run("ars343", 0, steps, 0) == (3, 0)
    for scheme in Engine.SCHEMES:
        for mode in (0, 1, 2):
            for steps in (1, 3):
                made1, shared1 = run(scheme, mode, steps, 1)
                made0, shared0 = run(scheme, mode, steps, 0)
                assert shared0 == 0 and made1 <= made0, (scheme, mode, steps)
                # a shared copy is either never needed or made later (when its slot is written in place): never more than asked for
                assert made1 + shared1 >= made0 and made1 >= made0 - shared1, (scheme, mode, steps)
    # run() above also asserted the symbolic soundness check of the dry run (return code 0: every read through the instance maps
    # finds the value the reference's instance holds) -- for every scheme, mode, with and without sharing; over more steps:
    for scheme in Engine.SCHEMES:
        for mode in (0, 1, 2):
            run(scheme, mode, 8, 1)
    # ... and the check has teeth: with every share accepted and no copy-on-write (share = 2) it reports stale reads
    c = ctypes.c_int(); s = ctypes.c_int()
    stale = [lib.tmx_debug_program_copies(Engine.SCHEMES[sch], 0, 4, 2, ctypes.byref(c), ctypes.byref(s)) for sch in ("ars222", "ars443", "ark232", "strang")]
    assert all(rc <= -1000 for rc in stale), stale
    assert lib.tmx_debug_program_copies(99, 1, 1, 1, ctypes.byref(c), ctypes.byref(s)) != 0


def test_options_are_explicit_and_reported():
    """The library reads no environment variable on its own: options arrive through tmx_set_option (validated, reported by
    tmx_options_report) or through an explicit tmx_options_from_environment call that says what it took.  Plan-only engines:
    no device needed."""
    from tempestmodel_amd import engine as eng
    lib = eng.load_library()
    cfg = eng.TmxConfig()
    cfg.abi_version = eng.TMX_ABI_VERSION
    cfg.horizontal_order, cfg.vertical_order, cfg.levels, cfg.n_patches, cfg.n_instances = 4, 1, 4, 6, 7
    cfg.hypervis_order, cfg.rank, cfg.n_ranks, cfg.device = 4, 0, 1, -2
    cfg.element_delta_a, cfg.grav, cfg.Rd, cfg.cp, cfg.p0 = 0.1, 9.8, 287.0, 1004.5, 1e5

    def report(h):
        n = lib.tmx_options_report(h, None, 0)
        buf = ctypes.create_string_buffer(n + 1)
        lib.tmx_options_report(h, buf, n + 1)
        return dict(l.split("=", 1) for l in buf.value.decode().splitlines())

    old = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith("TMX_")}
    try:
        os.environ["TMX_VI_GROUP"] = "1"; os.environ["TMX_MFMA"] = "1"; os.environ["TMX_QUIET"] = "1"
        h = ctypes.c_void_p()
        assert lib.tmx_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
        r = report(h)
        assert r["vi_group"] == "-1" and r["contraction_mfma"] == "0" and "from_environment" not in r      # tmx_create looked at nothing
        assert lib.tmx_options_from_environment(h) == 2
        r = report(h)
        assert r["vi_group"] == "1" and r["contraction_mfma"] == "1" and "TMX_VI_GROUP=1" in r["from_environment"]
        assert lib.tmx_set_option(h, b"unique_layout", ctypes.c_double(0.0)) == 0 and report(h)["unique_layout"] == "0"
        assert lib.tmx_set_option(h, b"no_such_option", ctypes.c_double(1.0)) != 0
        assert lib.tmx_set_option(h, b"unique_tile_shape", ctypes.c_double(7.0)) != 0
        v = ctypes.c_double(-5.0)
        assert lib.tmx_get_option(h, b"vi_group_max", ctypes.byref(v)) == 0 and v.value == 6400.0
        # archived experiments and cross-check kernels: the production library holds neither the kernels nor the switch
        lib.tmx_info.restype = ctypes.c_longlong
        assert lib.tmx_info(h, 18) == 0      # TMX_INFO_EXPERIMENTS_BUILD
        for name, value in ((b"hvis_pull", 1.0), (b"hvis_block", 1.0), (b"vi_pair", 0.0), (b"vi_split_kernels", 1.0), (b"unique_blocks", 1.0), (b"unique_tile_shape", 4.0),
                            (b"debug_skip_exchange", 1.0), (b"vx_fused", 1.0), (b"vt_explicit_v1", 1.0), (b"vi_split_back", 1.0),
                            # round 5: A/B switches no test of the production library uses went the same way
                            (b"vi_stagger", 500.0), (b"xcd_vertical", 0.0), (b"unique_xcd_order", 0.0), (b"tracer_lincomb_pass", 1.0),
                            (b"udv_separate", 1.0), (b"vt_lanes", 32.0), (b"vt_lw8", 1.0), (b"vt_row_lanes", 8.0)):
            assert lib.tmx_set_option(h, name, ctypes.c_double(value)) == -2, name      # TMX_ERR_UNSUPPORTED
            assert b"experiments flavour" in lib.tmx_last_error()
        assert lib.tmx_set_option(h, b"vi_pair", ctypes.c_double(1.0)) == 0 and lib.tmx_set_option(h, b"hvis_pull", ctypes.c_double(0.0)) == 0
        os.environ["TMX_HVIS_PULL"] = "1"      # and a variable that cannot be honoured is an error, not a silent default
        assert lib.tmx_options_from_environment(h) == -2 and b"TMX_HVIS_PULL=1" in lib.tmx_last_error()
        os.environ.pop("TMX_HVIS_PULL")
        lib.tmx_destroy(h)
        xl = eng.load_library("experiments")
        h = ctypes.c_void_p()
        assert xl.tmx_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
        assert xl.tmx_info(h, 18) == 1 and xl.tmx_set_option(h, b"hvis_pull", ctypes.c_double(1.0)) == 0
        xl.tmx_destroy(h)
    finally:
        for k in ("TMX_VI_GROUP", "TMX_MFMA", "TMX_QUIET", "TMX_HVIS_PULL"):
            os.environ.pop(k, None)
        os.environ.update(old)


def _unique_tables(e, shape):
    lib = e.lib
    lib.tmx_debug_unique_tables.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    out = {}
    for what, name in enumerate(("sizes", "t_dcol", "t_ucol", "t_sdst", "t_sred", "gsrc", "gdst", "gn", "slot_ucol", "send_slots", "u_rep", "ucol_of_dcol", "tinfo",
                                 "b_sdst", "b_sred", "blk_info", "b_gsrc", "b_gdst", "b_gn", "blks_early", "blks_late")):
        n = lib.tmx_debug_unique_tables(e.h, shape, what, None, 0)
        assert n >= 0, name
        a = np.zeros(max(n, 1), dtype=np.int32)
        assert lib.tmx_debug_unique_tables(e.h, shape, what, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), n) == n
        out[name] = a[:n]
    return out


@pytest.mark.parametrize("ne,npatch,n_ranks", [(5, 6, 1), (6, 24, 1), (12, 24, 3), (7, 6, 1), (6, 54, 5), (8, 96, 7), (9, 54, 3)])
@pytest.mark.parametrize("shape", [0, 1, 2, 3, 4])
def test_unique_layout_tables(ne, npatch, n_ranks, shape):
    """Host logic of the node-unique layout (tmxu_tables), on plan-only engines: every unique node is stored by exactly one place
    -- one lane of one tile (alone, or averaging the copies its wavefront holds), or the DSS kernel through the partial slots of all
    its copies; averaging recipes name lanes of the same tile that hold copies of the same node; the per-tile node arithmetic
    reproduces the per-thread table; what other ranks need are slots."""
    from tempestmodel_amd.engine import Engine
    g, _ = gu.make_grid(ne, 4, npatch)
    for rank in range(n_ranks):
        e = Engine(g, device=-2, rank=rank, n_ranks=n_ranks)
        T = _unique_tables(e, shape)
        NU, NUS, NTS, nslots, ngroups, nlocal, nearly, nlate, b_ngroups, b_nlocal, nblocks, nb_early, nb_late = (int(v) for v in T["sizes"])
        el = len([p for p in range(npatch) if e.owner[p] == rank]) * (ne * ne * 6 // npatch)
        per_patch = ne * ne * 6 // npatch
        side = int(round(per_patch ** 0.5))
        assert NU == len(e.local_patches) * (3 * side + 1) ** 2
        assert (NTS == (el * 16 + 63) // 64 * 64) if shape != 4 else (NTS >= el * 16 and NTS % 64 == 0)      # (the block order pads ragged blocks to whole wavefronts)
        tu, sd, sr, td = T["t_ucol"], T["t_sdst"], T["t_sred"], T["t_dcol"]
        valid = td >= 0
        assert valid.sum() == el * 16 and sorted(td[valid]) == list(range(el * 16))          # thread order = a permutation of the elements' nodes
        assert (tu[valid] == T["ucol_of_dcol"][td[valid]]).all() and (tu[~valid] == -1).all() and (sd[~valid] == -1).all()
        # who stores each unique node
        stores = np.zeros(NU, dtype=np.int64)
        np.add.at(stores, sd[sd >= 0], 1)
        slot_nodes = T["slot_ucol"][:nslots]
        slots = -2 - sd[sd <= -2]
        assert sorted(slots) == list(range(nslots)) and (slot_nodes[slots] == tu[sd <= -2]).all()
        via_dss = np.zeros(NU, dtype=np.int64)
        gd = T["gdst"].reshape(-1, 4); gs = T["gsrc"].reshape(-1, 4); gn = T["gn"]
        assert len(gn) == ngroups
        for k in range(ngroups):
            d = gd[k][gd[k] >= 0]
            assert len(set(d)) == len(d) and len(d) >= 1
            via_dss[d] += 1
            loc = gs[k][:gn[k]]
            loc = loc[(loc >= 0) & (loc < max(nslots, 1) + 63)]      # (ghosts are numbered from the padded slot count on)
            assert set(slot_nodes[loc[loc < nslots]]) <= set(d)        # a group's local copies are copies of the nodes it stores
        assert ((stores + via_dss) == 1).all(), "a unique node is stored twice or never"
        # copies handed to the DSS kernel never coincide with nodes a lane stores
        assert not (set(slot_nodes) & set(sd[sd >= 0]))
        # averaging recipes
        for t in np.nonzero(sr)[0]:
            n = sr[t] & 15
            assert n in (2, 4) and sd[t] == tu[t]
            lanes = [(sr[t] >> s) & 63 for s in (4, 10, 16)][:n - 1]
            base = (t // 64) * 64
            for l in lanes:
                assert tu[base + l] == tu[t] and sd[base + l] == -1 and base + l != t
            assert len(set(lanes)) == n - 1
        # lanes that store nothing are exactly the partners of some recipe
        partners = set()
        for t in np.nonzero(sr)[0]:
            partners |= {(t // 64) * 64 + ((sr[t] >> s) & 63) for s in (4, 10, 16)[:(sr[t] & 15) - 1]}
        assert partners == set(np.nonzero(valid & (sd == -1))[0])
        # per-tile arithmetic == per-thread table
        ti = T["tinfo"].reshape(-1, 8)
        for t in range(NTS):
            b, nj = ti[t // 64][(t % 64) // 16], ti[t // 64][4 + (t % 64) // 16]
            assert (tu[t] == -1 and b < 0) or tu[t] == b + ((t >> 2) & 3) * nj + (t & 3)
        # ---- the same one level up: blocks of (up to) four wavefronts (block kernels)
        bs, br, bi = T["b_sdst"], T["b_sred"], T["blk_info"]
        assert len(bi) == nblocks and b_ngroups <= ngroups and b_nlocal <= b_ngroups
        first, nt = bi >> 3, bi & 7
        assert ((nt >= 1) & (nt <= 4)).all() and first[0] == 0 and (first[1:] == first[:-1] + nt[:-1]).all() and first[-1] + nt[-1] == NTS // 64      # the blocks tile the thread space
        blk_of = np.repeat(np.arange(nblocks), nt * 64)
        bstores = np.zeros(NU, dtype=np.int64)
        np.add.at(bstores, bs[bs >= 0], 1)
        bvia = np.zeros(NU, dtype=np.int64)
        bgd = T["b_gdst"].reshape(-1, 4); bgs = T["b_gsrc"].reshape(-1, 4)
        assert len(T["b_gn"]) == b_ngroups
        for k in range(b_ngroups):
            d = bgd[k][bgd[k] >= 0]
            bvia[d] += 1
        assert ((bstores + bvia) == 1).all(), "block tables: a unique node is stored twice or never"
        # a copy that goes to a slot in the block tables goes to the SAME slot in the wavefront-level tables (one slot array serves both)
        assert (bs[bs <= -2] == sd[bs <= -2]).all() and set(-2 - bs[bs <= -2]) <= set(slots)
        # the DSS list behind a block kernel is a sub-list of the full one, in its order
        full = [tuple(r) for r in np.concatenate([gs, gd], axis=1)]
        sub = [tuple(r) for r in np.concatenate([bgs, bgd], axis=1)]
        it = iter(full)
        assert all(any(x == y for y in it) for x in sub)
        for t in np.nonzero(br)[0]:
            n = br[t] & 15
            assert n in (2, 4) and bs[t] == tu[t]
            base = first[blk_of[t]] * 64
            poss = [(br[t] >> sh) & 255 for sh in (4, 12, 20)][:n - 1]
            for q in poss:
                assert q < nt[blk_of[t]] * 64 and tu[base + q] == tu[t] and bs[base + q] == -1 and base + q != t
            assert len(set(poss)) == n - 1
        bpartners = set()
        for t in np.nonzero(br)[0]:
            bpartners |= {first[blk_of[t]] * 64 + ((br[t] >> sh) & 255) for sh in (4, 12, 20)[:(br[t] & 15) - 1]}
        assert bpartners == set(np.nonzero(valid & (bs == -1))[0])
        assert sorted(list(T["blks_early"]) + list(T["blks_late"])) == list(range(nblocks)) and (nb_early, nb_late) == (len(T["blks_early"]), len(T["blks_late"]))
        early_tiles = set(np.nonzero(np.isin(np.arange(NTS // 64), [first[b] + w for b in T["blks_early"] for w in range(nt[b])]))[0])
        # what other ranks need
        send = e.plan(0).reshape(-1, 4) if n_ranks > 1 else np.zeros((0, 4), dtype=np.int32)
        assert len(T["send_slots"]) == len(send) and (nearly > 0) == (len(send) > 0)
        assert nlocal <= ngroups
        e.close()


def test_every_stepper_program_has_a_node_unique_form():
    """Host logic of the node-unique interpreter (no device): all ten stepper programs decompose into its units -- fused
    [Copy | LinearCombine]; H; V; DSS stages, column solves, hyperviscosity steps, DSS, copies and combinations -- also on
    the first and the last step of a run (Strang's programs differ there)."""
    from tempestmodel_amd import engine as eng
    lib = eng.load_library()
    expect_stages = {"ars343": 4, "ars232": 3, "ars222": 2, "ars443": 4, "ark232": 4, "strang": 5, "strang_fe": 1, "strang_rk4": 4, "strang_ssp3": 3, "strang_ssprk53": 5}
    for name, sid in eng.Engine.SCHEMES.items():
        for first in (0, 1):
            for last in (0, 1):
                units = lib.tmx_debug_program_unique(sid, first, last)
                assert units >= expect_stages[name] + 1, (name, first, last, units)      # at least its explicit stages and the step after the sub-cycle
    assert lib.tmx_debug_program_unique(99, 0, 0) == -1


def test_which_programs_may_read_the_model_state_copy_by_copy():
    """Host logic of "unique_mixed" (no device): a live-in element-major instance may be read copy by copy while the step runs on the
    node-unique layout iff only explicit stages read it before the program overwrites it.  ARS343 / ARS222 / ARK232 start with an explicit
    stage of the model state and combine it into the later ones; ARS232 / ARS443 and the Strang family start with a column solve or a
    whole-instance pass of instance 0 and keep the round-4 rule (check, convert or run element-major)."""
    from tempestmodel_amd import engine as eng
    lib = eng.load_library()
    can = {"ars343": 1, "ars232": 0, "ars222": 1, "ars443": 0, "ark232": 1, "strang": 0, "strang_fe": 0, "strang_rk4": 0, "strang_ssp3": 0, "strang_ssprk53": 0}
    for name, sid in eng.Engine.SCHEMES.items():
        for first in (0, 1):
            for last in (0, 1):
                assert lib.tmx_debug_program_mixed(sid, first, last, 0) == can[name], (name, first, last)
    assert lib.tmx_debug_program_mixed(99, 0, 0, 0) == -1


def test_which_programs_keep_the_unique_layout_with_a_rayleigh_layer():
    """Host logic (no device): the Rayleigh relaxation at the end of StepAfterSubCycle leaves its result element-major; the node-unique
    interpreter serves the programs in which only copies of that result follow (the ARS / ARK family), Strang's -- which close with a
    column solve of the relaxed state -- stay element-major."""
    from tempestmodel_amd import engine as eng
    lib = eng.load_library()
    for name, sid in eng.Engine.SCHEMES.items():
        for first in (0, 1):
            assert lib.tmx_debug_program_rayleigh(sid, first, 0) == (0 if name.startswith("strang") else 1), (name, first)
    assert lib.tmx_debug_program_rayleigh(99, 0, 0) == -1


def test_stage_combinations_that_start_from_a_stored_partial_sum():
    """Host logic (no device): which explicit stages of a program can start from the leading partial sum of their combination,
    stored by an earlier stage that reads the same instances.  ARS343: operation 12 (third stage, LinearCombine of instances 0..4)
    serves operation 18 (fourth stage: 1, 0, 2, 3, 4, 5, 6 in accumulation order), which then reads four instances less; the
    second stage's three sources would serve the third stage too, but there is one slot and the larger saving wins."""
    import ctypes
    from tempestmodel_amd import engine as eng
    lib = eng.load_library()
    pairs = (ctypes.c_int * 16)(); terms = (ctypes.c_int * 8)()
    for first in (0, 1):
        assert lib.tmx_debug_program_prefix(eng.Engine.SCHEMES["ars343"], first, 0, pairs, terms, 8) == 1
        assert list(pairs[:2]) == [12, 18] and terms[0] == 4
    for name, sid in eng.Engine.SCHEMES.items():
        n = lib.tmx_debug_program_prefix(sid, 0, 0, pairs, terms, 8)
        assert n >= 0 and all(pairs[2 * k] < pairs[2 * k + 1] and terms[k] >= 1 for k in range(n)), name
        assert all(pairs[2 * k + 1] < pairs[2 * k + 2] for k in range(n - 1)), name      # one slot: the spans do not touch
    assert lib.tmx_debug_program_prefix(99, 0, 0, pairs, terms, 8) == -1


def test_lu_flavour_probe_recognises_the_callers_dgbsv():
    """tmx_lu_flavour_from_dgbsv: which band LU does the caller's LAPACK compute?  Handed the C oracle's dgbsv in either mode (a
    restatement that is pinned to the real libraries: test_oracle_vs_reference.py) it must name the mode and set the engine's option; handed
    scipy's LAPACK (an OpenBLAS with FMA kernels on this host, or whatever is installed) it must name one of the two; handed a routine
    that solves something else it must say so (-1) and leave the option alone."""
    from tempestmodel_amd import engine as eng
    from tempestmodel_amd.engine import Engine
    from oracle_lib import lib as olib
    lib = eng.load_library()
    FN = ctypes.CFUNCTYPE(None, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                          ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double),
                          ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int))
    lib.tmx_lu_flavour_from_dgbsv.argtypes = [ctypes.c_void_p, FN]
    o = olib()
    o.orc_dgbsv.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]

    def via_oracle(n, kl, ku, nrhs, ab, ldab, ipiv, b, ldb, info):
        info[0] = o.orc_dgbsv(n[0], kl[0], ku[0], ab, ldab[0], ipiv, b)

    def via_scipy(n, kl, ku, nrhs, ab, ldab, ipiv, b, ldb, info):
        from scipy.linalg import lapack
        A = np.ctypeslib.as_array(ab, shape=(n[0], ldab[0])).T.copy()      # column-major [ldab][n]
        rhs = np.ctypeslib.as_array(b, shape=(n[0],)).copy()
        lub, piv, x, inf = lapack.dgbsv(kl[0], ku[0], A, rhs)
        np.ctypeslib.as_array(b, shape=(n[0],))[:] = x
        info[0] = inf

    def wrong(n, kl, ku, nrhs, ab, ldab, ipiv, b, ldb, info):
        np.ctypeslib.as_array(b, shape=(n[0],))[:] = 1.0
        info[0] = 0

    g, _ = gu.make_grid(3, 4, 6)
    e = Engine(g, device=-2)
    try:
        for mode in (0, 1):
            o.orc_set_lu_fma(mode)
            assert lib.tmx_lu_flavour_from_dgbsv(e.h, FN(via_oracle)) == mode and e.get_option("lu_fma") == mode
        o.orc_set_lu_fma(1)
        assert lib.tmx_lu_flavour_from_dgbsv(None, FN(via_scipy)) in (0, 1)
        assert lib.tmx_lu_flavour_from_dgbsv(e.h, FN(wrong)) == -1 and b"neither" in lib.tmx_last_error() and e.get_option("lu_fma") == 1
    finally:
        o.orc_set_lu_fma(1)
        e.close()
