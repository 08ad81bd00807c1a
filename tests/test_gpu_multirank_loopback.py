"""Several rank engines in one process on one GPU (loopback wire): packing, ghost indexing, tile lists, boundary-first stages and the DSS with remote members are the production code; results identical to one engine."""
import os
import numpy as np
import pytest
import golden_util as gu
from parity_common import EXACT, TOL_CALL, TOL_STEPS, _cmp, UDIFF, _supercell_case, _rank_engines_step, INFO_EARLY_TILES, INFO_LATE_TILES      # noqa: F401

pytestmark = pytest.mark.gpu


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
