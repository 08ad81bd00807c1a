"""N>1 path on CPU: world_size-2 gloo processes execute the engine's exchange plan (send lists, ghost
order, DSS groups, covector matrices) on numpy data and must reproduce the single-process DSS of the
oracle on the patches they own.  The transport is gloo here and RCCL on the GPUs; the plan, ordering
and group arithmetic under test are the engine's own (plan-only engines, no kernels)."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _col_lookup(grid, owner, rank, np_):
    """column index -> (patch, i, j) for the local patches, same enumeration as the engine."""
    cols = {}
    base = 0
    for P in grid.patches:
        if owner[P.index] != rank:
            continue
        nea = (P.ga1 - P.ga0) // np_; neb = (P.gb1 - P.gb0) // np_
        for a in range(nea):
            for b in range(neb):
                for ii in range(np_):
                    for jj in range(np_):
                        cols[(base + a * neb + b) * 16 + ii * 4 + jj] = (P.index, 1 + a * np_ + ii, 1 + b * np_ + jj)
        base += nea * neb
    return cols


def _worker(rank, world, port, q):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import golden_util as gu
    from oracle_lib import Oracle
    from tempestmodel_amd.engine import Engine, default_owner
    d = gu.load("steps_ne4_L6_p24.npz")
    g, _ = gu.make_grid(4, 6, 24)
    L = g.L
    start = gu.expand_compact(d, "step2", g)
    # reference result: single-process oracle DSS over all patches
    o = Oracle(g); o.set_state(0, start); o.apply_dss(0)
    want = o.get_state(0)

    owner = default_owner(24, world)
    e = Engine(g, device=-2, rank=rank, n_ranks=world, owner=owner)
    NS = int(e.plan(3)[0])
    send = e.plan(0).reshape(-1, 4); recv = e.plan(1).reshape(-1, 4)
    grp = e.plan(2).reshape(-1, 5); gx = e.plan(4); gt = e.plan(5); M = e.plan_matrices()

    def column_values(p, i, j):      # U,V,T,R at L levels, W at L+1
        n, ed = start[p]
        return np.concatenate([n[0, i, j], n[1, i, j], n[2, i, j], n[4, i, j], ed[3, i, j]])

    # exchange: one message per peer, in plan order
    ghost = np.zeros((len(recv), 5 * L + 1))
    reqs = []
    bufs = {}
    for peer in range(world):
        if peer == rank:
            continue
        rows = send[send[:, 3] == peer]
        out = torch.from_numpy(np.stack([column_values(*r[:3]) for r in rows]))
        reqs.append(dist.isend(out, peer))
        n_in = int((recv[:, 3] == peer).sum())
        bufs[peer] = torch.zeros((n_in, 5 * L + 1), dtype=torch.float64)
        reqs.append(dist.irecv(bufs[peer], peer))
    for r in reqs:
        r.wait()
    for peer, b in bufs.items():
        ghost[recv[:, 3] == peer] = b.numpy()

    # group averaging exactly as k_dss does it
    cols = _col_lookup(g, owner, rank, g.np)
    got = [(n.copy(), ed.copy()) for (n, ed) in start]
    worst = 0.0
    for gi, row in enumerate(grp):
        n = row[0]; mem = row[1:1 + n]
        vals = []
        for c in mem:
            vals.append(column_values(*cols[c]) if c < NS else ghost[c - NS])
        vals = np.array(vals)

        ty = [(int(gt[gi]) >> (2 * m)) & 3 for m in range(4)]

        def comb(v, m):
            # k_dss / GridCSGLL::ApplyDSS order of operations (see tmx_host.hip, "device group tables")
            if n == 2:
                return 0.5 * (v[0] + v[1])
            if n == 4:
                pr = {0: ((0, 1), (2, 3)), 1: ((0, 2), (1, 3)), 2: ((0, 3), (1, 2))}[ty[m]]
                return 0.5 * (0.5 * (v[pr[0][0]] + v[pr[0][1]]) + 0.5 * (v[pr[1][0]] + v[pr[1][1]]))
            nx, pv = (m + 1) % 3, (m + 2) % 3
            f, s2 = (pv, nx) if ty[m] else (nx, pv)
            return (1.0 / 3.0) * ((v[m] + v[f]) + v[s2])
        res = np.array([comb(vals, m) for m in range(n)])
        if gx[gi] >= 0:
            T = M[gx[gi]]
            ua = vals[:, 0:L]; ub = vals[:, L:2 * L]
            for m in range(n):
                fa = np.array([ua[q] if q == m else T[m, q, 0, 0] * ua[q] + T[m, q, 0, 1] * ub[q] for q in range(n)])
                fb = np.array([ub[q] if q == m else T[m, q, 1, 0] * ua[q] + T[m, q, 1, 1] * ub[q] for q in range(n)])
                res[m, 0:L] = comb(fa, m)
                res[m, L:2 * L] = comb(fb, m)
        for m, c in enumerate(mem):
            if c >= NS:
                continue
            p, i, j = cols[c]
            wn, we = want[p]
            ref = np.concatenate([wn[0, i, j], wn[1, i, j], wn[2, i, j], wn[4, i, j], we[3, i, j]])
            scale = np.array([np.abs(wn[0]).max()] * L + [np.abs(wn[1]).max()] * L + [np.abs(wn[2]).max()] * L
                             + [np.abs(wn[4]).max()] * L + [max(np.abs(we[3]).max(), 1e-300)] * (L + 1))
            worst = max(worst, float(np.max(np.abs(res[m] - ref) / scale)))
    e.close()
    q.put((rank, worst, len(send), len(recv)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_and_dss_match_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, ns, nr in res:
        assert ns > 0 and nr > 0
        assert worst == 0.0, (rank, worst)      # the plan reproduces the reference's DSS (the oracle's) bit for bit
