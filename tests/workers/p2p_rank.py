#!/usr/bin/env python3
"""One rank of a multi-process run with the peer-to-peer halo transport (tmx_halo_p2p_*), for tests/test_gpu_two_process.py and
tools/p2p_timing.py.  Every rank is its own process; all of them may share HIP device 0 (a 1-GPU box).  The blobs are gathered
over torch.distributed/gloo on 127.0.0.1.  usage: p2p_rank.py <rank> <world> <port> <outdir> <scheme> <ntracers> <steps> [ne L npatch]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE); ROOT = os.path.dirname(TESTS)
for q in (ROOT, TESTS):
    if q not in sys.path:
        sys.path.insert(0, q)
import numpy as np
import torch.distributed as dist
import golden_util as gu
from tempestmodel_amd.engine import Engine


def main():
    rank, world, port, outdir, scheme, ntr, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
    ne, L, npatch = (int(sys.argv[8]), int(sys.argv[9]), int(sys.argv[10])) if len(sys.argv) > 10 else (4, 6, 24)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, st = gu.make_grid(ne, L, npatch, ntracers=ntr)
    e = Engine(g, rank=rank, n_ranks=world, device=int(os.environ.get("TMX_TEST_DEVICE", "0")))
    try:
        blobs = [None] * world
        dist.all_gather_object(blobs, e.halo_p2p_export())
        e.halo_p2p_connect(blobs)
        dist.barrier()
        e.upload_state(0, st)
        if ntr:
            e.upload_tracers(0, [g.initial_tracers[P.index] for P in g.patches])
        dt = 200.0 * 4 / ne
        if scheme == "absent-neighbour":
            # rank 0 steps alone: its exchange must give up after TMX_P2P_TIMEOUT_S and report it, not hang the device
            if rank == 0:
                try:
                    e.step("ars343", dt, first=True); e.sync()
                    outcome = "no error"
                except Exception as ex:
                    outcome = str(ex)
                with open(os.path.join(outdir, "rank0.txt"), "w") as f:
                    f.write(outcome)
            dist.barrier()
            return
        if scheme.startswith("recover-"):
            # rank 0 steps alone and times out (TMX_P2P_TIMEOUT_S); then ALL ranks reset the transport between two barriers,
            # upload the state again and step together: the result must be the undisturbed one
            if rank == 0:
                try:
                    e.step("ars343", dt, first=True); e.sync()
                    outcome = "no error"
                except Exception as ex:
                    outcome = str(ex)
                with open(os.path.join(outdir, "rank0.txt"), "w") as f:
                    f.write(outcome)
            dist.barrier()
            e.halo_p2p_reset()
            dist.barrier()
            e.upload_state(0, st)
            scheme = scheme[len("recover-"):]
        e.step(scheme, dt, first=True); e.sync()
        dist.barrier()
        t0 = time.perf_counter()
        for k in range(1, steps):
            e.step(scheme, dt)
        e.sync()
        wall = time.perf_counter() - t0
        dist.barrier()
        got = e.download_state(0)
        out = {"wall": np.array([wall])}
        for p in e.local_patches:
            out["n%d" % p] = got[p][0]; out["e%d" % p] = got[p][1]
        if ntr:
            tr = e.download_tracers(0)
            for p in e.local_patches:
                out["t%d" % p] = tr[p]
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
        dist.barrier()          # neighbours write into this rank's block until they are done
    finally:
        e.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
