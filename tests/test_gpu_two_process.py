"""Real rank-to-rank halo exchange on the device: every rank is its own process with its own engine, the boundary columns
travel through the peer-to-peer transport (tmx_halo_p2p_*: HIP IPC mapped ghost buffers, arrival counters), and the ranks'
results after whole steps of the production program are bit-identical to one engine holding all patches.  The ranks share
HIP device 0, so this runs on a 1-GPU box (RCCL refuses two ranks on one device; the transport under test has no such limit)."""
import os
import socket
import subprocess
import sys
import tempfile
import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "workers", "p2p_rank.py")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


NE, LEV, NPATCH = 12, 6, 24      # ne12 on 24 patches: the smallest layout on which every rank owns early AND late tiles (boundary-first stages on)


def _run_ranks(world, scheme, ntr, steps, out, **extra_env):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), out, scheme, str(ntr), str(steps), str(NE), str(LEV), str(NPATCH)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill(); o, _ = p.communicate()
        logs.append(o.decode(errors="replace")[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)


@pytest.mark.parametrize("world,scheme,ntr,env", [
    (2, "ars343", 0, {}), (4, "ars343", 2, {}), (3, "strang", 0, {}),
    # the exchange on the engine's own stream, whole-patch launches (no boundary-first split)
    (2, "ars343", 0, {"TMX_NO_EXCHANGE_OVERLAP": "1"}), (3, "ars343", 0, {"TMX_NO_SPLIT": "1"}),
    # a time-out on rank 0 (it steps alone for one step), then tmx_halo_p2p_reset on every rank and the run proper
    (2, "recover-ars343", 0, {"TMX_P2P_TIMEOUT_S": "2"})])
def test_ranks_in_separate_processes_match_one_engine(world, scheme, ntr, env):
    from tempestmodel_amd.engine import Engine
    steps = 3
    g, st = gu.make_grid(NE, LEV, NPATCH, ntracers=ntr)
    single = Engine(g)
    try:
        single.upload_state(0, st)
        if ntr:
            single.upload_tracers(0, [g.initial_tracers[P.index] for P in g.patches])
        for k in range(steps):
            single.step(scheme.replace("recover-", ""), 200.0 * 4 / NE, first=(k == 0))
        single.sync()
        ref = single.download_state(0)
        reft = single.download_tracers(0) if ntr else None
    finally:
        single.close()
    with tempfile.TemporaryDirectory() as out:
        _run_ranks(world, scheme, ntr, steps, out, **env)
        seen = set()
        for r in range(world):
            d = np.load(os.path.join(out, "rank%d.npz" % r))
            for key in d.files:
                if key[0] != "n":
                    continue
                p = int(key[1:]); seen.add(p)
                assert np.array_equal(d["n%d" % p][[0, 1, 2, 4], 1:-1, 1:-1], ref[p][0][[0, 1, 2, 4], 1:-1, 1:-1]), (r, p)
                assert np.array_equal(d["e%d" % p][3, 1:-1, 1:-1], ref[p][1][3, 1:-1, 1:-1]), (r, p)
                if ntr:
                    assert np.array_equal(d["t%d" % p][:, 1:-1, 1:-1], reft[p][:, 1:-1, 1:-1]), (r, p)
        assert seen == set(range(24))
        if scheme.startswith("recover-"):
            msg = open(os.path.join(out, "rank0.txt")).read()
            assert "did not arrive" in msg, msg       # the lone step did fail, and said so


def test_bench_runs_as_two_rank_processes_on_one_device():
    """bench.py under torch.distributed.run with two ranks (the driver's multi-GPU launch line), both on device 0 with the
    peer-to-peer halo; TMX_BENCH_AUTO_SELFTEST walks the start-up comparison of `--halo auto` as well."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TMX_BENCH_AUTO_SELFTEST="1")
    env.pop("TMX_HALO", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--develop", "1", "--ne", "8", "--levels", "6", "--dt", "500", "--share-device", "--halo", "auto", "--no-cpu-baseline", "--sustained-steps", "50"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=400, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert "identical bytes" in d["config"]["parallelism"], d["config"]["parallelism"]
    # the line's own parity proof: rank 0's patches after three steps against a single-rank engine stepped beside it
    par = d["multi_rank"]["parity"]
    assert par["identical_to_single_rank_engine"] is True and par["steps"] == 3, par
    assert d["multi_rank"]["halo_transport"] == "p2p" and d["sustained"]["steps"] > 0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (how a driver may spell the multi-GPU run): the script starts torch.distributed.run
    itself -- as a child process, before anything in the parent has touched the GPU -- and relays rank 0's JSON line and the exit code.
    Both ranks on device 0 here (peer-to-peer halo); the line carries every rank's kernel split."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TMX_HALO"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--develop", "1", "--ne", "8", "--levels", "6",
                        "--dt", "500", "--share-device", "--halo", "p2p", "--no-cpu-baseline", "--sustained-steps", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=400, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
    assert "without a launcher" in r.stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["multi_rank"]["parity"]["identical_to_single_rank_engine"] is True
    split = d["multi_rank"]["per_rank_kernel_ms_per_step"]
    assert len(split) == 2 and all(s_["vi_solve"] > 0 and s_["h_explicit"] > 0 for s_ in split)
    # a world that contradicts --gpus is refused, not re-launched
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=dict(env, RANK="0", WORLD_SIZE="1"),
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=120, cwd=root)
    assert r2.returncode != 0 and "must equal WORLD_SIZE" in (r2.stderr + r2.stdout)


def test_absent_neighbour_is_reported_not_hung():
    """A rank whose neighbour never sends: the wait kernel gives up after TMX_P2P_TIMEOUT_S and tmx_sync returns TMX_ERR_COMM."""
    with tempfile.TemporaryDirectory() as out:
        _run_ranks(2, "absent-neighbour", 0, 1, out, TMX_P2P_TIMEOUT_S="2")
        msg = open(os.path.join(out, "rank0.txt")).read()
        assert "did not arrive" in msg, msg
