#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: whole ARS343 IMEX steps of the Jablonowski-Williamson
baroclinic wave on the cubed sphere (BASELINE.json metric: element-DOF-updates/s and SYPD).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one full TimestepSchemeARS343::Step (4 H-explicit, 4 V-explicit, 3 V-implicit column
solves, 6 DSS incl. exchange, stage algebra, two-pass hyperviscosity) on synthetic input of the
reference's own closed-form initial state, resident in HBM.  Rank 0 prints ONE JSON line.

  value       element-DOF-updates/s = N_el * 16 * L * 5 * K / wall      (whole job, all ranks)
  roofline    dominant operation: algorithmic bytes / measured device time (HIP events)
  cpu_baseline  the real reference (oracle/_ref, "reference") when its binary travelled with the
              snapshot, else the C oracle ("port"), timed on 1 host core on a bounded sample
  cpu_baseline_mpi  (extra) the same reference binary under `mpiexec -np 6`, when mpiexec is there
  cpu_baseline_allcores  (extra) the reference under `mpiexec -np 24` on its 24-patch layout when the host has >= 24 cores

The timed region starts from a DEVELOPED state: `--develop` (10) untimed steps precede the W warm-up steps, so that W != 0
and the pivot pattern of the column solve is the one of a running model; the fraction of pivot steps that
took the wave-uniform path during the profiled steps is reported as roofline.uniform_pivot_fraction.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# Algorithmic bytes per point (stored node x level) per LAUNCH of each operation, from SURVEY.md section 8(d)
# (DESIGN.md "Roofline accounting" derives them).  One ARS343 step = 4 h_explicit + 3 v_implicit + 6 dss +
# 2 hypervis launches; the stage combinations and VerticalDynamicsFEM::StepExplicit are fused into h_explicit.
#   h_explicit : H row 200 B (5 R state + 5 R + 5 W update + 10 metric) minus the 5 R of the update instance that
#                the fused combination replaces (40 B) plus the stage-combination row, 600 B/step over 4 launches
#                (150 B); the V-explicit row (80 B) is NOT counted: fusion removes its traffic altogether
#   v_implicit : 272 B (5 R + 3 W + 26 metric)
#   dss        : 60 B
#   hypervis   : 100 B per pass.  SURVEY's row is 216 B (the reference's separate scalar and vector passes); ONE kernel does
#                both here, and what it cannot avoid moving is 5 R + 5 W in pass 1 and 5 R (Laplacians) + 5 R (base) + 5 W in
#                pass 2 = 12.5 doubles per pass on average.  Pricing it with the 216 B row put it above the 8 TB/s peak.
#   lincomb    : the copies left un-fused (device copies of whole / partial instances); no SURVEY row of its own
ALG_BYTES = {"h_explicit": 200.0 - 40.0 + 150.0, "v_implicit": 272.0, "dss": 60.0, "hypervis": 100.0, "lincomb": 0.0}
# SURVEY.md 8(d)'s own per-launch rows, printed beside the figures above wherever they differ (the contract's numbers stay visible)
SURVEY_8D_BYTES = {"h_explicit": 200.0 + 80.0, "v_implicit": 272.0, "dss": 60.0, "hypervis": 216.0, "lincomb": 0.0}
ALG_BYTES_PER_STEP = 3328.0          # SURVEY.md 8(d) total: 416 doubles per point per step
# What limits each operation (the roofline that bounds the PATH is HBM bandwidth):
OP_BOUND = {"h_explicit": "hbm", "dss": "hbm (gathers: partially used sectors)", "hypervis": "hbm", "lincomb": "hbm",
            "v_implicit": "stream + two chains: the pivoted band LU must write its U rows (277 MB per launch at ne30) and read them back -- 5.6 x the "
                          "kernel's compulsory bytes -- while two dependent fp64 chains per column group run beside it: forward phase ~113 us = "
                          "max(assembly ~104 us alone, elimination ~96 us alone with its stores, 63 without) now that the roles hand over through a "
                          "three-deep LDS ring instead of a barrier per block row; back substitution ~50 us = the stream read back at ~6 TB/s "
                          "(profiles/r05_column_solve_ablation.txt, r05_pair_timing.txt)"}
# engine profile slots (HIP events) that make up each operation
OP_KERNELS = {"h_explicit": ["h_explicit", "v_explicit"], "v_implicit": ["vi_assemble", "vi_solve"],
              "dss": ["dss"], "hypervis": ["hypervis"], "lincomb": ["lincomb"]}
HBM_PEAK_GBS = 8000.0
PROFILE_TAG = next((t for t in ("r06", "r05", "r04", "r03", "r02") if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", t + "_pmc_traffic_ne30_L30_n1.json"))), "r02")
# rocprof kernel-name prefixes per operation, for the PMC traffic table (profiles/r01_pmc_traffic.json)
OP_ROCPROF = {"h_explicit": ["k_h_explicit", "k_h_w_update", "k_h_walk"], "v_implicit": ["k_vi_fused", "k_vi_pair", "k_vi_group"], "dss": ["k_dss", "k_dss_u"],
              "hypervis": ["k_hypervis", "k_hv_walk"], "lincomb": ["k_lincomb"]}


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources the loaded library was built from (tempestmodel_amd/csrc/*.hip, *.h,
    include/*.h, in name order).  tools/summarise_profiles.py stamps the committed counter tables with the same figure, so the line
    can say whether the offline numbers belong to the code that ran."""
    import glob, hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "tempestmodel_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "tempestmodel_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


_LIB_KERNELS = None


def library_kernels():
    """Base names of the __global__ kernels the loaded product library holds (mangled host-stub names in its symbol strings)."""
    global _LIB_KERNELS
    if _LIB_KERNELS is None:
        import re
        from tempestmodel_amd import engine as _eng
        path = os.environ.get("TMX_LIBRARY") or getattr(_eng, "LIB_PATH", None) or os.path.join(ROOT, "tempestmodel_amd", "libtempest_mi355x.so")
        names = set()
        try:
            blob = open(path, "rb").read()
            for m in re.finditer(rb"_Z(?:N\d+[a-z_]+?)?(\d+)(k_[a-z0-9_]+)", blob):      # (plain and namespaced: _ZN6lu_fma9k_vi_pair...)
                n = int(m.group(1))
                names.add(m.group(2)[:n].decode())
        except Exception:
            pass
        _LIB_KERNELS = names
    return _LIB_KERNELS


def kernel_base(name):
    """rocprof kernel name -> base name: return type, template arguments, parameters and namespaces (lu_fma:: / lu_nofma::) stripped."""
    return name.replace("void ", "").split("<")[0].split("(")[0].split("::")[-1]


def profile_kernels_known(names):
    """(ok, unknown): every kernel base name of a committed profile must exist in the loaded library, else its numbers are refused."""
    lib = library_kernels()
    if not lib:
        return True, []      # symbol strings not readable: nothing to check against
    unknown = sorted({n for n in names if n.startswith("k_") and n not in lib})
    return (not unknown), unknown


def traffic_table(ne, L, world):
    """The newest committed PMC table of the shape (path, parsed json) or (None, None); refused (None) when it names kernels the loaded library does not hold."""
    for t in ("r06", "r05", "r04", "r03", "r02"):      # (the newest table of the shape; r04 on: the node-unique layout, the default)
        c = os.path.join(ROOT, "profiles", "%s_pmc_traffic_ne%d_L%d_n%d.json" % (t, ne, L, world))
        if os.path.exists(c):
            try:
                tab = json.load(open(c))
            except Exception:
                return None, None
            ok, unknown = profile_kernels_known([kernel_base(k) for k in tab["kernels"]])
            if not ok:
                sys.stderr.write("bench: %s names kernels the loaded library does not hold (%s): its traffic figures are not used\n" % (os.path.relpath(c, ROOT), ", ".join(unknown)))
                return None, None
            return c, tab
    return None, None


def layout_model(eng, L, unique):
    """Bytes per launch each operation cannot avoid moving IN THE LAYOUT THAT RUNS (SURVEY 8(d)'s rule -- every needed array read or
    written once -- applied to the arrays that exist): with the node-unique layout a state / update / combination-source array has one
    entry per patch node (NU columns), the geometry is the 2-D per-(element, node) slabs the closed-form metric needs (NS columns, no
    level factor; the 26 three-dimensional metric doubles of the 8(d) row are not stored at all), the DSS reads the partial slots and
    writes the nodes on tile and patch edges.  Element-major: state arrays have NS columns.  ARS343's four explicit launches read
    1 + 4 + 6 + 7 source instances (the stage combinations are evaluated inside them)."""
    NS, NU, NPs, NG = eng.info(0), eng.info(1), eng.info(15), eng.info(16)
    closed = eng.info(6) == 1
    nslab = 5 * L + 1
    cols = NU if unique else NS
    inst = 8.0 * cols * nslab                      # one state instance
    slots = 8.0 * NPs * nslab if unique else 0.0   # per-element values of the seam nodes on tile / patch edges
    geo2d_h = 8.0 * NS * 20                        # 2-D slabs of the horizontal kernels (metric, Coriolis, spacing, closed-form factors)
    geo3d_h = 0.0 if closed else 8.0 * NS * L * 10
    geo_v = 8.0 * cols * 14 + (0.0 if closed else 8.0 * cols * L * 26)
    h_reads = (1 + 4 + 6 + 7) / 4.0                # source instances per explicit launch, ARS343 average
    out = {
        "h_explicit": h_reads * inst + geo2d_h + geo3d_h + inst + slots,
        "v_implicit": inst + 8.0 * cols * (3 * L + 1) + geo_v,
        "dss": (slots + 8.0 * NG * nslab) if unique else 0.75 * 2 * inst,
        "hypervis": 1.25 * inst + inst + slots,    # pass 1: state in, Laplacians out; pass 2: Laplacians + base in, state out
        "lincomb": 0.0,
    }
    return out


def measured_traffic(op, ne, L, world, launches_per_step):
    """HBM bytes per launch of the operation from the committed PMC passes (FETCH_SIZE / WRITE_SIZE collected in
    separate rocprofv3 runs and corrected as MI355X_MICROARCH.md prescribes; tools/pmc_table.py).  One table per
    workload shape, profiles/r02_pmc_traffic_ne<ne>_L<L>_n<ranks>.json (n > 1: the share of rank 0, taken with a lone
    rank-0 engine of the n-rank layout, tools/rank_share_timing.py); a shape without a table -> None."""
    path, tab = traffic_table(ne, L, world)
    if path is None:
        return None
    try:
        per_step = 0.0
        for name, row in tab["kernels"].items():
            base = kernel_base(name)
            if base in OP_ROCPROF[op]:
                per_step += row["hbm_bytes_per_launch"] * row["launches"] / tab["steps"]
        return per_step / launches_per_step if per_step > 0 and launches_per_step > 0 else None
    except Exception:
        return None


def latest_profile(suffix):
    """Newest committed profiles/rNN_<suffix> (the rounds keep their own files), or None."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return c[-1] if c else None


def mfma_report(on=False):
    """MFMA utilisation of the path that ran (measured offline with rocprofv3 counters) and the A/B of the opt-in matrix-unit
    contractions against the default LDS + VALU ones: the NEWEST committed profiles/rNN_mfma_ab.json."""
    path = latest_profile("mfma_ab.json")
    ab = None
    try:
        ab = json.load(open(path))
        ab["source"] = os.path.relpath(path, ROOT)
    except Exception:
        pass
    active = 0.0
    if on and ab and "mfma_busy_fraction_of_explicit_kernel" in ab.get("TMX_MFMA=1", {}):
        active = ab["TMX_MFMA=1"]["mfma_busy_fraction_of_explicit_kernel"]
    return {"active": active, "ab": ab}


def dominant_kernel_from_stats(ne, L, world):
    """The dominant KERNEL as `rocprofv3 --kernel-trace --stats` ranks them in the committed profile of this workload shape
    (profiles/rNN_kernel_stats*.csv): (operation, kernel name, share) or None when no profile of the shape is committed.  The
    HIP events of this run time whole operations (an explicit stage = several instantiations of k_h_explicit + k_h_w_update),
    so which single kernel dominates is taken from the profile, and its launch time from this run."""
    import csv
    tag = "" if (ne, L, world) == (30, 30, 1) else ("_ne15" if (ne, L, world) == (15, 30, 1) else None)
    if tag is None:
        return None
    path = latest_profile("kernel_stats%s.csv" % tag)
    if not path:
        return None
    try:
        def base_of(r):
            return kernel_base(r["Name"])
        known = {n for names in OP_ROCPROF.values() for n in names}
        rows = [r for r in csv.DictReader(open(path)) if base_of(r) in known]      # the step's kernels (not the memsets / copies of the set-up)
        ok, unknown = profile_kernels_known([base_of(r) for r in rows])
        if not ok:      # a profile of kernels this library no longer holds says nothing about this run
            sys.stderr.write("bench: %s names kernels the loaded library does not hold (%s): not used\n" % (os.path.relpath(path, ROOT), ", ".join(unknown)))
            return None
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        for op, names in OP_ROCPROF.items():
            if base_of(top) in names:
                return op, top["Name"].split("(")[0], float(top["TotalDurationNs"]) / tot, os.path.relpath(path, ROOT)
    except Exception:
        pass
    return None


def smi_snapshot():
    """Clocks / power / temperature of GPU 0 as rocm-smi reports them (strings as printed), or None."""
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--json"], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, universal_newlines=True, timeout=20).stdout
        d = json.loads(out)
        card = d.get("card0", next(iter(d.values())))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "power", "temperature (sensor junction)", "temperature (sensor edge)")):
                keep[k.strip()] = v
        return keep or None
    except Exception:
        return None


def cpu_baseline_mpi(ne, L, dt, nsteps_ref, nranks=6):
    """The same real reference under its own MPI decomposition (one patch per rank, 6 ranks = the reference's minimum
    patch count): reported beside the one-core baseline so that the host comparison does not rest on a single core.
    None when mpiexec or the binary is missing or the run fails."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "tmx_ref_dump")
    mpiexec = next((c for c in ("/opt/conda/bin/mpiexec", "mpiexec") if shutil.which(c)), None)
    if not (os.path.exists(ref_bin) and mpiexec) or (os.cpu_count() or 1) < nranks:
        return None
    try:
        env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
        out = subprocess.run([mpiexec, "-np", str(nranks), ref_bin, "--mode", "time", "--ne", str(ne), "--levels", str(L),
                              "--dt", str(dt), "--steps", str(nsteps_ref), "--warm", "1", "--patches", str(nranks)],
                             env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True, timeout=600).stdout
        import re
        tm = [float(v) for v in re.findall(r"min_s=([0-9.]+)", out)]      # one line per rank (lines of different ranks may interleave)
        if len(tm) < max(1, nranks - 1):
            return None
        t = max(tm)
        return {"value": 6 * ne * ne * 16 * L * 5 / t, "unit": "element-DOF-updates/s", "cores": nranks, "kind": "reference",
                "sample": "real reference binary under mpiexec -np %d (one patch per rank), ne%d L%d ARS343 dt=%gs, slowest rank's "
                          "min of %d steps after 1 warm-up step; %.3f s/step" % (nranks, ne, L, dt, nsteps_ref, t),
                "sypd": dt / t / 365.0}
    except Exception as ex:
        sys.stderr.write("MPI reference baseline failed (%s)\n" % ex)
        return None


def cpu_baseline_allcores(ne, L, dt, nsteps_ref):
    """The reference on a many-patch layout with one MPI rank per patch ('N = all physical cores, stated').
    The layouts the reference accepts are 6 n^2 patches with ne divisible by n; on the 256-thread hosts of the GPU boxes the
    ne30 case runs fastest at 54 ranks (n = 3: 0.32 s per step; 24 ranks 0.34 s, 150 ranks 1.2 s, 216 ranks 1.8 s -- measured,
    the case is too small for more ranks; 96 patches do not divide ne30), so 54 is tried first where the host has the
    threads, then 24."""
    n = os.cpu_count() or 1
    for nr, nn in ((54, 3), (24, 2)):
        if n >= nr and ne % nn == 0:
            b = cpu_baseline_mpi(ne, L, dt, nsteps_ref, nranks=nr)
            if b is not None:
                return b
    return None


def cpu_baseline(ne, L, dt, nsteps_ref, nsteps_port):
    """Host-CPU baseline on rank 0: the real reference if its binary is present, else the C oracle."""
    dof_per_step = 6 * ne * ne * 16 * L * 5
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "tmx_ref_dump")
    if os.path.exists(ref_bin):
        try:
            env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
            out = subprocess.run([ref_bin, "--mode", "time", "--ne", str(ne), "--levels", str(L), "--dt", str(dt),
                                  "--steps", str(nsteps_ref), "--warm", "1", "--patches", "6"],
                                 env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 universal_newlines=True, timeout=900).stdout
            for line in out.splitlines():
                if line.startswith("timing"):
                    f = dict(kv.split("=") for kv in line.split()[1:])
                    tmin = float(f["min_s"])
                    return {"value": dof_per_step / tmin, "unit": "element-DOF-updates/s", "cores": 1,
                            "kind": "reference",
                            "sample": "real reference binary (oracle/_ref), ne%d L%d ARS343 dt=%gs, 1 MPI rank, "
                                      "min of %d steps after 1 warm-up step; %.3f s/step" % (ne, L, dt, nsteps_ref, tmin),
                            "sypd": dt / tmin / 365.0}
        except Exception as ex:  # fall through to the port
            sys.stderr.write("reference baseline failed (%s), using the C oracle\n" % ex)
    import golden_util as gu
    from oracle_lib import Oracle
    g, states = gu.make_grid(ne, L, 6)
    o = Oracle(g)
    o.set_state(0, states)
    o.step_ars343(dt)
    ts = []
    for _ in range(nsteps_port):
        t0 = time.time(); o.step_ars343(dt); ts.append(time.time() - t0)
    tmin = min(ts)
    return {"value": dof_per_step / tmin, "unit": "element-DOF-updates/s", "cores": 1, "kind": "port",
            "sample": "C oracle (oracle/tmx_oracle.c), ne%d L%d ARS343 dt=%gs, 1 thread, min of %d steps; %.3f s/step"
                      % (ne, L, dt, nsteps_port, tmin),
            "sypd": dt / tmin / 365.0}


def projected_from_rank_share(world, ne, L, measured_ms):
    """What this N-rank step was expected to take, from the compute share of rank 0 measured on ONE GPU with the wire left out
    (tools/rank_share_timing.py -> profiles/rNN_rank_shares.json): the share, the share + 6 exchanges at 10 / 30 us of exposed latency,
    and the measured step beside them.  None when no table of this shape is committed."""
    path = latest_profile("rank_shares.json")
    if not path or (ne, L) != (30, 30):
        return None
    try:
        tab = json.load(open(path))
        share = tab["ms_per_step"].get(str(world))
        if share is None:
            return None
        return {"source": os.path.relpath(path, ROOT), "compute_share_ms": share, "kernel_ms_per_step": tab.get("kernel_ms_per_step", {}).get(str(world)),
                "projected_ms_per_step": {"exposed_exchange_10us": share + 0.06, "exposed_exchange_30us": share + 0.18},
                "single_gpu_ms_per_step": tab["ms_per_step"].get("1"), "measured_ms_per_step": measured_ms}
    except Exception:
        return None


def self_launch(n):
    """Run this script as n ranks under torch.distributed.run (127.0.0.1 rendezvous on a free port) and return the launcher's exit code.
    stdout of the ranks passes through (rank 0 prints the one JSON line), stderr too.  Nothing in this process has initialised the GPU."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL / IPC-mapped ghost buffers)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %s\n" % (n, " ".join(cmd[1:9])))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ne", type=int, default=30)
    ap.add_argument("--levels", type=int, default=30)
    ap.add_argument("--dt", type=float, default=150.0)
    ap.add_argument("--ppd", type=int, default=2, help="patches per panel direction (2 -> 24 patches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--develop", type=int, default=10, help="untimed steps before the warm-up (developed state)")
    ap.add_argument("--halo", default=os.environ.get("TMX_HALO", "rccl"), choices=["auto", "rccl", "p2p"],
                    help="multi-rank halo transport: rccl (default) = RCCL send/recv; p2p = peer-to-peer writes into IPC-mapped ghost buffers; "
                         "auto = set up both, check that they deliver identical bytes, time them on this machine and keep the faster "
                         "(falls back to RCCL when the peer-to-peer mapping cannot be set up or disagrees)")
    ap.add_argument("--share-device", action="store_true",
                    help="all ranks on HIP device 0 (1-GPU box; needs --halo p2p, host rendezvous over gloo): exercises the multi-process path, not a scaling number")
    ap.add_argument("--physics", default="none", help="hs: tmx_physics_held_suarez after every step (BASELINE config 5's shape; not the headline metric)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong (default, the headline): the ne30 problem is fixed as --gpus grows.  weak: the grid grows with the rank count, "
                         "ne = the multiple of 2 nearest to --ne * sqrt(gpus) (30 / 42 / 60 / 84 at 1 / 2 / 4 / 8), dt scaled with the element width")
    ap.add_argument("--sustained-steps", type=int, default=2000,
                    help="steps of the untimed sustained leg after the timed region (clock / power before and after are reported; 0: skip)")
    ap.add_argument("--parity-steps", type=int, default=3,
                    help="N > 1: after the timing, all ranks and a single-rank engine on rank 0 step this many times from the initial state and "
                         "rank 0's patches are compared bit for bit (0: skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # `python bench.py --gpus N` without a launcher: start the ranks ourselves, as a CHILD process, before anything here has
            # touched the GPU (no exec of a process that initialised HIP), and relay rank 0's JSON line and the launcher's exit code
            sys.exit(self_launch(args.gpus))
        sys.exit("bench.py --gpus %d inside a world of %d ranks: --gpus must equal WORLD_SIZE (one rank per GPU)" % (args.gpus, world))
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the engine has no CPU path")
    if args.share_device:
        if world > 1 and args.halo == "rccl":
            sys.exit("--share-device needs --halo p2p (RCCL refuses two ranks on one device)")
        # TMX_BENCH_AUTO_SELFTEST=1 walks the start-up comparison of --halo auto on a 1-GPU box: "RCCL" is then the
        # peer-to-peer transport again (there is no second device to give RCCL), so only the bench's own logic is exercised
        args.halo = "auto" if os.environ.get("TMX_BENCH_AUTO_SELFTEST") == "1" and args.halo == "auto" else "p2p"
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.share_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import golden_util as gu
    from tempestmodel_amd.cubed_sphere import CubedSphereGrid, BaroclinicWaveJW
    from tempestmodel_amd.engine import Engine

    ne, L, dt = args.ne, args.levels, args.dt
    if args.scaling == "weak" and world > 1:
        ne = 2 * int(round(args.ne * (world ** 0.5) / 2.0))
        dt = args.dt * args.ne / ne
    grid = CubedSphereGrid(ne, L, 30000.0, ppd=args.ppd)
    states = grid.evaluate_test_case(BaroclinicWaveJW(0.0, 30000.0, "exp"))
    eng = Engine(grid, rank=rank, n_ranks=world, device=local_rank)
    halo_note = "RCCL send/recv"
    if world > 1:
        hdev = "cpu" if args.share_device else "cuda"

        def agree(flag):          # 1 only if every rank says 1
            t = torch.tensor([int(flag)], dtype=torch.int32, device=hdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item()) == 1

        if args.halo in ("rccl", "auto") and not args.share_device:
            if rank == 0:
                uid = torch.tensor(list(Engine.comm_unique_id()), dtype=torch.uint8, device="cuda")
            else:
                uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            dist.broadcast(uid, 0)
            eng.comm_init(bytes(uid.cpu().tolist()))
        p2p_ok, why = False, ""
        if args.halo in ("p2p", "auto"):
            try:
                blob = eng.halo_p2p_export()
            except Exception as ex:
                blob, why = None, str(ex)
            blobs = [None] * world
            dist.all_gather_object(blobs, blob)
            ok = all(b is not None for b in blobs)
            if ok:
                try:
                    eng.halo_p2p_connect(blobs)
                except Exception as ex:
                    ok, why = False, str(ex)
            p2p_ok = agree(ok)
            dist.barrier()
            if args.halo == "p2p" and not p2p_ok:
                sys.exit("--halo p2p: the peer-to-peer mapping could not be set up on every rank (%s)" % why)
            if p2p_ok:
                halo_note = "peer-to-peer writes into IPC-mapped ghost buffers"
            elif ok:
                eng.halo_p2p_enable(False)
    eng.upload_state(0, states)
    if world > 1 and args.halo == "auto" and p2p_ok:
        # both transports are up: same bytes?  which is faster here?  (instance 1 is scratch until the first step)
        import numpy as np

        enable = (lambda on: eng.halo_p2p_enable(True)) if args.share_device else eng.halo_p2p_enable

        def dss_result(p2p):
            enable(p2p)
            eng.copy_data(0, 1)
            eng.apply_dss(1)
            eng.sync()
            got = eng.download_state(1)
            return [got[p] for p in eng.local_patches]

        def dss_time(p2p, n=20):
            enable(p2p)
            for _ in range(3):
                eng.apply_dss(1)
            eng.sync(); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                eng.apply_dss(1)
            eng.sync(); dist.barrier()
            t = torch.tensor([(time.perf_counter() - t0) / n], dtype=torch.float64, device=hdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        # every leg is entered by all ranks or by none: a rank that failed in the peer-to-peer leg (e.g. a neighbour's message
        # never became visible: TMX_ERR_COMM after the wait kernel's time-out) must not leave the others alone in the RCCL leg
        same = True
        try:
            a = dss_result(True)
        except Exception as ex:
            same, why = False, str(ex)
        if agree(same):
            b = dss_result(False)
            same = all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a, b))
            same = agree(same)
            if not same:
                why = "different bytes"
        else:
            same = False
        if not same:
            enable(False)
            halo_note = "RCCL send/recv (auto: the peer-to-peer transport was set up but did not reproduce RCCL's bytes: %s)" % why
            if rank == 0:
                sys.stderr.write("bench: peer-to-peer halo disagrees with RCCL, using RCCL\n")
        else:
            t_p2p, t_rccl = dss_time(True), dss_time(False)
            use_p2p = t_p2p <= t_rccl
            enable(use_p2p)
            halo_note = "%s (auto: one DSS incl. exchange takes %.1f us peer-to-peer, %.1f us with RCCL send/recv; identical bytes%s)" % (
                "peer-to-peer writes into IPC-mapped ghost buffers" if use_p2p else "RCCL send/recv", 1e6 * t_p2p, 1e6 * t_rccl,
                "; SELF-TEST of this comparison on one device: both legs are the peer-to-peer transport" if args.share_device else "")
    elif world > 1 and args.halo == "auto":
        halo_note = "RCCL send/recv (auto: peer-to-peer mapping not available: %s)" % why[:200]
    eng.upload_state(0, states)

    hs = (args.physics == "hs")
    if hs:
        eng.set_physics_inputs(None)
    _step = eng.step_ars343

    def step_with_forcing(dt_):
        _step(dt_)
        eng.held_suarez(0, dt_)
    if hs:
        eng.step_ars343 = step_with_forcing

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    for _ in range(args.develop):       # developed state: W != 0, realistic pivot pattern (not part of W or K)
        eng.step_ars343(dt)
    for _ in range(args.warmup):
        eng.step_ars343(dt)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step_ars343(dt)
    barrier()
    wall = time.perf_counter() - t0
    unique_after_timed = eng.info(13)      # data instances in node-unique form when the timed region ended
    # per-kernel device time: the same K steps again with HIP events around every launch on the
    # engine's stream (kept out of the timed region: the event markers cost a few % of wall time)
    eng.profile_enable(True)
    eng.profile_reset()
    eng.pivot_stats(True)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        eng.step_ars343(dt)
    barrier()
    wall_prof = time.perf_counter() - t1
    prof = eng.profile()
    pivot = eng.pivot_stats(False)
    eng.profile_enable(False)
    if world > 1:
        tt = torch.tensor([wall], dtype=torch.float64, device="cpu" if args.share_device else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())

    # sustained leg (untimed for `value`): enough steps for clocks and power to settle, with the SMI readings on both sides
    sustained = None
    if args.sustained_steps > 0:
        smi0 = smi_snapshot() if rank == 0 else None
        barrier()
        ts0 = time.perf_counter()
        for _ in range(args.sustained_steps):
            eng.step_ars343(dt)
        barrier()
        ts = time.perf_counter() - ts0
        if world > 1:
            tt = torch.tensor([ts], dtype=torch.float64, device="cpu" if args.share_device else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ts = float(tt.item())
        sustained = {"steps": args.sustained_steps, "seconds": ts, "ms_per_step": 1e3 * ts / args.sustained_steps,
                     "smi_before": smi0, "smi_after": smi_snapshot() if rank == 0 else None}

    # sanity: the state the timed and sustained legs produced must still be finite (checked BEFORE the parity leg re-uploads the initial state)
    import numpy as np
    chk = eng.download_state(0)
    state_finite = all(np.isfinite(chk[p][0]).all() and np.isfinite(chk[p][1][3]).all() for p in eng.local_patches)
    del chk

    # N > 1: the line carries its own parity proof -- every rank steps again from the initial state, rank 0 also steps a
    # single-rank engine holding all patches, and rank 0's patches must come out identical, bit for bit
    parity = None
    if world > 1 and args.parity_steps > 0 and not hs:
        eng.upload_state(0, states)
        for _ in range(args.parity_steps):
            _step(dt)
        barrier()
        got = eng.download_state(0)
        if rank == 0:
            one = Engine(grid, rank=0, n_ranks=1, device=local_rank)
            one.upload_state(0, states)
            for _ in range(args.parity_steps):
                one.step_ars343(dt)
            one.sync()
            ref = one.download_state(0)
            same = all(np.array_equal(got[p][0], ref[p][0]) and np.array_equal(got[p][1], ref[p][1]) for p in eng.local_patches)
            worst = max(float(np.max(np.abs(got[p][0] - ref[p][0]))) for p in eng.local_patches)
            parity = {"steps": args.parity_steps, "patches_compared": len(eng.local_patches), "identical_to_single_rank_engine": bool(same),
                      "max_abs_difference": worst}
            one.close()
        barrier()

    if parity is not None or (world > 1 and args.parity_steps > 0 and not hs):      # ... and the state behind the parity leg as well
        chk = eng.download_state(0)
        state_finite = state_finite and all(np.isfinite(chk[p][0]).all() and np.isfinite(chk[p][1][3]).all() for p in eng.local_patches)
    ablation_build = bool(os.environ.get("TMX_LIBRARY"))      # tools/ablate.sh builds skip parts of the arithmetic on purpose
    if not state_finite and not ablation_build:
        sys.exit("bench.py: state became non-finite")

    # N > 1: every rank's kernel split travels to rank 0 (the first multi-GPU record should explain itself)
    rank_kernels = None
    if world > 1:
        mine = {k: v[0] / args.steps for k, v in prof.items()}
        rank_kernels = [None] * world
        dist.all_gather_object(rank_kernels, mine)

    if rank == 0:
        n_el = 6 * ne * ne
        dof_per_step = n_el * 16 * L * 5
        points = 6 * (ne * 4) ** 2 * L
        ms_per_step = 1e3 * wall / args.steps
        value = dof_per_step * args.steps / wall
        # dominant operation of the step on this rank; a launch = one call of the operation (its kernels together)
        op_time = {op: sum(prof[k][0] for k in ks) for op, ks in OP_KERNELS.items()}
        # The roofline object is for the dominant KERNEL as `rocprofv3 --stats` ranks them.  The HIP events of this run time whole
        # operations (the explicit stage = several instantiations of k_h_explicit + k_h_w_update), so the kernel ranking is read
        # from the committed kernel-trace profile of this workload shape; without one the top operation is taken unscaled.
        dk = dominant_kernel_from_stats(ne, L, world)
        if dk is not None:
            dom, dom_note = dk[0], "dominant kernel per %s: %s (%.1f %% of kernel time there)" % (dk[3], dk[1], 100.0 * dk[2])
        else:
            dom = max((op for op in op_time if ALG_BYTES[op] > 0), key=lambda op: op_time[op])
            dom_note = "no committed kernel-trace profile of this shape: the operation with the largest summed time of this run (several kernels)"
        op_launches = {op: max(sum(prof[k][1] for k in ks), 1) for op, ks in OP_KERNELS.items()}
        avg_ms = op_time[dom] / op_launches[dom]
        local_points = points / world
        # headline `achieved` / `frac`: SURVEY 8(d)'s row x the points the kernel really PROCESSES -- unique (node, level) points where the
        # node-unique layout ran or the operation is the column solve, stored points otherwise; the same row on the stored points (rounds
        # 1-5's headline) stays beside it as frac_8d_row_on_stored_points
        proc_points = (eng.info(1) * L) if (eng.info(12) == 1 and unique_after_timed > 0) or dom == "v_implicit" else local_points
        achieved = ALG_BYTES[dom] * proc_points / (avg_ms * 1e-3) / 1e9
        achieved_stored = ALG_BYTES[dom] * local_points / (avg_ms * 1e-3) / 1e9
        op_share = {op: op_time[op] / max(sum(op_time.values()), 1e-12) for op in op_time}
        dom_op = max(op_share, key=lambda op: op_share[op])
        walk = eng.get_option("h_walk") != 0 and eng.info(12) == 1 and unique_after_timed > 0
        kernel_name = {"h_explicit": "k_h_walk<NT> (column-segment walk)" if walk else "k_h_explicit<true,NT>+k_h_w_update<NT>", "v_implicit": "k_vi_pair (two wavefronts per column group)",
                       "dss": "k_dss", "hypervis": "k_hypervis", "lincomb": "k_lincomb"}[dom]
        if dom == "v_implicit" and eng.info(20) == 2:      # TMX_INFO_COLUMN_KERNEL: the kernel that really ran
            kernel_name = "k_vi_group (one column per 16-lane group)"
        mfma = mfma_report(eng.get_option("contraction_mfma") != 0)
        # second accounting, for the layout that ran (layout_model above) and the measured HBM rate = counter traffic / this run's time
        unique_ran = eng.info(12) == 1 and unique_after_timed > 0
        lay = layout_model(eng, L, unique_ran)
        lay_dom = lay[dom]
        dom_traffic = measured_traffic(dom, ne, L, world, op_launches[dom] / args.steps)
        tpath, ttab = traffic_table(ne, L, world)
        step_traffic = None
        if ttab is not None:
            known = {n for names in OP_ROCPROF.values() for n in names}
            step_traffic = sum(r["hbm_bytes_per_launch"] * r["launches"] / ttab["steps"] for k, r in ttab["kernels"].items()
                               if kernel_base(k) in known)
        sha_now = csrc_sha16()
        sha_prof = (ttab or {}).get("csrc_sha16")

        def per_op(op):
            # algorithmic figure (SURVEY 8d row x points) beside the counter-measured one: the 8d rows are not lower bounds for
            # fused passes (hypervis reads less than its row says), so "measured_GBs" is the rate to judge those by
            ms = op_time[op] / op_launches[op]
            tr = measured_traffic(op, ne, L, world, op_launches[op] / args.steps)
            extra = {}
            if op == "v_implicit":
                # what the column solve cannot avoid moving as designed (unique columns read once, the three solved variables
                # written to every stored column): the SURVEY row above also counts 26 metric doubles this kernel no longer
                # reads, and the measured traffic is mostly the U-row stream of the band LU going to HBM and back
                extra["compulsory_bytes_per_launch"] = 8.0 * ((5 * L + 1) * eng.info(1) + (3 * L + 1) * eng.info(0))
            return {"launches_per_step": op_launches[op] / args.steps, "ms_per_launch": ms, **extra,
                    "achieved_GBs": ALG_BYTES[op] * local_points / max(ms * 1e-3, 1e-12) / 1e9,
                    "bytes_per_point": ALG_BYTES[op], "survey_8d_bytes_per_point": SURVEY_8D_BYTES[op],
                    "achieved_GBs_with_survey_8d_row": SURVEY_8D_BYTES[op] * local_points / max(ms * 1e-3, 1e-12) / 1e9,
                    "traffic": tr, "measured_GBs": (tr / max(ms * 1e-3, 1e-12) / 1e9) if tr else None,
                    "layout_aware_bytes_per_launch": lay[op], "layout_aware_GBs": lay[op] / max(ms * 1e-3, 1e-12) / 1e9,
                    "bound": OP_BOUND[op]}
        out = {
            "metric": "element-DOF-updates/s (JW baroclinic wave ne%d np4 L%d, ARS343 IMEX)" % (ne, L),
            "value": value, "unit": "element-DOF-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "sypd": dt / (wall / args.steps) / 365.0,
            "config": {"workload": "Jablonowski-Williamson baroclinic wave, cubed sphere ne%d np4 L%d, ztop 30 km, "
                                   "ARS343 dt=%gs, hyperviscosity order 4, %d patches" % (ne, L, dt, len(grid.patches)),
                       "elements": n_el, "dof_per_step": dof_per_step, "dt_s": dt,
                       "parallelism": "%d patches over %d %s, halo exchange = %s" % (
                           len(grid.patches), world, "processes sharing ONE GPU (multi-process path check, not a scaling number)" if args.share_device and world > 1 else "GPU(s)",
                           halo_note),
                       "hbm_allocated_GiB_per_gpu": eng.info(5) / 2.0 ** 30,
                       # the sustained leg beside the K timed steps (full record: top-level "sustained")
                       "sustained_steps": sustained["steps"] if sustained else 0,
                       "sustained_ms_per_step": sustained["ms_per_step"] if sustained else None,
                       "sustained_value": (dof_per_step / (sustained["ms_per_step"] * 1e-3)) if sustained else None,
                       # tmx_set_option("unique_layout") / TMX_UNIQUE: inside tmx_step the state is held once per patch node
                       "state_layout": ("node-unique inside tmx_step (%d instances in that form at the end of the timed region, %d partial slots, "
                                        "%d node groups left to the DSS kernel of %d)" % (unique_after_timed, eng.info(15), eng.info(16), eng.info(2)))
                                       if eng.info(12) == 1 else "element-major (16 nodes per element, seam nodes duplicated)"},
            # `bound`: the roofline `frac` is priced against (HBM: every kernel of the path is fp64 stencil / band work, no GEMM); `limiter`: what
            # limits the dominant operation as measured ("hbm", or "latency" for the column solve: dependent fp64 chains of its two co-scheduled
            # wavefronts beside the U-row stream, see `limited_by`)
            "roofline": {"bound": "hbm", "limiter": "hbm" if OP_BOUND[dom].startswith("hbm") else "latency", "priced_against": "hbm",
                         "kernel": kernel_name, "operation": dom, "dominant_by": dom_note,
                         "dominant_operation": dom_op, "operation_share_of_kernel_time": {op: round(v, 4) for op, v in op_share.items()},
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "points_processed_per_launch": proc_points, "points_stored": local_points,
                         "achieved_8d_row_on_stored_points": achieved_stored, "frac_8d_row_on_stored_points": achieved_stored / HBM_PEAK_GBS,
                         "traffic": dom_traffic, "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": ALG_BYTES[dom] * proc_points,
                         # (1) the same 8(d) row on the points this layout holds (the column solve works on the unique columns, not the stored ones)
                         "achieved_8d_row_on_layout_points": (ALG_BYTES[dom] * (eng.info(1) if (unique_ran or dom == "v_implicit") else eng.info(0)) * L) / (avg_ms * 1e-3) / 1e9,
                         "frac_8d_row_on_layout_points": (ALG_BYTES[dom] * (eng.info(1) if (unique_ran or dom == "v_implicit") else eng.info(0)) * L) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         # (2) layout-aware algorithmic bytes: what the operation cannot avoid moving in the layout that runs (layout_model)
                         "layout_aware_bytes_per_launch": lay_dom, "achieved_layout_aware": lay_dom / (avg_ms * 1e-3) / 1e9,
                         "frac_layout_aware": lay_dom / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         # (3) the HBM rate itself: counter traffic of the committed PMC table / this run's launch time
                         "hbm_rate_GBs": (dom_traffic / (avg_ms * 1e-3) / 1e9) if dom_traffic else None,
                         "hbm_rate_frac": (dom_traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if dom_traffic else None,
                         "traffic_over_layout_aware_bytes": (dom_traffic / lay_dom) if (dom_traffic and lay_dom) else None,
                         "whole_step_traffic_bytes": step_traffic,
                         "whole_step_hbm_rate_GBs": (step_traffic / (ms_per_step * 1e-3) / 1e9) if step_traffic else None,
                         "whole_step_hbm_rate_frac": (step_traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_traffic else None,
                         "whole_step_layout_aware_GBs": sum(lay[op] * op_launches[op] / args.steps for op in lay) / (ms_per_step * 1e-3) / 1e9,
                         "traffic_source": os.path.relpath(tpath, ROOT) if tpath else None,
                         "traffic_source_csrc_sha16": sha_prof, "library_csrc_sha16": sha_now,
                         "traffic_source_matches_library_sources": (sha_prof == sha_now) if sha_prof else None,
                         "per_operation": {op: per_op(op) for op in OP_KERNELS},
                         "limited_by": OP_BOUND[dom],
                         "uniform_pivot_fraction": (pivot[0] / pivot[1]) if pivot[1] else None,
                         "whole_step_algorithmic_GBs": ALG_BYTES_PER_STEP * points * args.steps / wall / 1e9,
                         # default path: no MFMA instruction is issued (the bit-exact contractions are LDS + VALU); with TMX_MFMA=1
                         # the fused explicit kernel issues 12 v_mfma_f64_4x4x4_4b per point: counters in profiles/r02_mfma_ab.json
                         "mfma_utilisation": mfma["active"], "mfma_ab": mfma["ab"],
                         "offline_sources": "traffic, mfma_utilisation and mfma_ab are read from the committed rocprofv3 passes "
                                            "(profiles/%s_*), not measured by this run; times and rates are this run's" % PROFILE_TAG,
                         },
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items()},
            "ms_per_step_with_event_markers": 1e3 * wall_prof / args.steps,
            "developed_steps": args.develop, "host_cpu_count": os.cpu_count(),
            "physics_after_every_step": args.physics,
            "state_finite": bool(state_finite),
            "sustained": sustained,
            # every engine option as this run used it (the library reads no environment variable on its own; what this script's
            # plumbing took from TMX_* variables is listed under from_environment)
            "engine_options": eng.options_report(),
        }
        if world > 1:
            send = eng.plan(0).reshape(-1, 4)
            per_peer = {}
            for r_ in send[:, 3]:
                per_peer[int(r_)] = per_peer.get(int(r_), 0) + 1
            nslab = 5 * L + 1
            out["multi_rank"] = {"rccl_comm_ranks": eng.info(9), "halo_transport": {0: "none", 1: "rccl", 2: "p2p", 3: "loopback"}.get(eng.info(11), "?"),
                                 "rank0_message_bytes_per_peer": {str(k): v * nslab * 8 for k, v in sorted(per_peer.items())},
                                 "rank0_early_tiles": eng.info(7), "rank0_late_tiles": eng.info(8),
                                 "parity": parity,
                                 "per_rank_kernel_ms_per_step": rank_kernels,
                                 "projected_from_rank_share": projected_from_rank_share(world, ne, L, ms_per_step),
                                 "scaling_note": "strong scaling of a ~1.2 ms step is Amdahl-limited: the per-rank compute share measured on one GPU (wire left out) "
                                                 "bounds the speed-up before any message is sent; weak scaling (--scaling weak: the grid grows with the rank count) is "
                                                 "the regime the design is for (11.2-12.8 G element-DOF-updates/s per GPU from ne30 to ne84, profiles/r06_other_shapes.txt)"}
        if ablation_build:      # a library named by TMX_LIBRARY: never a result, say so in the line itself
            out["ablation_build"] = True
            out["library"] = os.environ["TMX_LIBRARY"]
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(ne, L, dt, 6, 6)      # ~15 s of CPU work (1 warm-up + 6 timed steps)
            mpi = cpu_baseline_mpi(ne, L, dt, 6)                     # + the reference's own 6-rank MPI run, ~20 core-seconds
            if mpi is not None:
                out["cpu_baseline_mpi"] = mpi
            allc = cpu_baseline_allcores(ne, L, dt, 4)              # + 24 ranks on the 24-patch layout where the host has them
            if allc is not None:
                out["cpu_baseline_allcores"] = allc
                # the fairest host figure travels INSIDE the cpu_baseline object as well (its headline fields stay the one-core run)
                out["cpu_baseline"]["all_cores"] = {k: allc[k] for k in ("value", "unit", "cores", "kind", "sample") if k in allc}
            if mpi is not None:
                out["cpu_baseline"]["mpi_6_ranks"] = {k: mpi[k] for k in ("value", "unit", "cores", "kind", "sample") if k in mpi}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()        # peer-to-peer transport: neighbours write into this rank's block until they are done
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
