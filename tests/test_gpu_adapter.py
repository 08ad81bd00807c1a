"""End-to-end drop-in check on the GPU: the REAL reference (Model, Grid, BaroclinicWaveJWTest,
TimestepSchemeARS343, ApplyDSS ... compiled from /root/reference into oracle/_ref) running with the
adapter plugins of adapter/TempestMI355X.h, compared inside one process with the pure reference.

The binary is built in the build container (`make -C oracle ref`) and travels with the snapshot;
the test is skipped where it is absent."""
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "tmx_adapter_run")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu      # noqa: E402
TOL = "1e-300" if gu.exact_tolerance() == 0.0 else "1e-10"      # identical doubles where the host's libm is the one tmx_refmath.h restates


@pytest.mark.parametrize("mode,args", [
    ("production", ["--ne", "4", "--levels", "10", "--steps", "5", "--patches", "6"]),
    ("production", ["--ne", "4", "--levels", "6", "--steps", "3", "--patches", "24"]),
    ("parity", ["--ne", "4", "--levels", "10", "--steps", "3", "--patches", "6"]),
    # element counts that are not powers of two: the reference's per-patch element spacing (a difference of edge
    # coordinates, GridPatchGLL.cpp:67-75) is not pi / (2 ne) there, and with 24 patches it need not be one value
    ("production", ["--ne", "6", "--levels", "8", "--steps", "3", "--patches", "24"]),
    ("production", ["--ne", "5", "--levels", "6", "--steps", "3", "--patches", "6"]),
    ("production", ["--ne", "10", "--levels", "6", "--steps", "2", "--patches", "24"]),
    # the reference's ARK232 over the plugin classes: StepImplicitTermsExplicitly and the sub-cycled explicit stages
    ("parity", ["--scheme", "ark232", "--ne", "3", "--levels", "6", "--steps", "2", "--patches", "6"]),
    # state kept on the device between steps (SetHostSyncInterval), downloaded once at the end
    ("production", ["--ne", "4", "--levels", "10", "--steps", "5", "--patches", "6", "--hostsync", "100"]),
    # row H7: Schar mountain on the reduced-radius sphere, Rayleigh layer (dt 0.5 s)
    ("production", ["--case", "schar", "--ne", "3", "--levels", "6", "--steps", "3", "--patches", "6", "--dt", "0.5"]),
    # rows H8 / V8: two tracer densities
    ("production", ["--tracers", "2", "--ne", "3", "--levels", "6", "--steps", "3", "--patches", "6"]),
    ("parity", ["--tracers", "2", "--ne", "3", "--levels", "6", "--steps", "2", "--patches", "6"]),
    # Held-Suarez forcing as a WorkflowProcess after every step (HeldSuarezPhysicsMI355X)
    ("production", ["--physics", "1", "--ne", "2", "--levels", "30", "--steps", "3", "--patches", "6"]),
    # BASELINE config 4 dynamics: planet / 120, uniform diffusion, --explicitvertical, two tracers (dt 1 s)
    ("production", ["--case", "supercell", "--tracers", "2", "--ne", "3", "--levels", "6", "--steps", "3", "--patches", "6", "--dt", "1"]),
    ("parity", ["--case", "supercell", "--tracers", "2", "--ne", "3", "--levels", "6", "--steps", "2", "--patches", "6", "--dt", "1"]),
    # the other steppers as production drop-ins (TimestepScheme*_MI355X)
    ("production", ["--scheme", "ars232", "--ne", "3", "--levels", "6", "--steps", "3", "--patches", "6"]),
    ("production", ["--scheme", "ars222", "--ne", "3", "--levels", "6", "--steps", "3", "--patches", "6"]),
    ("production", ["--scheme", "ars443", "--ne", "3", "--levels", "6", "--steps", "3", "--patches", "6"]),
    ("production", ["--scheme", "ark232", "--ne", "3", "--levels", "6", "--steps", "3", "--patches", "6"]),
    ("production", ["--scheme", "ark232", "--tracers", "2", "--ne", "2", "--levels", "5", "--steps", "3", "--patches", "6", "--dt", "300"]),
    ("parity", ["--scheme", "ark232", "--tracers", "2", "--ne", "2", "--levels", "5", "--steps", "2", "--patches", "6", "--dt", "300"]),
    ("production", ["--scheme", "strang", "--ne", "3", "--levels", "6", "--steps", "3", "--patches", "6"]),
    ("production", ["--restart", "2", "--ne", "3", "--levels", "6", "--steps", "4", "--patches", "6", "--tracers", "2"]),
    ("production", ["--restart", "2", "--case", "heldsuarez", "--physics", "1", "--ne", "2", "--levels", "12", "--steps", "4", "--patches", "6"]),
    ("production", ["--scheme", "strang_rk4", "--offcentering", "0.25", "--ne", "2", "--levels", "4", "--steps", "3", "--patches", "6", "--dt", "400"]),
    ("production", ["--scheme", "strang_ssprk53", "--ne", "2", "--levels", "4", "--steps", "3", "--patches", "6", "--dt", "400"]),
    ("production", ["--scheme", "strang_ssp3", "--ne", "2", "--levels", "4", "--steps", "2", "--patches", "6", "--dt", "400"]),
    ("production", ["--scheme", "strang_fe", "--ne", "2", "--levels", "4", "--steps", "2", "--patches", "6", "--dt", "50"]),
    # BASELINE config 1: the stock SWTest2 (shallow water, Strang / KGU35) drops in unchanged
    ("production", ["--case", "sw2", "--ne", "8", "--steps", "5", "--patches", "6"]),
    # BASELINE config 5 shape: the stock HeldSuarezTest at L60, forcing on the device after every step; then with the
    # state resident on the device across steps AND forcing calls (host refreshed once at the end)
    ("production", ["--case", "heldsuarez", "--ne", "3", "--levels", "60", "--steps", "4", "--warm", "2", "--patches", "6"]),
    ("production", ["--case", "heldsuarez", "--ne", "3", "--levels", "60", "--steps", "4", "--warm", "2", "--patches", "6", "--hostsync", "100"]),
    ("production", ["--physics", "1", "--ne", "2", "--levels", "30", "--steps", "3", "--patches", "6", "--hostsync", "2"]),
    # BASELINE config 4: the stock SupercellTest (Fortran initial state) + KesslerPhysics, L40, --explicitvertical
    ("production", ["--case", "supercellreal", "--ne", "2", "--levels", "40", "--dt", "0.5", "--steps", "3", "--warm", "2", "--patches", "6"]),
    ("production", ["--case", "supercellreal", "--ne", "2", "--levels", "40", "--dt", "0.5", "--steps", "3", "--warm", "2", "--patches", "6", "--hostsync", "100"]),
    ("parity", ["--case", "supercellreal", "--ne", "2", "--levels", "40", "--dt", "0.5", "--steps", "2", "--warm", "2", "--patches", "6"]),
    # Model::Go's own sequence (Step with fLastStep on the last step, then the WorkflowProcess, then the host state is read
    # with NO manual flush): with a deferred host copy the physics plugin that runs after a flushing Step must flush again,
    # the Kessler precipitation included (flush hook); intervals 1, 2 (flush in the middle and at the end) and 100 (last step only)
    ("production", ["--modelgo", "1", "--case", "heldsuarez", "--ne", "2", "--levels", "12", "--steps", "4", "--warm", "2", "--patches", "6", "--hostsync", "100"]),
    ("production", ["--modelgo", "1", "--case", "heldsuarez", "--ne", "2", "--levels", "12", "--steps", "4", "--warm", "2", "--patches", "6", "--hostsync", "2"]),
    ("production", ["--modelgo", "1", "--case", "supercellreal", "--ne", "2", "--levels", "40", "--dt", "0.5", "--steps", "3", "--warm", "2", "--patches", "6", "--hostsync", "100"]),
    ("production", ["--modelgo", "1", "--case", "supercellreal", "--ne", "2", "--levels", "40", "--dt", "0.5", "--steps", "4", "--warm", "2", "--patches", "6", "--hostsync", "2"]),
    ("production", ["--modelgo", "1", "--ne", "3", "--levels", "6", "--steps", "3", "--patches", "6", "--hostsync", "1"]),
])
def test_reference_driver_with_adapter_plugins(mode, args):
    gu.need_live_reference(BIN)
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL")
    # the plugins see the reference's own grid objects, so the device reproduces the reference BIT FOR BIT: the tolerance
    # only has to let an exact zero through (worst < tol)
    r = subprocess.run([BIN, "--mode", mode, "--dt", "200", "--warm", "1", "--tol", TOL] + args,     # a later --dt / --warm overrides
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    tail = "\n".join(r.stdout.splitlines()[-6:])
    print(tail)
    assert r.returncode == 0 and "ADAPTER_OK" in r.stdout, tail
    # the metric factors read from the reference's own grid objects reproduce its arrays bit for bit
    if "sw2" not in args:       # the shallow-water set has no 3-D metric
        assert "verified bit for bit): active" in r.stdout, tail


@pytest.mark.parametrize("nranks,args", [
    (2, ["--ne", "4", "--levels", "6", "--steps", "3", "--patches", "24"]),
    (4, ["--tracers", "2", "--ne", "4", "--levels", "6", "--steps", "3", "--patches", "24"]),
    (3, ["--scheme", "strang", "--ne", "4", "--levels", "6", "--steps", "3", "--patches", "6"]),
])
def test_reference_driver_under_mpi_with_peer_to_peer_halo(nranks, args):
    """The reference's own MPI run (mpiexec, its patches spread over the ranks, its exchange over MPI) against the adapter in
    the same processes with the halo on the device (TMX_HALO=p2p: the ranks share the one GPU of the box, their engines
    exchange through IPC-mapped ghost buffers): every rank reports a zero difference on the patches it owns."""
    import shutil
    mpiexec = next((c for c in ("/opt/conda/bin/mpiexec", "mpiexec") if shutil.which(c)), None)
    gu.need_live_reference(BIN)
    if mpiexec is None:
        pytest.skip("needs mpiexec")
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", TMX_HALO="p2p", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([mpiexec, "-np", str(nranks), BIN, "--mode", "production", "--dt", "200", "--warm", "1", "--tol", TOL] + args,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    tail = "\n".join(r.stdout.splitlines()[-24:])
    print(tail)
    assert r.returncode == 0 and r.stdout.count("ADAPTER_OK") == nranks and "ADAPTER_MISMATCH" not in r.stdout, tail
