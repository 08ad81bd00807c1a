"""The host logic of the library -- plan tables, node-unique layout tables, stepper programs, option plumbing, ~4 000 lines that run before
any kernel does -- under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C tempestmodel_amd/csrc san`: the host side of every
translation unit instrumented, device code as in production; GPU sanitizers are not available on this pool).  CPU only: plan-only engines
(cfg.device = -2), odd layouts included (54 / 96 patches over 3 / 5 / 7 ranks)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tempestmodel_amd", "csrc")
SAN = os.path.join(ROOT, "tempestmodel_amd", "libtempest_mi355x_san.so")


def test_host_logic_under_address_and_undefined_behaviour_sanitizers():
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    assert rt, "the sanitizer runtime of the ROCm clang is missing"
    subprocess.run(["make", "-C", CSRC, "-j8", "san"], check=True, stdout=subprocess.DEVNULL)      # (stale objects only)
    env = dict(os.environ, LD_PRELOAD=rt[-1], TMX_LIBRARY=SAN, TMX_QUIET="1",
               ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    # every host-logic test that needs no second library flavour and no compiler (the preloaded runtime would instrument gcc's children too)
    skip = "not exports_every_declared and not options_are_explicit and not refmath_is_glibc"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_abi_and_host_logic.py"), "-x", "-q", "-p", "no:cacheprovider", "-k", skip],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=1500)
    tail = r.stdout[-4000:]
    assert "AddressSanitizer" not in r.stdout and "runtime error:" not in r.stdout, tail
    assert r.returncode == 0 and " passed" in r.stdout, tail
