#!/usr/bin/env python3
"""Host <-> device state transfer time at the drop-in boundary (tmx_upload_state / tmx_download_state: the reference's
[var][A+2][B+2][L] arrays <-> slab layout), whole instance, ne30 L30, 24 patches.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from tempestmodel_amd.cubed_sphere import CubedSphereGrid, BaroclinicWaveJW
from tempestmodel_amd.engine import Engine

ne = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = CubedSphereGrid(ne, 30, 30000.0, ppd=2)
st = g.evaluate_test_case(BaroclinicWaveJW(0.0, 30000.0, "exp"))
e = Engine(g)
nbytes = sum(n.nbytes + r.nbytes for n, r in st)
for name, fn in (("upload", lambda: e.upload_state(0, st)), ("download", lambda: e.download_state(0))):
    fn(); e.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    e.sync()
    dt = (time.perf_counter() - t0) / 3
    print("%s: %.1f ms per instance (%.0f MB host arrays incl. halo, %.2f GB/s)" % (name, dt * 1e3, nbytes / 1e6, nbytes / dt / 1e9))
# the C calls alone, into preallocated host arrays (what the adapter does with the reference's own DataArray4D storage)
import ctypes as C
PD = C.POINTER(C.c_double)
outs = [(np.zeros_like(n), np.zeros_like(r)) for n, r in st]
for name in ("tmx_upload_state", "tmx_download_state"):
    fn = getattr(e.lib, name)
    def call():
        for p, (n, r) in enumerate(outs if name == "tmx_download_state" else st):
            assert fn(e.h, p, 0, n.ctypes.data_as(PD), r.ctypes.data_as(PD)) == 0
    call(); e.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        call()
    e.sync()
    dt = (time.perf_counter() - t0) / 3
    print("%s (C calls only): %.1f ms per instance, %.2f GB/s" % (name, dt * 1e3, nbytes / dt / 1e9))
# the restart image: transposition on the device, one contiguous copy per patch (tmx_pack_active_state / tmx_unpack_active_state)
imgs = [np.zeros(e.lib.tmx_active_state_bytes(e.h, p), dtype=np.uint8) for p in range(len(st))]
VP = C.c_void_p
for name in ("tmx_pack_active_state", "tmx_unpack_active_state"):
    fn = getattr(e.lib, name)
    def call():
        for p, im in enumerate(imgs):
            assert fn(e.h, p, 0, im.ctypes.data_as(VP), im.size) == 0
    call(); e.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        call()
    e.sync()
    dt = (time.perf_counter() - t0) / 3
    nb = sum(im.size for im in imgs)
    print("%s (C calls only, pageable host memory): %.1f ms per instance, %.0f MB image, %.2f GB/s" % (name, dt * 1e3, nb / 1e6, nb / dt / 1e9))
e.close()
