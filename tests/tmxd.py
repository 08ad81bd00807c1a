"""Reader/writer for the TMXD record container used by the golden fixtures.

Layout (little endian): magic ``TMXD0001`` then records
``u32 name_len | name | u32 dtype (0=f64, 1=i32) | u32 ndim | u64 dims[ndim] | raw data``.
Written by ``oracle/ref_dump.cpp`` (the driver that links the real reference).
"""
import struct
import numpy as np


def read(path):
    out = {}
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != b"TMXD0001":
        raise ValueError("not a TMXD file: %s" % path)
    off = 8
    n = len(buf)
    while off < n:
        (nl,) = struct.unpack_from("<I", buf, off); off += 4
        name = buf[off:off + nl].decode(); off += nl
        dt, nd = struct.unpack_from("<II", buf, off); off += 8
        dims = struct.unpack_from("<%dQ" % nd, buf, off); off += 8 * nd
        cnt = int(np.prod(dims)) if nd else 1
        if dt == 0:
            a = np.frombuffer(buf, dtype="<f8", count=cnt, offset=off).reshape(dims).copy(); off += 8 * cnt
        else:
            a = np.frombuffer(buf, dtype="<i4", count=cnt, offset=off).reshape(dims).copy(); off += 4 * cnt
        out[name] = a
    return out


def write(path, records):
    with open(path, "wb") as f:
        f.write(b"TMXD0001")
        for name, a in records.items():
            a = np.asarray(a)
            if a.dtype.kind == "f":
                a = np.ascontiguousarray(a, dtype="<f8"); dt = 0
            else:
                a = np.ascontiguousarray(a, dtype="<i4"); dt = 1
            nb = name.encode()
            f.write(struct.pack("<I", len(nb))); f.write(nb)
            f.write(struct.pack("<II", dt, a.ndim))
            f.write(struct.pack("<%dQ" % a.ndim, *a.shape))
            f.write(a.tobytes())
