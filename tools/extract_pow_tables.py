#!/usr/bin/env python3
"""Reads __pow_log_data out of the libm.so.6 of this image (glibc 2.35, x86-64): Ln2hi, Ln2lo, the 7 polynomial coefficients
and the 128 {invc, pad, logc, logctail} entries that __pow_fma addresses at 0xb1b20 (tools: llvm-objdump of pow's IFUNC
target), checks the recognisable constants and prints the C initialisers used in
tempestmodel_amd/csrc/tmx_refmath.h.  The data is a table of numbers (1/c, log c for 128 subintervals), not code."""
import struct, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/usr/lib/x86_64-linux-gnu/libm.so.6"
vaddr = int(sys.argv[2], 16) if len(sys.argv) > 2 else 0xb1b20
b = open(path, "rb").read()
assert b[:4] == b"\x7fELF" and b[4] == 2
phoff, = struct.unpack_from("<Q", b, 0x20); phentsize, phnum = struct.unpack_from("<HH", b, 0x36)
off = None
for i in range(phnum):
    p_type, p_flags, p_offset, p_vaddr, p_paddr, p_filesz, p_memsz, p_align = struct.unpack_from("<IIQQQQQQ", b, phoff + i * phentsize)
    if p_type == 1 and p_vaddr <= vaddr < p_vaddr + p_filesz:
        off = p_offset + (vaddr - p_vaddr)
assert off is not None
d = struct.unpack_from("<%dd" % (2 + 7 + 128 * 4), b, off)
assert d[0].hex() == "0x1.62e42fefa3800p-1" and d[1].hex() == "0x1.ef35793c76730p-45" and d[2] == -0.5, (d[0].hex(), d[1].hex(), d[2])
print("/* ln2hi %s ln2lo %s */" % (d[0].hex(), d[1].hex()))
print("poly:", ", ".join(x.hex() for x in d[2:9]))
tab = d[9:]
assert all(tab[4 * i + 1] == 0.0 for i in range(128))
print("TMX_RM_TAB double tmx_rm_powlog_tab[384] = {      /* invc, logc, logctail of __pow_log_data (128 subintervals) */")
for i in range(128):
    print("\t%s, %s, %s," % (tab[4 * i].hex(), tab[4 * i + 2].hex(), tab[4 * i + 3].hex()))
print("};")
