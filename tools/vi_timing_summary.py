#!/usr/bin/env python3
"""Summarise the per-wavefront cycle records of tools/vi_timing.py (stdin: its stderr lines)."""
import sys, re, collections
rows = []
for l in sys.stdin:
    m = re.match(r'pairtiming wave (\d+) role (\d) hw_id (\w+) xcc (\d+) work (\d+) wait (\d+) forward (\d+) back (\d+)', l)
    if m:
        w, role, hw, xcc, work, wait, fwd, back = m.groups()
        hw = int(hw, 16)
        rows.append(dict(wave=int(w), role=int(role), simd=(hw >> 4) & 3, cu=(hw >> 8) & 15, se=(hw >> 13) & 7, sh=(hw >> 12) & 1, xcc=int(xcc), work=int(work), wait=int(wait), fwd=int(fwd), back=int(back)))
# waves per physical CU
percu = collections.Counter((r['xcc'], r['se'], r['sh'], r['cu']) for r in rows)
def stat(v):
    v = sorted(v); n = len(v)
    return "n %4d  min %7d  med %7d  p90 %7d  max %7d" % (n, v[0], v[n // 2], v[int(n * 0.9)], v[-1]) if n else "n 0"
for load in sorted(set(percu.values())):
    sel = [r for r in rows if percu[(r['xcc'], r['se'], r['sh'], r['cu'])] == load]
    print("CUs holding %d wavefronts of this kernel: %d" % (load, sum(1 for v in percu.values() if v == load)))
    for role, name in ((0, "producer"), (1, "consumer")):
        s = [r for r in sel if r['role'] == role]
        print("  %s work   " % name, stat([r['work'] for r in s]))
        print("  %s wait   " % name, stat([r['wait'] for r in s]))
        print("  %s forward" % name, stat([r['fwd'] for r in s]))
        if role == 1:
            print("  %s back   " % name, stat([r['back'] for r in s]))
