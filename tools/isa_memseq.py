#!/usr/bin/env python3
"""Memory-op / waitcnt sequence of one kernel's ISA (labels kept), to spot vmcnt(0) stalls.
usage: isa_memseq.py <mangled-name-prefix> [max_chars]"""
import re, subprocess, sys
src = '/root/repo/tempestmodel_amd/csrc/tmx_k_column.hip'
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-S',
                       '--cuda-device-only', '-o', '/tmp/k.s', src], stderr=subprocess.DEVNULL, cwd='/tmp')
txt = open('/tmp/k.s').read()
name = sys.argv[1]
i = txt.index(name); i = txt.index(':', i)
j = txt.index('.end_amdhsa_kernel', i)
seq = []
for l in txt[i:j].splitlines():
    m = re.match(r'^\s+(s_waitcnt\s+.*|global_load\w+|global_store\w+|s_cbranch\w+\s+\S+|s_branch\s+\S+|s_barrier)', l)
    lab = re.match(r'^(\.LBB\d+_\d+):', l)
    if lab: seq.append('\n' + lab.group(1) + ':')
    if m:
        t = m.group(1).split()[0]
        if t == 's_waitcnt': seq.append('W[' + ' '.join(m.group(1).split()[1:]) + ']')
        elif t.startswith('global_load'): seq.append('L')
        elif t.startswith('global_store'): seq.append('S')
        elif t == 's_barrier': seq.append('BAR')
        else: seq.append('->' + m.group(1).split()[1])
out = []; prev = None; cnt = 0
for x in seq:
    if x == prev and len(x) == 1: cnt += 1
    else:
        if prev: out.append(prev + (str(cnt) if cnt > 1 else ''))
        prev = x; cnt = 1
out.append(prev + (str(cnt) if cnt > 1 else ''))
print(' '.join(out)[:int(sys.argv[2]) if len(sys.argv) > 2 else 100000])
