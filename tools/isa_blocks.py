#!/usr/bin/env python3
"""Basic-block table of one kernel's ISA: per block the instruction mix (VALU, of it fp64 / transcendental, SALU, LDS,
global loads / stores, scratch = spill traffic, waits, branches) and where it branches to, to find a loop's real
instruction budget.
usage: isa_blocks.py <asm file from hipcc -S --cuda-device-only> <mangled-name> [min_instructions]"""
import re, sys
txt = open(sys.argv[1]).read()
name = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 0
i = txt.index('\n' + name + ':')
j = txt.index('.end_amdhsa_kernel', i)
j = txt.rindex('s_endpgm', i, j)
blocks = []
cur = {'label': 'entry', 'ins': []}
for l in txt[i:j].splitlines()[1:]:
    lab = re.match(r'^(\.LBB\d+_\d+):', l)
    if lab:
        blocks.append(cur); cur = {'label': lab.group(1), 'ins': []}
        continue
    m = re.match(r'^\s+([a-z_0-9]+)\s*(.*)$', l)
    if m and not m.group(1).startswith('.'): cur['ins'].append((m.group(1), m.group(2).split(';')[0].strip()))
blocks.append(cur)
print('%-12s %5s %5s %5s %5s %5s %5s %5s %5s %5s %5s  %s' % ('block', 'n', 'valu', 'f64', 'trans', 'salu', 'lds', 'gld', 'gst', 'scr', 'wait', 'branches'))
tot = {}
for b in blocks:
    c = dict(n=len(b['ins']), valu=0, f64=0, trans=0, salu=0, lds=0, gld=0, gst=0, scr=0, wait=0)
    br = []
    for op, arg in b['ins']:
        if op.startswith('v_'):
            c['valu'] += 1
            if '_f64' in op: c['f64'] += 1
            if re.match(r'v_(rcp|rsq|sqrt|exp|log|sin|cos)_', op): c['trans'] += 1
        elif op.startswith('ds_'): c['lds'] += 1
        elif op.startswith('global_load') or op.startswith('buffer_load'): c['gld'] += 1
        elif op.startswith('global_store') or op.startswith('buffer_store'): c['gst'] += 1
        elif op.startswith('scratch_'): c['scr'] += 1
        elif op == 's_waitcnt': c['wait'] += 1
        elif op.startswith('s_cbranch') or op == 's_branch': br.append(op.replace('s_cbranch_', '') + '->' + arg); c['salu'] += 1
        elif op.startswith('s_'): c['salu'] += 1
    if b['label'] == 'entry' or c['n'] >= minn:
        print('%-12s %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d  %s' % (b['label'], c['n'], c['valu'], c['f64'], c['trans'], c['salu'], c['lds'], c['gld'], c['gst'], c['scr'], c['wait'], ' '.join(br)))
    for k, v in c.items(): tot[k] = tot.get(k, 0) + v
print('%-12s %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d' % ('total', tot['n'], tot['valu'], tot['f64'], tot['trans'], tot['salu'], tot['lds'], tot['gld'], tot['gst'], tot['scr'], tot['wait']))
