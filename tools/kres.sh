#!/bin/bash
# kernel resource usage table of one translation unit: tools/kres.sh [horizontal|vertical|column|exchange|physics|unique] [extra flags]
TU=${1:-column}; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c /root/repo/tempestmodel_amd/csrc/tmx_k_$TU.hip -o /tmp/k.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'remark:\s+([\w \[\]/]+?): (\S+) \[-Rpass',l)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2)
    if k=='Function Name':
        if cur: print(cur)
        cur=v[:36]+' '
    elif k.split()[0] in ('VGPRs','AGPRs','Occupancy','LDS','ScratchSize') and 'Spill' not in k:
        cur+=k.split()[0][:4]+'='+v+' '
if cur: print(cur)
"
