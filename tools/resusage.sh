#!/bin/bash
# compact per-kernel register / LDS / occupancy table of one csrc file:  tools/resusage.sh fit.hip [extra flags]
cd /root/repo/psi-release_amd/csrc
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -munsafe-fp-atomics "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/ru.o 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1)[:80]; d={}
    for k,n in (('VGPRs','vgpr'),('AGPRs','agpr'),('TotalSGPRs','sgpr'),('Occupancy \[waves/SIMD\]','occ'),('LDS Size \[bytes/block\]','lds'),('ScratchSize \[bytes/lane\]','scratch')):
        m=re.search(k+r': (\d+)',l)
        if m: d[n]=m.group(1)
    if 'LDS Size' in l: print(cur,d)
"
