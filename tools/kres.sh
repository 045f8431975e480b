#!/bin/bash
# Register / LDS / spill summary of one HIP source's kernels (compile only):  bash tools/kres.sh mlp32.hip [name-filter]
R=$(cd "$(dirname "$0")/.." && pwd)
F=${1:-mlp32.hip}
PAT=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I$R/include $KRES_FLAGS \
  -c $R/enerf_amd/csrc/$F -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys
cur = None; rows = {}
for ln in sys.stdin:
    m = re.search(r'Function Name: (\S+)', ln)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)', ln)
    if m and cur: rows[cur][m.group(1).split()[0] + ('Spill' if 'Spill' in m.group(1) else '')] = int(m.group(2))
import subprocess
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    name = re.sub(r'\((?!.*>).*', '', name)
    if re.search(r'$PAT', name):
        print(f\"{name:60s} v{v.get('VGPRs',0):4d} a{v.get('AGPRs',0):4d} spill{v.get('VGPRsSpill',0):4d} scratch{v.get('ScratchSize',0):5d} occ{v.get('Occupancy',0):2d} lds{v.get('LDS',0):7d}\")
"
rm -f /tmp/kres_$$.o
