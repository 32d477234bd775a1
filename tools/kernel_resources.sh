#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [extra hipcc flags]   -- one line per kernel: VGPRs, spills, occupancy, LDS (hipcc remarks)
F=$1; shift
cd "$(dirname "$0")/../deepcgp_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $F -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import re, subprocess, sys
cur = None
rows = []
for ln in sys.stdin:
    m = re.search(r'remark:\s+(.*?) \[-Rpass', ln)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1); cur[k.strip()] = v.strip()
for r in rows:
    try: name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', r['name']], capture_output=True, text=True).stdout.strip()
    except Exception: name = r['name']
    name = re.sub(r'\(anonymous namespace\)::', '', name)[:70]
    print('%-70s VGPR %4s  AGPR %3s  spillV %3s  spillS %3s  occ %s  LDS %s' % (name, r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill'), r.get('SGPRs Spill'), r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]')))
"
rm -f /tmp/kr_$$.o
