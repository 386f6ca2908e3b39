#!/usr/bin/env python3
"""Register / scratch usage of the kernels of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel:
    python tools/kernel_resources.py telescope_amd/csrc/tsem_fz_p4.hip [filter]
A spill in the fused kernel is a scratch access in the in-order memory pipe behind the stream: the table is how a change is checked."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
csrc = os.path.join(ROOT, 'telescope_amd', 'csrc')
cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics'] + ([] if os.path.basename(src).startswith('tsem_fz_p') else ['-ffp-contract=off']) + ['-I' + os.path.join(ROOT, 'include'),
       '-I' + csrc, '-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'] + sys.argv[3:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in err.splitlines():
    m = re.search(r'Function Name: (\S+)', ln) or re.search(r'remark: .*?Name: (\S+)', ln)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r'remark: [^:]*:\d+:\d+:\s+([A-Za-z ]+?)(?: \[bytes/lane\])?(?: \[waves/SIMD\])?: (\d+)', ln) or re.search(r'\s+([A-Za-z ]+?)(?: \[bytes/lane\])?(?: \[waves/SIMD\])?: (\d+) \[-Rpass', ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    try:
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except OSError:
        dem = name
    if flt and flt not in dem:
        continue
    print('%-60s VGPR %3d  AGPR %3d  scratch %4d  vgpr-spill %3d  sgpr-spill %3d  LDS %6d' % (
        dem[:60], r.get('VGPRs', -1), r.get('AGPRs', -1), r.get('ScratchSize', -1), r.get('VGPRs Spill', -1), r.get('SGPRs Spill', -1),
        r.get('LDS Size', -1)))
