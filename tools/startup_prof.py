"""Start-up / tail timeline of ONE fused EM launch (every workgroup's thread 0, 100 MHz wall clock):
    python tools/startup_prof.py [rows=6250000] [nnz_row=40] [option=value ...]
Prints, in microseconds relative to the first workgroup's entry: when the workgroups entered, had counted the tickets, had zeroed
their LDS, had their tables, started the loop, left the loop and exited (min / median / max over the workgroups of teams that
formed), and the per-iteration wall time of tsem_em_chunk around it."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine
from telescope_amd.likelihood import TelescopeLikelihood


class O:
    em_epsilon = 0.0; max_iter = 3; pi_prior = 0; theta_prior = 200000


rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6_250_000
nnz_row = int(sys.argv[2]) if len(sys.argv) > 2 and '=' not in sys.argv[2] else 40
eng = Engine(0)
eng.set_option('em_kernel', 2)
for kv in sys.argv[2:]:
    if '=' in kv:
        k, v = kv.split('=')
        eng.set_option(k, int(v))
eng.generate(0, rows, 30000, synthetic.poisson_cdf_u32(nnz_row), 42, 1, 0.0)
tl = TelescopeLikelihood.from_engine(eng, O())
eng.set_option('kernel_timing', 0)
eng.em_chunk(10, 0.0, False)
eng.synchronize()
t0 = time.perf_counter()
eng.em_chunk(200, 0.0, False)
eng.synchronize()
per_iter = (time.perf_counter() - t0) / 200 * 1e6
eng.set_option('fused_prof', 2)
eng.em_steps(1, False)
t = eng.fused_startup().astype(np.int64)
info = eng.layout_info()
print(info)
live = t[:, 4] > 0
t = t[live]
base = t[:, 0].min()
names = ['entry', 'tickets counted', 'LDS zeroed', 'tables loaded', 'loop start', 'loop end', 'exit']
print('rows %d, %d per row: %.1f us per iteration (tsem_em_chunk, 200 iterations, no events); %d workgroups in teams' % (rows, nnz_row, per_iter, len(t)))
print('%-18s %10s %10s %10s   (us after the first workgroup entered)' % ('stamp', 'min', 'median', 'max'))
for i, n in enumerate(names):
    v = (t[:, i] - base) / 100.0
    print('%-18s %10.2f %10.2f %10.2f' % (n, v.min(), np.median(v), v.max()))
loop = (t[:, 5] - t[:, 4]) / 100.0
nblk = t[:, 7] & 0xFFFF
print('loop time per workgroup: min %.1f median %.1f max %.1f us; blocks per team %d..%d; us per block (median) %.3f'
      % (loop.min(), np.median(loop), loop.max(), nblk.min(), nblk.max(), np.median(loop / np.maximum(1, nblk + 5))))
if os.environ.get('PER_TEAM'):
    for rep in range(3):
        team = (t[:, 7] >> 32) & 0xFFFF; xcc = t[:, 7] >> 48
        order = np.argsort(team * 8 + ((t[:, 7] >> 16) & 0xFFFF))
        tt = ((t[:, 5] - base) / 100.0)[order]; tm = team[order]; xc = xcc[order]
        per_team = tt.reshape(-1, info['P']).max(1)
        print('launch %d: loop end per team (us), by XCD:' % rep)
        for x in range(8):
            sel = xc.reshape(-1, info['P'])[:, 0] == x
            print('  xcd %d: ' % x + ' '.join('%6.1f' % v for v in per_team[sel]))
        eng.set_option('fused_prof', 2)
        eng.em_steps(1, False)
        t = eng.fused_startup().astype(np.int64); t = t[t[:, 4] > 0]; base = t[:, 0].min()
print('kernel span (first entry -> last exit): %.1f us' % ((t[:, 6].max() - base) / 100.0))
