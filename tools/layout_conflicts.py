"""LDS conflict profile of the blocked layout: for the wave instructions of a few sub-blocks, the largest number
of lanes of a 16-lane group whose slots agree modulo 16 (the granularity at which ds_add_f64 serialises:
tools/ubench/lds.hip) — columns for the accumulator scatter, rows for the row-sum scatter — and modulo 32 within 32
lanes for the pi*theta gather (ds_read_b64).  python tools/layout_conflicts.py [k=v ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine
from telescope_amd.likelihood import TelescopeLikelihood

class O: em_epsilon = 0.0; max_iter = 1; pi_prior = 0; theta_prior = 200000
eng = Engine(0)
for a in sys.argv[1:]:
    eng.set_option(a.split('=')[0], int(a.split('=')[1]))
eng.generate(0, 2_000_000, 30000, synthetic.poisson_cdf_u32(40), 42, 1, 0.0)
tl = TelescopeLikelihood.from_engine(eng, O())
info = eng.layout_info()
col_max, row_max, row_dup, col32 = [], [], [], []
for b in range(10, 40):
    for p in range(info['P']):
        w = eng.debug_subblock(b, p)
        n = len(w) // 64 * 64
        g = w[:n].reshape(-1, 16, 4)                       # 16-lane group, lane, instruction slot
        for j in range(4):
            cols, rows = g[:, :, j] & 0xFFFF, g[:, :, j] >> 16
            for x in range(g.shape[0]):
                col_max.append(np.bincount(cols[x] & 15, minlength=16).max())
                row_max.append(np.bincount(rows[x] & 15, minlength=16).max())
                row_dup.append(np.unique(rows[x], return_counts=True)[1].max())
        g32 = w[:len(w) // 128 * 128].reshape(-1, 32, 4)
        for j in range(4):
            for x in range(g32.shape[0]):
                cc = g32[x, :, j] & 0xFFFF
                u = np.unique(cc)                                   # equal addresses broadcast
                col32.append(np.bincount(u & 31, minlength=32).max())
print(info)
print('mean over 16-lane groups of the worst class (mod 16) multiplicity:  columns %.2f   rows %.2f   (same row in one group: %.2f)'
      % (np.mean(col_max), np.mean(row_max), np.mean(row_dup)))
print('mean over 32-lane groups of the worst bank-pair (mod 32) load of the column gather: %.2f' % np.mean(col32))
