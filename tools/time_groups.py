"""Per-barcode sums (tsem_reassign_groups) at the scale BASELINE config 5 names, against one report pass over the same matrix:
    python tools/time_groups.py [rows=5000000] [cols=50000] [nnz_row=100] [groups=2000]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine, Z_PREV
from telescope_amd.likelihood import TelescopeLikelihood


class O:
    em_epsilon = 0.0; max_iter = 3; pi_prior = 0; theta_prior = 200000


rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
nnz_row = int(sys.argv[3]) if len(sys.argv) > 3 else 100
n_groups = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
eng = Engine(0)
eng.generate(0, rows, cols, synthetic.poisson_cdf_u32(nnz_row), 42, 1, 0.05)
tl = TelescopeLikelihood.from_engine(eng, O())
tl.em()
_, _, nnz = eng.dims()
print('%d rows x %d loci, %d stored entries, %d groups: the count matrix is %.0f MB' % (rows, cols, nnz, n_groups, n_groups * cols * 8 / 1e6))


def timed(f, n=3):
    f()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    eng.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


rep = timed(lambda: eng.report_colsums(Z_PREV, 0.9))
print('one report pass (tsem_report_colsums: conf | exclude | average + tie list, incl. its host copies)  %8.2f ms' % rep)
bc = np.random.RandomState(11).randint(0, n_groups, rows).astype(np.int32)
t0 = time.perf_counter(); eng.set_groups(bc, n_groups); eng.synchronize()
print('tsem_set_groups (copy + range check of the %d-entry map)                                    %8.2f ms' % (rows, (time.perf_counter() - t0) * 1e3))
out = np.zeros((n_groups, cols))
for method in ('exclude', 'average', 'conf', 'unique', 'all'):
    ms = timed(lambda: eng.reassign_groups(method, 0.9, Z_PREV, None, n_groups, out=out))
    print('tsem_reassign_groups %-8s incl. the %4.0f MB copy to the host  %8.2f ms  = %.2f x the report pass' % (method, out.nbytes / 1e6, ms, ms / rep))
t0 = time.perf_counter(); out[:] = 0; dt = (time.perf_counter() - t0) * 1e3
print('(for scale: zero-filling the host matrix once takes %.1f ms)' % dt)
