"""The report pass over the FINAL z on the bench workload: wall clock per call of tsem_report_colsums, the packed fp32 kernel
(default) against the capacity kernel (report_dbg = 8) and the packed kernel's timing experiments (report_dbg 16 / 32 / 64: wrong
results).   python tools/time_report_final.py [rows] [nnz_row] [cols] [dbg values ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine, Z_PREV
from telescope_amd.likelihood import TelescopeLikelihood

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
d = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
dbgs = [int(x) for x in sys.argv[4:]] or [0, 8]
class O: em_epsilon = 0.0; max_iter = 5; pi_prior = 0; theta_prior = 200000
eng = Engine(0)
eng.generate(0, rows, cols, synthetic.poisson_cdf_u32(d), 42, 1, 0.05)
tl = TelescopeLikelihood.from_engine(eng, O())
tl.em()
ref = None
for dbg in dbgs:
    eng.set_option('report_dbg', dbg)
    best = 1e9
    for _ in range(4):
        eng.synchronize(); t0 = time.perf_counter()
        sums, r, c = eng.report_colsums(Z_PREV, 0.9)
        best = min(best, time.perf_counter() - t0)
    if ref is None:
        ref = sums
    same = np.array_equal(ref['exclude'], sums['exclude']) and np.allclose(ref['conf'], sums['conf'], rtol=1e-12, atol=1e-9)
    print('report_dbg=%3d  final z  %7.2f ms per call  deferred+near rows %d  same-as-first %s' % (dbg, best * 1e3, eng.layout_info()['near_tie_rows'], same), flush=True)
