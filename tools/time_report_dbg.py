"""Timing experiments on k_report_rows (option report_dbg drops parts of its work; results are then wrong)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine, Z_INITIAL, Z_PREV
from telescope_amd.likelihood import TelescopeLikelihood
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
d = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
class O: em_epsilon = 0.0; max_iter = 5; pi_prior = 0; theta_prior = 200000
eng = Engine(0)
eng.generate(0, rows, 30000, synthetic.poisson_cdf_u32(d), 42, 1, 0.05)
tl = TelescopeLikelihood.from_engine(eng, O())
tl.em()
for dbg in [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '0,1,2,3,4,7').split(',')]:
    eng.set_option('report_dbg', dbg)
    for which, name in ((Z_PREV, 'final'), (Z_INITIAL, 'initial')):
        best = 1e9
        for _ in range(3):
            eng.synchronize(); t0 = time.perf_counter()
            eng.report_colsums(which, 0.9)
            best = min(best, time.perf_counter() - t0)
        print('dbg=%d %-7s %7.2f ms (wall)' % (dbg, name, best * 1e3), flush=True)
