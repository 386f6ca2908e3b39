"""What option `reproducible` costs: the same EM run (synthetic Zipf matrix, eps 1e-4) in the default mode and with exact
column sums, twice each; prints iterations, wall-clock of em(), time per iteration, repeated passes, and whether the two
runs of a mode agree bit for bit.
python tools/time_reproducible.py [rows] [nnz_row] [value_format] [cols]
(reproducible = 2: the two-pass form; 1: both pieces in one pass when three tables per part fit the LDS, else the same as 2)"""
import os, sys, time, logging
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine
from telescope_amd.likelihood import TelescopeLikelihood
logging.disable(logging.WARNING)

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
d = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
fmt = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cols = int(sys.argv[4]) if len(sys.argv) > 4 else 30000
class O: em_epsilon = 1e-4; max_iter = 100; pi_prior = 0; theta_prior = 200000; use_likelihood = False
print('# rows %d  entries/row %.0f  value_format %d  columns %d' % (rows, d, fmt, cols))
for rep_mode in (0, 2, 1):
    out = []
    for run in range(2):
        eng = Engine(0)
        eng.set_option('reproducible', rep_mode); eng.set_option('value_format', fmt)
        eng.generate(0, rows, cols, synthetic.poisson_cdf_u32(d), 42, 1, 0.05)
        tl = TelescopeLikelihood.from_engine(eng, O())
        eng.synchronize(); t0 = time.perf_counter()
        tl.em()
        eng.synchronize(); t = time.perf_counter() - t0
        info = eng.layout_info()
        out.append((tl.n_iter, tl.pi.copy(), tl.theta.copy(), tl.lnl))
        print('reproducible=%d run %d: %3d iterations  em() %8.1f ms  %6.2f ms/iteration  repeated passes %d  lnl %.17g  P %d geometry %d one-pass %d'
              % (rep_mode, run, tl.n_iter, t * 1e3, t * 1e3 / tl.n_iter, info['bin_repeats'], tl.lnl, info['P'], info['geometry'], info['exact_single']), flush=True)
        del tl, eng
    a, b = out
    print('reproducible=%d: runs agree bit for bit: pi %s  theta %s  lnl %s  (max rel pi difference %.2e)'
          % (rep_mode, np.array_equal(a[1], b[1]), np.array_equal(a[2], b[2]), a[3] == b[3],
             float(np.max(np.abs(a[1] - b[1]) / np.maximum(a[1], 1e-300)))), flush=True)
