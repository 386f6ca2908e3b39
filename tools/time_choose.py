"""Where `choose` (initial z) spends its time on the bench workload.  python tools/time_choose.py [rows]"""
import os, sys, time, logging
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine, Z_INITIAL
from telescope_amd.likelihood import TelescopeLikelihood
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
class O: em_epsilon = 1e-7; max_iter = 3; pi_prior = 0; theta_prior = 200000
eng = Engine(0)
eng.generate(0, rows, 30000, synthetic.poisson_cdf_u32(40), 42, 1, 0.05)
tl = TelescopeLikelihood.from_engine(eng, O()); tl.em(loglev=logging.DEBUG)
t = [time.perf_counter()]
def lap(name):
    eng.synchronize(); t.append(time.perf_counter()); print('%-40s %8.1f ms' % (name, (t[-1] - t[-2]) * 1e3))
sums, r, c = eng.report_colsums(Z_INITIAL, 0.9); lap('report pass + tie list to the host (%d tied rows)' % len(r))
np.random.seed(1)
d = np.random.randint(0, c); lap('np.random.randint(0, counts)  [numpy, for comparison]')
from telescope_amd import _lib
np.random.seed(1)
d2 = _lib.legacy_randint(c); lap('_lib.legacy_randint(counts)    [what choose uses: same picks, same state]')
assert np.array_equal(d, d2)
d = d.astype(np.int32); lap('astype int32')
cs = eng.reassign_rows('choose', 0.9, Z_INITIAL, r, d); lap('reassign_rows (upload rows + picks, kernel)')
cs = eng.reassign_rows('choose', 0.9, Z_INITIAL, None, d, n=len(r)); lap('reassign_rows (rows left on the device)')
