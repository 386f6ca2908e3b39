"""Wall clock of the drop-in constructor on a HOST scipy matrix (what telescope_assign.run hands over):
TelescopeLikelihood(raw_scores, opts) -> em() -> the seven report column sums.   python tools/time_constructor.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, scipy.sparse as sp
from telescope_amd import synthetic
from telescope_amd._lib import Engine
from telescope_amd.likelihood import TelescopeLikelihood

class O: em_epsilon = 1e-7; max_iter = 100; pi_prior = 0; theta_prior = 200000
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
src = Engine(0)
src.generate(0, rows, 30000, synthetic.poisson_cdf_u32(40), 42, 1, 0.0)
indptr, indices, raw = src.export_csr(); src.close()
m = sp.csr_matrix((raw, indices, indptr), shape=(rows, 30000))
m.has_canonical_format                                       # (scipy caches the answer on the matrix; the loader's matrices have it)
print('host matrix: %d rows, %d entries, indptr %s' % (rows, m.nnz, m.indptr.dtype))
t0 = time.perf_counter(); tl = TelescopeLikelihood(m, O()); tl._eng.synchronize(); t1 = time.perf_counter()
tl.em(); t2 = time.perf_counter()
for method, initial in (('conf', False), ('all', False), ('unique', False), ('exclude', False), ('choose', False),
                        ('average', False), ('exclude', True)):
    tl.reassign(method, 0.9, initial).sum(0)
t3 = time.perf_counter()
print('constructor (host checks + copy + setup) %.1f ms | em() %.1f ms | seven report columns %.1f ms | total %.1f ms'
      % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3))
# the same steps one by one
from telescope_amd.likelihood import score_lut
t = [time.perf_counter()]
def lap(name, eng=None):
    if eng is not None: eng.synchronize()
    t.append(time.perf_counter()); print('  %-34s %7.1f ms' % (name, (t[-1] - t[-2]) * 1e3))
r = sp.csr_matrix(m); r.has_canonical_format; lap('csr_matrix(m), canonical flag')
eng = Engine(0); lap('Engine(0)')
eng.load_scores(r.indptr, r.indices, r.data.astype(np.uint16, copy=False), 30000, None); lap('load_scores (copy + validation)', eng)
mx = eng.max_score(); lap('max_score (device)', eng)
eng.set_lut(score_lut(mx)); lap('score table', eng)
st = eng.rowstats(); lap('rowstats', eng)
eng.set_model(st[0], st[1], st[2], st[3], 0.0, 200000.0); lap('set_model (layout)', eng)
