import sys, os
sys.path[:0] = ['/root/repo', '/root/repo/tests']
import numpy as np, scipy.sparse as sp
from test_gpu_round6 import _near_tie_matrix
from conftest import Opts
from oracle.telescope_oracle import OracleModel
from telescope_amd import _lib
from telescope_amd.likelihood import TelescopeLikelihood, score_lut
raw, pi, theta = _near_tie_matrix(1)
n, k = raw.shape
eng = _lib.Engine(0)
eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), k, score_lut(int(raw.data.max())))
tl = TelescopeLikelihood.from_engine(eng, Opts(max_iter=1, em_epsilon=0.0)); tl._raw = raw
eng.set_params(pi, theta)
om = OracleModel(raw, 0, 200000); om.z = om.estep(pi, theta)
for rep in range(2):
  for m in ('exclude', 'average', 'exclude', 'conf'):
    cs, mask = eng.reassign(m, 0.9, _lib.Z_CUR, want_mask=True)
    mo = sp.csr_matrix(om.reassign(m, 0.9)).astype(np.float64)
    dm = sp.csr_matrix((mask, raw.indices.copy(), raw.indptr.copy()), shape=raw.shape); dm.eliminate_zeros()
    d = (dm - mo).tocsr(); d.eliminate_zeros()
    rows = np.flatnonzero(np.diff(d.indptr))
    print(rep, m, 'rows differing', len(rows), rows[:10], 'near', eng.layout_info()['near_tie_rows'])
    for r in rows[:3]:
        s, e = raw.indptr[r], raw.indptr[r+1]
        print('  row', r, 'len', e - s, 'engine', mask[s:e], 'oracle', mo[r].toarray().ravel()[raw.indices[s:e]])
        zz = sp.csr_matrix(om.z)[r].toarray().ravel()[raw.indices[s:e]]
        print('  z', [x.hex() for x in zz])
