"""Ad-hoc GPU bring-up script (not a pytest file): python tools/gpu_quick.py"""
import logging, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.conftest import load_case, case_matrix, Opts, case_names
from telescope_amd.likelihood import TelescopeLikelihood

logging.basicConfig(level=logging.INFO, format='%(message)s')
for name in (sys.argv[1:] or case_names()):
    c = load_case(name)
    raw = case_matrix(c)
    tl = TelescopeLikelihood(raw, Opts(c))
    t = time.time()
    tl.em(use_likelihood=bool(c['use_likelihood']), loglev=logging.DEBUG)
    dt = time.time() - t
    print('%-20s iters %d/%d lnl %.9f ref %.9f rel %.2e  pi maxrel %.2e  (%.3fs) layout %s' % (
        name, tl.n_iter, int(c['n_iter']), tl.lnl, float(c['lnl']), abs(tl.lnl - float(c['lnl'])) / abs(float(c['lnl'])),
        np.max(np.abs(tl.pi - c['pi']) / np.maximum(np.abs(c['pi']), 1e-300)), dt, tl._eng.layout_info()))
    np.random.seed(int(c['seed']))
    for meth in ('conf', 'all', 'unique', 'exclude', 'choose', 'average'):
        for initial in (False, True):
            np.random.seed(int(c['seed']))
            cs = tl.reassign_colsums(meth, 0.9, initial)
            ref = c['ra_%s_%d_colsum' % (meth, int(initial))]
            ok = np.allclose(cs, ref, rtol=1e-9, atol=1e-12)
            if not ok:
                print('   MISMATCH', meth, initial, cs[:8], ref[:8])
