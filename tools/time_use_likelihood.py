"""ms per EM iteration under --use_likelihood: the carrying pass (option use_likelihood = 1: fused kernel MODE 4) against the
separate lnl pass per iteration, and against a default iteration.  python tools/time_use_likelihood.py [rows=50000000] [nnz_row=40] [cols=30000]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine
from telescope_amd.likelihood import TelescopeLikelihood


class O:
    em_epsilon = 0.0; max_iter = 1000; pi_prior = 0; theta_prior = 200000


rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
nnz_row = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 30_000
res = {}
for name, opt, use in (('default iteration (no lnl)', 0, False), ('use_likelihood, lnl pass per iteration', 0, True),
                       ('use_likelihood, lnl carried by the EM pass', 1, True)):
    eng = Engine(0)
    eng.set_option('use_likelihood', opt)
    for kv in sys.argv[4:]:
        k, v = kv.split('=')
        eng.set_option(k, int(v))
    eng.generate(0, rows, cols, synthetic.poisson_cdf_u32(nnz_row), 42, 1, 0.05)
    tl = TelescopeLikelihood.from_engine(eng, O())
    eng.set_option('kernel_timing', 0)
    eng.em_chunk(3, 0.0, use, first=True)
    eng.synchronize()
    n = 20
    t0 = time.perf_counter()
    eng.em_chunk(n, 0.0, use)
    eng.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    info = eng.layout_info()
    _, l_, _ = eng.em_chunk(2, 0.0, use, last=True)
    res[name] = (ms, info, None if l_ is None else float(l_[-1]))
    print('%-46s %8.3f ms per iteration   P %d Kp %d R %d geometry %d lnl_fused %d  lnl %r' %
          (name, ms, info['P'], info['Kp'], info['R'], info['geometry'], info['lnl_fused'], res[name][2]), flush=True)
    eng.close()
    del tl
a, b = res['use_likelihood, lnl pass per iteration'], res['use_likelihood, lnl carried by the EM pass']
print('rows %d x cols %d x %d per row: %.3f -> %.3f ms per --use_likelihood iteration; lnl after 25 iterations: %r / %r (rel. delta %.2e)'
      % (rows, cols, nnz_row, a[0], b[0], a[2], b[2], abs(a[2] - b[2]) / abs(a[2])))
