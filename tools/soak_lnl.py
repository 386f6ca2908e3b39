"""Soak of the carried log-likelihood (`--use_likelihood`, fused kernel MODE 4): thousands of iterations through tsem_em_chunk with the
lagged convergence test armed (epsilon 0: it never fires): every value of the trace finite, the trace settling (the reference's
lnl = sum z log1p(Q c), model.py:744-760, is not the likelihood EM maximises: it need not grow), no fall-back.
    python tools/soak_lnl.py [rows] [iterations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from telescope_amd import _lib, synthetic   # noqa: E402
from telescope_amd.likelihood import TelescopeLikelihood   # noqa: E402


class Opts(object):
    em_epsilon, max_iter, pi_prior, theta_prior, use_likelihood = 0.0, 10 ** 9, 0, 200000, True


rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
for d, cols in ((40, 30_000), (18, 30_000), (100, 50_000)):
    eng = _lib.Engine(0)
    eng.set_option('kernel_timing', 0)
    eng.set_option('use_likelihood', 1)
    eng.generate(0, rows, cols, synthetic.poisson_cdf_u32(d), 42, synthetic.DIST_CODE['zipf'], 0.02)
    tl = TelescopeLikelihood.from_engine(eng, Opts())
    eng.prepare_likelihood()
    info = eng.layout_info()
    done, trace, carry = 0, [], None
    t0 = time.perf_counter()
    while done < iters:
        n = min(16, iters - done)
        diffs, lnls, stopped = eng.em_chunk(n, 0.0, True, first=(done == 0), last=(done + n >= iters))
        if carry is not None:
            trace.append(float(eng.lnl_carry))
        trace.extend(float(x) for x in lnls[:-1])
        carry = lnls[-1] if len(lnls) else None
        if carry is not None and carry == carry:
            trace.append(float(carry)); carry = None
        done += len(diffs)
        assert not stopped
    eng.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / iters
    tr = np.asarray(trace)
    ok = bool(np.isfinite(tr).all()) and len(tr) == iters and abs(tr[-1] - tr[-2]) <= 1e-6 * abs(tr[-1])
    print('%9d x %6d x %3d  %5d iterations  %.3f ms/iteration  lnl_fused %d P %d geo %d  fallbacks %d  trace %d values, all finite and settled: %s  (first %.6e last %.6e)'
          % (rows, cols, d, done, ms, info['lnl_fused'], info['P'], info['geometry'], eng.layout_info()['fallbacks'], len(tr), ok, tr[0], tr[-1]), flush=True)
    assert ok and eng.layout_info()['fallbacks'] == 0
    eng.close()
    del tl
