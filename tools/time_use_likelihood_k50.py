import os, sys, time
sys.path[:0] = ['.', 'tests']
from telescope_amd import _lib, synthetic
from telescope_amd.likelihood import TelescopeLikelihood
class Opts(object):
    em_epsilon, max_iter, pi_prior, theta_prior, use_likelihood = 0.0, 64, 0, 200000, True
eng = _lib.Engine(0)
eng.set_option('kernel_timing', 0)
eng.set_option('use_likelihood', 1)
if len(sys.argv) > 1: eng.set_option('fused_dbg', int(sys.argv[1]))
eng.generate(0, 10_000_000, 50_000, synthetic.poisson_cdf_u32(100), 42, synthetic.DIST_CODE['zipf'], 0.02)
tl = TelescopeLikelihood.from_engine(eng, Opts())
eng.prepare_likelihood()
import logging; logging.disable(logging.CRITICAL)
for rep in range(3):
    eng.synchronize(); t = time.time()
    tl.em(use_likelihood=True)
    eng.synchronize(); print('em(use_likelihood) %d iterations: %.3f ms per iteration' % (tl.n_iter, (time.time() - t) / tl.n_iter * 1e3))
