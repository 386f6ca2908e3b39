"""Soak of option `reproducible`: thousands of exact EM iterations (eps = 0) on several shapes, one-pass and two-pass forms, two
contexts per shape compared bit for bit every 500 iterations; reports repeated passes, time-outs / fall-backs (must be 0).
python tools/soak_reproducible.py [iterations]"""
import os, sys, time, logging
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine
from telescope_amd.likelihood import TelescopeLikelihood
logging.disable(logging.WARNING)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
class O: em_epsilon = 0.0; max_iter = 10 ** 9; pi_prior = 0; theta_prior = 200000
for rows, cols, d, form in ((5_000_000, 15_000, 24, 1), (5_000_000, 15_000, 8, 1), (5_000_000, 30_000, 10, 1), (3_000_000, 30_000, 100, 1),
                            (5_000_000, 30_000, 40, 2), (2_000_000, 5_000, 30, 1)):
    engs = []
    for k in range(2):
        e = Engine(0); e.set_option('reproducible', form)
        e.generate(0, rows, cols, synthetic.poisson_cdf_u32(d), 42, 1, 0.05)
        TelescopeLikelihood.from_engine(e, O())
        engs.append(e)
    t0 = time.perf_counter(); done = 0; same = True
    while done < iters:
        for e in engs:
            e.em_chunk(500, 0.0, False)
        done += 500
        a, b = engs[0].get_params(), engs[1].get_params()
        same &= bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))
    el = time.perf_counter() - t0
    i0 = engs[0].layout_info()
    print('%9d x %5d x %3d  form %s  P %d geometry %d: %d iterations x 2 contexts, %.2f ms/iteration, repeated passes %d, fall-backs %d, bit-identical %s, pi sum %.15f'
          % (rows, cols, d, 'one-pass' if i0['exact_single'] else 'two-pass', i0['P'], i0['geometry'], done, el * 1e3 / (2 * done),
             i0['bin_repeats'], i0['fallbacks'], same, float(a[0].sum())), flush=True)
    del engs
