"""K beyond 8 x 7680 columns: the split layout of the fused kernel (two light passes per iteration) against the two-pass kernels
it replaces and against K = 50k on the ordinary fused kernel (<= 2x the per-entry time of K = 50k).
    python tools/time_large_k.py  [> profiles/r04_large_k.txt]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from telescope_amd import _lib, synthetic   # noqa: E402
from telescope_amd.likelihood import TelescopeLikelihood   # noqa: E402


class Opts(object):
    em_epsilon, max_iter, pi_prior, theta_prior = 0.0, 20, 0, 200000


def run(rows, cols, d, options=()):
    eng = _lib.Engine(0)
    eng.set_option('kernel_timing', 0)
    for k, v in options:
        eng.set_option(k, v)
    eng.generate(0, rows, cols, synthetic.poisson_cdf_u32(d), 42, synthetic.DIST_CODE['zipf'], 0.0)
    tl = TelescopeLikelihood.from_engine(eng, Opts())
    info = eng.layout_info()
    _, _, nnz = eng.dims()
    eng.em_chunk(3, 0.0, False, first=True)
    eng.synchronize()
    t0 = time.perf_counter()
    eng.em_chunk(20, 0.0, False)
    eng.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 20
    t0 = time.perf_counter()
    lnl = eng.final_lnl()
    t0 = time.perf_counter()
    lnl = eng.final_lnl()
    lnl_ms = (time.perf_counter() - t0) * 1e3
    pi, _ = eng.get_params()
    eng.close()
    del tl
    return ms, nnz, info, lnl, lnl_ms, pi


def line(tag, rows, cols, d, options=()):
    ms, nnz, info, lnl, lnl_ms, pi = run(rows, cols, d, options)
    print('%-44s %9d x %6d x %3d  nnz %.2e  %7.3f ms/iter  %6.2f ps/entry  lnl pass %6.2f ms   P %d Kp %d R %d geo %d fused %d split %d value_bytes %d'
          % (tag, rows, cols, d, nnz, ms, ms * 1e9 / nnz, lnl_ms, info['P'], info['Kp'], info['R'], info['geometry'], info['fused'],
             info['split'], info['value_bytes']), flush=True)
    return ms * 1e9 / nnz, lnl, pi


if __name__ == '__main__':
    import numpy as np
    if len(sys.argv) > 1 and sys.argv[1] == 'one':          # one configuration (for rocprofv3): one <rows> <cols> <per row> [value_format]
        line('one', int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), (('value_format', int(sys.argv[5])),) if len(sys.argv) > 5 else ())
        sys.exit(0)
    for rows, d in ((4_000_000, 100), (10_000_000, 40)):
        print('== %d rows x ~%d per row' % (rows, d))
        base, _, _ = line('K = 50k, fused (codes)', rows, 50_000, d)
        base8, _, _ = line('K = 50k, fused (fp64 entries)', rows, 50_000, d, (('value_format', 1),))
        a, l1, p1 = line('K = 100k, split layout (codes)', rows, 100_000, d)
        b, _, _ = line('K = 100k, split layout (fp64 entries)', rows, 100_000, d, (('value_format', 1),))
        c, l2, p2 = line('K = 100k, two-pass kernels (round 3)', rows, 100_000, d, (('em_kernel', _lib.EMK_TWOPASS),))
        print('   per entry against K = 50k: split %.2fx (codes) / %.2fx (fp64 entries), two-pass %.2fx;  lnl rel. delta split vs two-pass %.1e, '
              'pi max rel. delta %.1e' % (a / base, b / base8, c / base, abs(l1 - l2) / abs(l2), float(np.max(np.abs(p1 - p2) / np.maximum(p2, 1e-300)))))
    print('== K = 122 880 (the most the split layout takes), 4M rows x ~100 per row')
    line('K = 122 880, split layout', 4_000_000, 122_880, 100)
