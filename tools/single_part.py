"""how many ambiguous rows have ALL their entries in one column part (such a row would need no team exchange),
per column distribution and row length, with the iteration time beside it.   python tools/single_part.py [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from telescope_amd import _lib, synthetic   # noqa: E402
from telescope_amd.likelihood import TelescopeLikelihood   # noqa: E402


class Opts(object):
    em_epsilon, max_iter, pi_prior, theta_prior = 0.0, 20, 0, 200000


rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
print('%d rows x 30 000 loci; single-part rows = ambiguous rows whose entries all lie in one of the P column parts' % rows)
print('%-8s %4s %-7s %3s %6s %14s %10s %10s' % ('dist', 'nnz', 'format', 'P', 'R', 'single-part', 'ms/iter', 'roofline'))
for dist in ('zipf', 'uniform', 'family'):
    for d in (10, 18, 40):
        for fmt in (0, 1):
            eng = _lib.Engine(0)
            eng.set_option('value_format', fmt)
            eng.set_option('kernel_timing', 0)
            eng.generate(0, rows, 30_000, synthetic.poisson_cdf_u32(d), 42, synthetic.DIST_CODE[dist], 0.0)
            tl = TelescopeLikelihood.from_engine(eng, Opts())
            info = eng.layout_info()
            eng.em_chunk(3, 0.0, False, first=True)
            eng.synchronize()
            t0 = time.perf_counter()
            eng.em_chunk(20, 0.0, False)
            eng.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / 20
            ks = eng.kernel_stats()
            frac = ks['algo_bytes_per_pass'] / (ms * 1e-3) / 8e12
            print('%-8s %4d %-7s %3d %6d %13.4f%% %10.3f %10.3f' % (dist, d, 'codes' if info['value_bytes'] == 2 else 'fp64', info['P'], info['R'],
                                                                100.0 * info['single_part_rows'] / max(1, info['N_amb']), ms, frac), flush=True)
            eng.close()
            del tl
