"""Wall-clock of the report pass (tsem_report_colsums: conf | exclude | average of one z + the tie rows) on the bench
workload, the streaming kernel (k_report_rows) against the generic row pass, lanes per row and workgroups per CU swept.
python tools/time_report.py [rows] [nnz_row]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine, Z_INITIAL, Z_PREV
from telescope_amd.likelihood import TelescopeLikelihood

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
d = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
class O: em_epsilon = 0.0; max_iter = 5; pi_prior = 0; theta_prior = 200000
eng = Engine(0)
eng.generate(0, rows, 30000, synthetic.poisson_cdf_u32(d), 42, 1, 0.05)
tl = TelescopeLikelihood.from_engine(eng, O())
tl.em()
ref = {}
for kern, lanes, wgs, ent, dbg in ((0, 0, 2, 0, 0), (1, 0, 2, 0, 0), (1, 0, 2, 0, 8), (1, 0, 2, 1, 8), (1, 128, 2, 0, 8), (1, 256, 2, 0, 8)):
    # (report_dbg = 8: the capacity kernel k_report_rows for the final z instead of round 6's k_report_pack)
    eng.set_option('report_dbg', dbg); eng.set_option('report_kernel', kern); eng.set_option('report_lanes', lanes); eng.set_option('rowpass_wgs', wgs); eng.set_option('report_wgs2', ent)
    for which, name in ((Z_PREV, 'final'), (Z_INITIAL, 'initial')):
        best = 1e9
        for _ in range(3):
            eng.synchronize(); t0 = time.perf_counter()
            sums, r, c = eng.report_colsums(which, 0.9)
            best = min(best, time.perf_counter() - t0)
        key = name
        if key not in ref:
            ref[key] = (sums, r, c)
        a = ref[key]
        ok = (np.array_equal(a[0]['exclude'], sums['exclude']) and np.array_equal(a[1], r) and np.array_equal(a[2], c)
              and np.allclose(a[0]['conf'], sums['conf'], rtol=1e-11, atol=1e-9) and np.allclose(a[0]['average'], sums['average'], rtol=1e-11, atol=1e-9))
        print('kernel=%d dbg=%d cap=%3d wgs=%d wgs2=%d  %-7s %7.2f ms  ties %d  same-as-generic %s' % (kern, dbg, lanes, wgs, ent, name, best * 1e3, len(r), ok), flush=True)
# round 5: the initial z without a `conf` column (thresh < 0): best hits from the score codes alone (k_report_init_codes)
eng.set_option('report_kernel', 1); eng.set_option('report_lanes', 0); eng.set_option('report_wgs2', 0); eng.set_option('report_dbg', 0)
for wgs2 in (0, 1):
    eng.set_option('report_wgs2', wgs2)
    best = 1e9
    for _ in range(3):
        eng.synchronize(); t0 = time.perf_counter()
        sums, r, c = eng.report_colsums(Z_INITIAL, -1.0)
        best = min(best, time.perf_counter() - t0)
    a = ref['initial']
    ok = (np.array_equal(a[0]['exclude'], sums['exclude']) and np.array_equal(a[1], r) and np.array_equal(a[2], c)
          and np.allclose(a[0]['average'], sums['average'], rtol=1e-11, atol=1e-9))
    print('kernel=1 codes only (thresh < 0) wgs2=%d  initial %7.2f ms  ties %d  same-as-generic %s' % (wgs2, best * 1e3, len(r), ok), flush=True)
