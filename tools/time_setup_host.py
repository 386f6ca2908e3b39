"""Set-up on the REAL path in a fresh process: a host CSR (numpy generator) -> tsem_load_scores -> max score / table -> rowstats ->
set_model, wall clock per stage (TSEM_TRACE=1 adds the library's laps).  Shows where a process's one-off runtime costs land when
the matrix comes from the host: python tools/time_setup_host.py [rows=2000000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine
from telescope_amd.likelihood import score_lut

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
ip, ix, rw = synthetic.generate(rows, 30000, 40.0, seed=42, dist='zipf', uniq_frac=0.0)
t = [time.perf_counter()]
def lap(name, eng=None):
    if eng is not None: eng.synchronize()
    t.append(time.perf_counter()); print('%-34s %8.1f ms' % (name, (t[-1] - t[-2]) * 1e3), flush=True)
eng = Engine(0); lap('Engine(0)')
eng.load_scores(ip, ix, rw, 30000, None); lap('load_scores (copy + validation)', eng)
mx = eng.max_score(); eng.set_lut(score_lut(mx)); lap('max score + score table', eng)
st = eng.rowstats(); lap('rowstats', eng)
eng.set_model(st[0], st[1], st[2], st[3], 0.0, 200000.0); lap('set_model (layout)', eng)
eng.em_steps(1, False); lap('first EM step', eng)
