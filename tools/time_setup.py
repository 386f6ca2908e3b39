"""Wall-clock of the setup stages on the bench workload: python tools/time_setup.py [rows] [k=v ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine
from telescope_amd.likelihood import score_lut

rows = int(sys.argv[1]) if len(sys.argv) > 1 and '=' not in sys.argv[1] else 50_000_000
eng = Engine(0)
for a in sys.argv[1:]:
    if '=' in a:
        eng.set_option(a.split('=')[0], int(a.split('=')[1]))
t = [time.perf_counter()]
def lap(name):
    eng.synchronize(); t.append(time.perf_counter()); print('%-28s %8.1f ms' % (name, (t[-1] - t[-2]) * 1e3))
eng.generate(0, rows, 30000, synthetic.poisson_cdf_u32(40), 42, 1, 0.0); lap('generate (synthetic only)')
eng.set_lut(score_lut(eng.max_score())); lap('max score + score table')
stats, pisum0, cnt, hsh = eng.rowstats(); lap('rowstats')
eng.set_model(stats, pisum0, cnt, hsh, 0.0, 200000.0); lap('set_model (layout)')
eng.em_steps(1, False); lap('first EM step')
eng.em_steps(1, False); lap('second EM step')
print(eng.layout_info())
