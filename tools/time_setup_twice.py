"""The set-up stages twice in ONE process: what the first engine pays once per process (code objects are loaded on a
kernel's first launch) against what every further engine pays.   python tools/time_setup_twice.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from telescope_amd import synthetic
from telescope_amd._lib import Engine
from telescope_amd.likelihood import score_lut
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
for rep in (1, 2):
    eng = Engine(0)
    t = [time.perf_counter()]
    def lap(name):
        eng.synchronize(); t.append(time.perf_counter()); print('engine %d  %-28s %8.1f ms' % (rep, name, (t[-1] - t[-2]) * 1e3), flush=True)
    eng.generate(0, rows, 30000, synthetic.poisson_cdf_u32(40), 42, 1, 0.0); lap('generate (synthetic only)')
    eng.set_lut(score_lut(eng.max_score())); lap('max score + score table')
    stats, pisum0, cnt, hsh = eng.rowstats(); lap('rowstats')
    eng.set_model(stats, pisum0, cnt, hsh, 0.0, 200000.0); lap('set_model (layout)')
    eng.em_steps(1, False); lap('first EM step')
    eng.em_steps(1, False); lap('second EM step')
    l = eng.final_lnl(); lap('final lnl')
    l = eng.final_lnl(); lap('final lnl again')
    eng.close()
