"""End-to-end wall-clock of one run on the bench workload (device-generated matrix):
setup, em() to max_iter, the seven reassign column sums of output_report IN ITS ORDER (model.py:432-457: final conf,
initial all, unique, initial exclude / choose / average, then the final count column).  python tools/time_e2e.py [rows] [iters]"""
import os, sys, time, logging
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine
from telescope_amd.likelihood import TelescopeLikelihood

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
class O: em_epsilon = 1e-7; max_iter = iters; pi_prior = 0; theta_prior = 200000
t = [time.perf_counter()]
def lap(name, eng):
    eng.synchronize(); t.append(time.perf_counter()); print('%-34s %9.1f ms' % (name, (t[-1] - t[-2]) * 1e3))
eng = Engine(0)
eng.generate(0, rows, 30000, synthetic.poisson_cdf_u32(40), 42, 1, 0.05); lap('generate (synthetic only)', eng)
tl = TelescopeLikelihood.from_engine(eng, O()); lap('setup (table, row stats, layout)', eng)
tl.em(loglev=logging.DEBUG); lap('em(): %d iterations + final lnl' % tl.n_iter, eng)
np.random.seed(1)
for m, init in (('conf', False), ('all', True), ('unique', False), ('exclude', True), ('choose', True), ('average', True), ('exclude', False)):
    tl.reassign_colsums(m, 0.9, init); lap('reassign %-8s initial=%d' % (m, init), eng)
print('total %.1f ms' % ((t[-1] - t[0]) * 1e3))
