"""Per-member loop time of the fused kernel (FZ_EXPERIMENT build, fused_dbg 4096 [+2048: members free-run]):
TSEM_LIB=build_ab/exp.so python tools/member_times.py [dbg] [value_format]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine
from telescope_amd.likelihood import TelescopeLikelihood
class O: em_epsilon=0.0; max_iter=3; pi_prior=0; theta_prior=200000
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fmt = int(sys.argv[2]) if len(sys.argv) > 2 else 2
eng = Engine(0)
eng.set_option('em_kernel', 2); eng.set_option('value_format', fmt)
eng.generate(0, 50_000_000, 30000, synthetic.poisson_cdf_u32(40), 42, 1, 0.0)
tl = TelescopeLikelihood.from_engine(eng, O())
eng.em_steps(2, False)
eng.set_option('fused_prof', 1); eng.set_option('fused_dbg', dbg)
eng.em_steps(1, False)
t = eng.fused_prof().astype(np.int64).ravel()
info = eng.layout_info(); P = info['P']
n = 256 // P * P
cyc = t[0:2 * n:2].reshape(-1, P); blk = t[1:2 * n:2].reshape(-1, P)
ok = blk[:, 0] > 0
print('dbg', dbg, 'format', fmt, 'teams', ok.sum(), 'P', P)
print('cycles per block, by member (mean over teams):', np.round((cyc[ok] / blk[ok]).mean(0), 1))
print('   min / max over teams per member          :', np.round((cyc[ok] / blk[ok]).min(0), 1), np.round((cyc[ok] / blk[ok]).max(0), 1))
print('slowest / fastest member mean:', (cyc[ok] / blk[ok]).mean(0).max() / (cyc[ok] / blk[ok]).mean(0).min())
