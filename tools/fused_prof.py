"""Per-block timeline of the fused EM kernel (team 0, member 0): python tools/fused_prof.py [rows] [nnz_row=40] [option=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine
from telescope_amd.likelihood import TelescopeLikelihood

class O: em_epsilon=0.0; max_iter=3; pi_prior=0; theta_prior=200000
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
nnz_row = 40
cols = 30000
eng = Engine(0)
eng.set_option('em_kernel', 2)
for kv in sys.argv[2:]:
    k, v = kv.split('=')
    if k == 'nnz_row':
        nnz_row = int(v)
    elif k == 'cols':
        cols = int(v)
    else:
        eng.set_option(k, int(v))
eng.generate(0, rows, cols, synthetic.poisson_cdf_u32(nnz_row), 42, 1, 0.0)
tl = TelescopeLikelihood.from_engine(eng, O())
eng.em_steps(2, False)
eng.set_option('fused_prof', 1)
eng.em_steps(1, False)
t = eng.fused_prof().astype(np.int64)
print(eng.layout_info())
names = ['start', 'ops gathered', 'burst issued', 'P1 atomics', 'barrier', 'x:combined', 'x:start', 'x:issued', 'w13 pre-wait', 'x1:combined', 'x:published']
base = t[4, 0]
print('cycles relative to block start (blocks 4..11); clock ~2.1-2.4 GHz (shader clock / s_memtime)')
print('%-6s' % 'blk' + ''.join('%13s' % n for n in names) + '%12s' % 'blk total')
for i in range(4, 24):
    row = t[i, :11] - t[i, 0]
    print('%-6d' % i + ''.join('%13d' % v for v in row) + '%12d' % (t[i + 1, 0] - t[i, 0]))
