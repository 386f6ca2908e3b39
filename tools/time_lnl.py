"""Time tsem_lnl_pass (and the EM pass beside it) on the bench workload: python tools/time_lnl.py [rows] [nnz_row=40] [cols=30000] [k=v ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np
from test_gpu_parity import _synthetic_tl

rows = int(sys.argv[1]) if len(sys.argv) > 1 and '=' not in sys.argv[1] else 50_000_000
kv = dict(a.split('=') for a in sys.argv[1:] if '=' in a)
nnz_row = int(kv.pop('nnz_row', 40))                 # (not an engine option: entries per row of the synthetic matrix)
cols = int(kv.pop('cols', 30000))
opts = tuple((k, int(v)) for k, v in kv.items())
tl = _synthetic_tl(rows, cols, nnz_row, 'zipf', options=opts)
print({k: tl._eng.layout_info()[k] for k in ('P', 'geometry', 'R', 'lnl_tables', 'lnl_linear', 'value_bytes')})
e = tl._eng
for _ in range(3):
    e.em_pass(); e.em_update()
e.synchronize()
for name, fn in (('em_pass', e.em_pass), ('lnl_pass', e.lnl_pass)):
    fn(); e.synchronize()
    t = time.time()
    for _ in range(5):
        fn()
    e.synchronize()
    print('%s %.3f ms' % (name, (time.time() - t) / 5 * 1e3))
print('lnl', e.read_reduce(e.dims()[1], 1))
