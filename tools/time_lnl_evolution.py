"""The lnl pass as the parameters evolve: time of tsem_lnl_pass after 5, 20, 50, 100, 200, 400 EM iterations — as the library runs it
(the device picks the form per pass), with the log tables forced (fused MODE 9, fused_dbg 32768) and with the per-entry logarithm
forced (MODE 1, fused_dbg 8192).  Columns on their way to pi = 0 pass through the range where the log form takes its exact branch.  python tools/time_lnl_evolution.py [rows=10000000] [cols=30000] [nnz_row=40] [value_format=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np
from test_gpu_parity import _synthetic_tl

kv = dict(a.split('=') for a in sys.argv[1:] if '=' in a)
rows, cols, d = int(kv.pop('rows', 10_000_000)), int(kv.pop('cols', 30000)), int(kv.pop('nnz_row', 40))
opts = tuple((k, int(v)) for k, v in kv.items())
tl = _synthetic_tl(rows, cols, d, 'zipf', options=opts)
e = tl._eng
print({k: e.layout_info()[k] for k in ('P', 'geometry', 'R', 'value_bytes')}, flush=True)
done = 0
for target in (5, 20, 50, 100, 200, 400):
    e.em_steps(target - done, False); done = target
    out = []
    for dbg in (0, 32768, 8192):
        e.set_option('fused_dbg', dbg)
        e.lnl_pass(); e.synchronize()
        t = time.time()
        for _ in range(5):
            e.lnl_pass()
        e.synchronize()
        out.append((time.time() - t) / 5 * 1e3)
    e.set_option('fused_dbg', 0)
    e.lnl_pass(); e.synchronize()                              # (layout_info reports the count of the LAST pass: an armed one)
    pi, th = e.get_params()
    c = pi * th
    with np.errstate(divide='ignore'):
        lc = np.log(c)
    info = e.layout_info()
    print('after %3d iterations: lnl pass %.3f ms as shipped (entries counted for the exact branch: %d, limit %d) | %.3f ms log tables forced | %.3f ms per-entry logarithm;  '
          'columns with log(pi*theta) < -28: %d of %d' % (target, out[0], info['lnl_mid_entries'], info['lnl_mid_limit'], out[1], out[2], int((lc < -28).sum()), len(c)), flush=True)
