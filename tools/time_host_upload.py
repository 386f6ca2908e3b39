"""PCIe-inclusive cost of the boundary: tsem_load_scores takes HOST arrays (borrowed for the call) and copies them to
HBM.  Times that copy for a host CSR of the bench's shape and reports what it adds to a run.
python tools/time_host_upload.py [rows]      (default 10M rows x ~40 entries = 2.5 GB of host arrays)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from telescope_amd import synthetic
from telescope_amd._lib import Engine
from telescope_amd.likelihood import score_lut

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
src = Engine(0)
src.generate(0, rows, 30000, synthetic.poisson_cdf_u32(40), 42, 1, 0.0)
indptr, indices, raw = src.export_csr()                      # the same matrix as host arrays
lut = score_lut(src.max_score())
src.close()
nbytes = indptr.nbytes + indices.nbytes + raw.nbytes
eng = Engine(0)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    eng.load_scores(indptr, indices, raw, 30000, lut)
    eng.synchronize()
    best = min(best, time.perf_counter() - t0)
print('tsem_load_scores %.1f ms = %.1f GB/s host -> HBM (copy + validation on the device)' % (best * 1e3, nbytes / best / 1e9))
print('rows %d  entries %d  host arrays %.2f GB (int64 indptr, int32 indices, uint16 scores; pageable numpy memory)'
      % (rows, len(indices), nbytes / 1e9))
stats, pisum0, cnt, hsh = eng.rowstats()
eng.set_model(stats, pisum0, cnt, hsh, 0.0, 200000.0)
eng.em_steps(3, False); eng.synchronize()
t0 = time.perf_counter(); eng.em_steps(20, False); eng.synchronize(); it = (time.perf_counter() - t0) / 20
print('EM iteration on the resident matrix: %.3f ms  -> the upload costs as much as %.0f iterations; '
      'a 27-iteration run: %.3e nnz/s resident, %.3e nnz/s with the upload inside the clock'
      % (it * 1e3, best / it, len(indices) / it, 27 * len(indices) / (27 * it + best)))
