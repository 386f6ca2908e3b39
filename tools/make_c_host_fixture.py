#!/usr/bin/env python3
"""tests/golden/bundled_flat.bin for the C host of the boundary (tests/c_host/run_bundled.c): the bundled `telescope test` matrix of
tests/golden/bundled_raw_scores.npz as one flat little-endian file — int64 {rows, columns, stored entries, table length} | int64
indptr[rows + 1] | int32 indices[nnz] | uint16 scores[nnz] | zero padding to 8 bytes | float64 Q table[max score + 1], the table computed
with the reference's numpy expression (telescope_amd.likelihood.score_lut = model.py:653), so that a host without numpy installs the
reference's bits."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from telescope_amd.likelihood import score_lut  # noqa: E402

f = np.load(os.path.join(ROOT, 'tests', 'golden', 'bundled_raw_scores.npz'))
indptr, indices, data = f['indptr'].astype('<i8'), f['indices'].astype('<i4'), f['data'].astype('<u2')
lut = score_lut(int(data.max())).astype('<f8')
out = os.path.join(ROOT, 'tests', 'golden', 'bundled_flat.bin')
with open(out, 'wb') as fh:
    fh.write(np.array([f['shape'][0], f['shape'][1], data.size, lut.size], '<i8').tobytes())
    fh.write(indptr.tobytes()); fh.write(indices.tobytes()); fh.write(data.tobytes())
    fh.write(b'\0' * ((8 - (2 * data.size) % 8) % 8))
    fh.write(lut.tobytes())
print(out, os.path.getsize(out), 'bytes')
