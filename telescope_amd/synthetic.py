"""Synthetic fragment x locus score matrices (SURVEY.md section 8(d) spec).

A counter-based hash of (seed, row, slot) defines every entry, so the same
rows can be produced here on the CPU (numpy, <= ~1e6 rows, for parity tests)
and on the device by `tsem_generate` (1e7 - 2e8 rows, for the benchmark); the
two are bit-identical (tests/test_gpu_parity.py::test_device_generator).

Row i of the global matrix:
  * length  len_i = max(1, Poisson(d)), drawn by inverting an integer CDF
    table (`poisson_cdf_u32`) with a 32-bit hash -> exact on both sides;
    with probability `uniq_frac` the row is forced unique (len_i = 1);
  * with probability 5 % the row holds column 0 (`__no_feature`) in slot 0;
  * the other slots draw columns from [1, K) without replacement: slot k,
    attempt a uses u = hash(seed, i, k + 256*a); 'uniform': j = 1+floor((K-1)u),
    'zipf': j = 1+floor((K-1)*u*u*u) (hot-locus skew); a duplicate bumps a;
    'family' (round 4: what multi-mapping inside repeat families looks like): the columns 1 .. are cut into families of FAMILY
    consecutive loci; row i belongs to family f_i = floor(nfam * v^3), v = hash(seed ^ SALT_FAM, i, 0) (hot families), and draws
    all its columns inside it, j = 1 + FAMILY*f_i + floor(FAMILY*u) (columns past the last whole family stay empty);
  * columns are then sorted ascending (canonical CSR);
  * the raw score of sorted position p is 139 + hash(seed^SALT, i, p) % 162,
    i.e. uniform on [139, 300] like the bundled data's range with max AS 300.
"""
import math

import numpy as np

MASK64 = (1 << 64) - 1
GOLDEN = np.uint64(0x9E3779B97F4A7C15)
M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)
SALT_LEN = np.uint64(0xA5A5A5A5A5A5A5A5)
SALT_UNIQ = np.uint64(0x5BD1E9955BD1E995)
SALT_COL0 = np.uint64(0xC2B2AE3D27D4EB4F)
SALT_SCORE = np.uint64(0x165667B19E3779F9)
SALT_FAM = np.uint64(0x27D4EB2F165667C5)
FAMILY = 256               # loci per family of dist 'family' (> MAX_ROW_LEN: a row always fits its family)
SCORE_LO, SCORE_SPAN = 139, 162
MAX_ROW_LEN = 255          # slots per row are capped (k + 256*a addressing)
DIST_CODE = {'uniform': 0, 'zipf': 1, 'family': 2}


def mix64(z):
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over='ignore'):
        z = (z + GOLDEN).astype(np.uint64)
        z = ((z ^ (z >> np.uint64(30))) * M1).astype(np.uint64)
        z = ((z ^ (z >> np.uint64(27))) * M2).astype(np.uint64)
        return z ^ (z >> np.uint64(31))


def hash3(seed, row, k):
    """hash(seed, row, k) -> uint64.  `row`, `k` broadcastable integer arrays."""
    with np.errstate(over='ignore'):
        row = np.asarray(row).astype(np.uint64)
        k = np.asarray(k).astype(np.uint64)
        return mix64(mix64(np.uint64(seed) ^ (row * GOLDEN)) ^ (k * M1))


def poisson_cdf_u32(mean):
    """Integer thresholds T[n] = floor(2^32 * P(X <= n)), n = 0..; the sample
    for a 32-bit hash h is  #{n : T[n] <= h}."""
    out, n = [], 0
    p = math.exp(-mean)
    cdf = p
    while cdf < 1.0 - 1e-12 and n < 4 * int(mean) + 64:
        out.append(min(int(cdf * 4294967296.0), 4294967295))
        n += 1
        p *= mean / n
        cdf += p
    return np.asarray(out, dtype=np.uint32)


def row_lengths(seed, rows, mean_nnz, n_cols, uniq_frac=0.0, cdf=None):
    if cdf is None:
        cdf = poisson_cdf_u32(mean_nnz)
    h = (hash3(np.uint64(seed) ^ SALT_LEN, rows, 0) >> np.uint64(32)).astype(np.uint32)
    lens = np.searchsorted(cdf, h, side='right').astype(np.int64)
    lens = np.clip(lens, 1, min(MAX_ROW_LEN, n_cols - 1))
    if uniq_frac > 0:
        hu = (hash3(np.uint64(seed) ^ SALT_UNIQ, rows, 0) >> np.uint64(32)).astype(np.uint32)
        lens[hu < np.uint32(min(int(uniq_frac * 4294967296.0), 4294967295))] = 1
    return lens


def _draw(seed, rows, slot, n_cols, dist):
    h = hash3(seed, rows, slot)
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    if dist == 'family':
        nfam = (n_cols - 1) // FAMILY
        if nfam < 1:
            raise ValueError("dist='family' needs more than %d columns" % FAMILY)
        v = (hash3(np.uint64(seed) ^ SALT_FAM, rows, 0) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        f = np.floor(nfam * ((v * v) * v)).astype(np.int64)
        return 1 + FAMILY * f + np.floor(FAMILY * u).astype(np.int64)
    if dist == 'zipf':
        u = (u * u) * u
    return 1 + np.floor((n_cols - 1) * u).astype(np.int64)


def generate(n_rows, n_cols, mean_nnz, seed=42, dist='zipf', uniq_frac=0.0,
             row_begin=0, row_end=None):
    """Rows [row_begin, row_end) of the global matrix as CSR arrays.

    Returns (indptr int64[n+1], indices int32[nnz], raw uint16[nnz]).
    """
    if row_end is None:
        row_end = n_rows
    rows = np.arange(row_begin, row_end, dtype=np.int64)
    n = rows.size
    lens = row_lengths(seed, rows, mean_nnz, n_cols, uniq_frac)
    maxlen = int(lens.max()) if n else 0
    cols = np.full((n, maxlen), np.iinfo(np.int64).max, dtype=np.int64)
    has0 = (hash3(np.uint64(seed) ^ SALT_COL0, rows, 0) >> np.uint64(32)
            ).astype(np.uint32) < np.uint32(int(0.05 * 4294967296.0))
    for k in range(maxlen):
        act = np.nonzero(lens > k)[0]
        if act.size == 0:
            break
        if k == 0:
            z = act[has0[act]]
            cols[z, 0] = 0
            act = act[~has0[act]]
        attempt = 0
        while act.size:
            cand = _draw(seed, rows[act], k + 256 * attempt, n_cols, dist)
            dup = (cols[act, :k] == cand[:, None]).any(axis=1) if k else \
                np.zeros(act.size, dtype=bool)
            ok = act[~dup]
            cols[ok, k] = cand[~dup]
            act = act[dup]
            attempt += 1
    cols.sort(axis=1)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=indptr[1:])
    mask = np.arange(maxlen)[None, :] < lens[:, None]
    indices = cols[mask].astype(np.int32)
    pos = np.broadcast_to(np.arange(maxlen)[None, :], (n, maxlen))[mask]
    rrep = np.repeat(rows, lens)
    raw = (SCORE_LO + (hash3(np.uint64(seed) ^ SALT_SCORE, rrep, pos)
                       % np.uint64(SCORE_SPAN))).astype(np.uint16)
    return indptr, indices, raw


def generate_csr(n_rows, n_cols, mean_nnz, **kw):
    """Same as `generate` but wrapped as a scipy uint16 CSR matrix."""
    import scipy.sparse as sp
    indptr, indices, raw = generate(n_rows, n_cols, mean_nnz, **kw)
    return sp.csr_matrix((raw, indices, indptr),
                         shape=(len(indptr) - 1, n_cols))
