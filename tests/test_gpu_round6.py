"""Round 6, `-m gpu`: near-ties of the final z decided with the reference's own row sum (np.add.reduceat order), the CSR column ids
staying dropped through a report, the C host of the boundary, the config presets of bench.py.  Everything goes through the C ABI."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLD, ROOT, Opts

pytestmark = pytest.mark.gpu


def _near_tie_matrix(seed, n=6000, k=400, max_len=40, long_rows=0, scores=(3, 4)):
    """A matrix whose rows are full of near-ties once the parameters below are set: columns come in PAIRS (2j, 2j + 1) whose
    pi * theta differ by one to three ulp, a row takes both columns of a pair with the same score, so its two largest z values are
    a few ulp apart — whether they round together hangs on the last bit of 1 / rowsum, i.e. on the ORDER the row is added in."""
    rng = np.random.RandomState(seed)
    lens = rng.randint(1, max_len // 2 + 1, n) * 2
    lens[rng.rand(n) < 0.1] = 1
    if long_rows:
        lens[rng.choice(n, long_rows, replace=False)] = rng.choice([130, 258, 300, 398], long_rows)    # beyond the streaming kernel's 256 entries too
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.empty(indptr[-1], np.int32)
    data = np.empty(indptr[-1], np.uint16)
    for i, l in enumerate(lens):
        s = indptr[i]
        if l == 1:
            indices[s] = rng.randint(k); data[s] = rng.choice(scores)
            continue
        pairs = np.sort(rng.choice(k // 2, l // 2, replace=False))
        indices[s:s + l] = np.repeat(2 * pairs, 2) + np.tile([0, 1], l // 2)
        data[s:s + l] = np.repeat(rng.choice(scores, l // 2), 2)
    raw = sp.csr_matrix((data, indices, indptr), shape=(n, k))
    pi = rng.dirichlet(np.full(k, 2.0))
    theta = rng.dirichlet(np.full(k, 2.0))
    for j in range(0, k, 2):                                # pi * theta of a pair: equal, or one to three ulp apart
        theta[j + 1] = theta[j]
        p = pi[j]
        for _ in range(int(rng.randint(0, 4))):
            p = np.nextafter(p, 1.0)
        pi[j + 1] = p
    return raw, pi, theta


@pytest.mark.parametrize('seed,kw', [(1, {}), (2, dict(max_len=8)), (3, dict(long_rows=40, n=3000)), (4, dict(scores=(1, 2), k=64, max_len=60)),
                                     (5, dict(max_len=250, n=1500, k=600))])
@pytest.mark.parametrize('report_kernel', [1, 0])
def test_near_ties_are_decided_with_the_references_row_sum(gpu_device, seed, kw, report_kernel):
    """with the SAME parameters on both sides every integer output of the final z equals the oracle's bit for bit —
    column sums of exclude / choose / unique / all, best-hit counts, the masks — on matrices built so that thousands of rows sit on
    near-ties (two z values one to three ulp apart).  The streaming report kernel defers such rows to k_report_slow, the generic row
    pass to its FIX launch; both redo the row sum in np.add.reduceat's order (tsem_npsum.h) and count the rows they redid."""
    from oracle.telescope_oracle import OracleModel, binmax_rows
    from telescope_amd import _lib
    from telescope_amd.likelihood import TelescopeLikelihood, score_lut
    raw, pi, theta = _near_tie_matrix(seed, **kw)
    n, k = raw.shape
    eng = _lib.Engine(0)
    eng.set_option('report_kernel', report_kernel)
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), k, score_lut(int(raw.data.max())))
    tl = TelescopeLikelihood.from_engine(eng, Opts(max_iter=1, em_epsilon=0.0))
    tl._raw = raw
    eng.set_params(pi, theta)
    om = OracleModel(raw, 0, 200000)
    om.z = om.estep(pi, theta)
    zo = sp.csr_matrix(om.z)
    # a threshold that IS a z value of many rows (conf compares with >=): the largest z of some row
    zmax = zo.max(1).toarray().ravel()
    thresh = float(np.sort(zmax[zmax > 0.5])[len(zmax[zmax > 0.5]) // 2]) if np.any(zmax > 0.5) else 0.9
    for th in (0.9, thresh):
        sums, tie_rows, tie_cnt = eng.report_colsums(_lib.Z_CUR, th)
        want_ex = np.asarray(om.reassign('exclude', th).sum(0)).ravel()
        assert np.array_equal(sums['exclude'].astype(np.int64), np.rint(want_ex).astype(np.int64)), ('exclude', th)
        assert np.allclose(sums['average'], np.asarray(om.reassign('average', th).sum(0)).ravel(), rtol=1e-12, atol=1e-9)
        assert np.allclose(sums['conf'], np.asarray(om.reassign('conf', th).sum(0)).ravel(), rtol=1e-12, atol=1e-9), ('conf', th)
        nb = np.diff(binmax_rows(zo).indptr)
        assert np.array_equal(tie_rows, np.flatnonzero(nb > 1)) and np.array_equal(tie_cnt, nb[nb > 1])
    assert np.array_equal(eng.best_counts(_lib.Z_CUR), np.diff(binmax_rows(zo).indptr))
    for m in ('exclude', 'average', 'conf', 'all', 'unique'):
        cs, mask = eng.reassign(m, thresh, _lib.Z_CUR, want_mask=True)
        mo = sp.csr_matrix(om.reassign(m, thresh)).astype(np.float64)
        dense_mask = sp.csr_matrix((mask, raw.indices, raw.indptr), shape=raw.shape)
        dense_mask.eliminate_zeros()
        d = (dense_mask - mo)
        assert abs(d).max() <= 1e-12 if d.nnz else True, (m, abs(d).max())
        assert np.allclose(cs, np.asarray(mo.sum(0)).ravel(), rtol=1e-12, atol=1e-9), m
    redone = eng.layout_info()['near_tie_rows']
    assert redone > n // 20, redone                         # the matrix IS full of them
    eng.close()


def test_no_near_tie_rows_on_the_bundled_matrix(gpu_device):
    """The other side of the band: ordinary data has no row inside it — the bundled `telescope test` matrix through em() and all
    reports redoes nothing (and stays bit-exact against its golden, tests/test_gpu_parity.py)."""
    from telescope_amd.likelihood import TelescopeLikelihood
    f = np.load(os.path.join(GOLD, 'bundled_raw_scores.npz'))
    raw = sp.csr_matrix((f['data'], f['indices'], f['indptr']), shape=tuple(f['shape']))
    tl = TelescopeLikelihood(raw, Opts(), device=0)
    tl.em()
    for m in ('exclude', 'average', 'conf', 'all', 'unique'):
        tl.reassign_colsums(m)
        sp.csr_matrix(tl.reassign(m))
    assert tl._eng.layout_info()['near_tie_rows'] == 0
