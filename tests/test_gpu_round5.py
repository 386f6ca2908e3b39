"""Round 5, `-m gpu`: regressions for ADVICE r4, the log-table form of both log-likelihood passes, the streaming report
pass on the blocked layout, set-up products after the fused sweeps, the CSR drop.  Everything goes through the C ABI."""
import math
import os

import numpy as np
import pytest

from conftest import GOLD, Opts, case_matrix, case_names, load_case

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _synthetic_tl(rows, cols, d, dist, seed=42, uniq=0.0, options=(), opts=None):
    from telescope_amd import _lib, synthetic
    from telescope_amd.likelihood import TelescopeLikelihood
    eng = _lib.Engine(0)
    for k, v in options:
        eng.set_option(k, v)
    eng.generate(0, rows, cols, synthetic.poisson_cdf_u32(d), seed, synthetic.DIST_CODE[dist], uniq)
    return TelescopeLikelihood.from_engine(eng, opts or Opts(max_iter=5, em_epsilon=0.0))


# ---- per-group sums must not discard the tie list `choose` still needs --------------------------------------

def test_group_sums_between_two_choose_column_sums(gpu_device):
    """reassign_colsums('choose') -> reassign_group_sums('exclude') -> reassign_colsums('choose'): the streaming per-group pass used to
    free the tie list of the last report pass while the Python class still pointed at it (EngineError 'rows == NULL needs n == the
    tie count').  Both `choose` sums must come out, from the same RNG state, as the same numbers; and the group sums must add up."""
    tl = _synthetic_tl(300_000, 5_000, 12, 'zipf', uniq=0.05, opts=Opts(max_iter=4, em_epsilon=0.0))
    tl.em()
    st = np.random.get_state()
    np.random.seed(7)
    first = tl.reassign_colsums('choose', initial=True)       # (the initial z — equal scores — is where rows have several best hits)
    assert len(tl._report_cache) == 1 and next(iter(tl._report_cache.values()))['rows'].size > 0    # there ARE tied rows
    n_groups = 7
    groups = [np.arange(g, tl.N, n_groups) for g in range(n_groups)]
    for method in ('exclude', 'average', 'all', 'conf'):
        gs = tl.reassign_group_sums(method, groups, initial=True)
        tot = tl.reassign_colsums(method, initial=True)
        assert np.allclose(gs.sum(0), tot, rtol=1e-9, atol=1e-9), method
        np.random.seed(7)
        again = tl.reassign_colsums('choose', initial=True)   # on-device tie list: still there
        assert np.array_equal(first, again), method
    np.random.seed(7)
    assert np.array_equal(first, np.asarray(tl.reassign('choose', initial=True).sum(0)).ravel())
    np.random.set_state(st)


def test_group_cache_follows_the_content_of_the_grouping(gpu_device):
    """the layer cache and the resident device map were keyed on the IDENTITY of `group_rows`; a caller that refills
    the same list got the old grouping's sums."""
    tl = _synthetic_tl(100_000, 2_000, 10, 'uniform', opts=Opts(max_iter=3, em_epsilon=0.0))
    tl.em()
    groups = [np.arange(0, 50_000), np.arange(50_000, 100_000)]
    a = tl.reassign_group_sums('exclude', groups)
    groups[0], groups[1] = np.arange(0, 10_000), np.arange(10_000, 100_000)     # same list object, new content
    b = tl.reassign_group_sums('exclude', groups)
    assert not np.array_equal(a, b)
    assert np.array_equal(a.sum(0), b.sum(0))
    fresh = tl.reassign_group_sums('exclude', [np.arange(0, 10_000), np.arange(10_000, 100_000)])
    assert np.array_equal(b, fresh)
    # someone else replaces the engine's map: the class must notice
    tl._eng.set_groups(np.zeros(tl.N, np.int32), 2)
    assert np.array_equal(tl.reassign_group_sums('exclude', groups), b)


def test_em_without_final_lnl_leaves_no_stale_value(gpu_device):
    """em(final_lnl=False) used to keep (and log) the previous run's lnl."""
    tl = _synthetic_tl(50_000, 1_000, 10, 'uniform', opts=Opts(max_iter=3, em_epsilon=0.0))
    tl.em()
    assert math.isfinite(tl.lnl)
    tl.em(final_lnl=False)
    assert math.isnan(tl.lnl)
    tl.em()
    assert math.isfinite(tl.lnl)


# ---- the log-table form of the log-likelihood passes -----------------------------------------------------------------

def test_log1p_from_log_tables(gpu_device):
    """fz_log1p_of_log: log1p(Q c) from log Q + log c.  What a sum of z * log1p needs is ABSOLUTE accuracy relative to the size of
    the terms that dominate (20-100): <= 1e-15 * max(1, |log1p|) over the whole range — the fast range (Q c >= 2^27: L + exp(-L); the
    error is the rounding of the two table entries, half an ulp of |log Q| <= 100 and of |log c| each, i.e. <= 1e-14 absolute on
    terms of 18.7 and more), the exact middle range, and 0 below e^-40 (absolute error < 4.3e-18)."""
    from telescope_amd import _lib
    rng = np.random.RandomState(3)
    q = np.concatenate([np.exp(rng.uniform(46, 100, 300000)), np.exp(rng.uniform(-7, 100, 300000)), np.exp(rng.uniform(46, 100, 50))])
    c = np.concatenate([10.0 ** rng.uniform(-12, 0, 300000), 10.0 ** rng.uniform(-300, 0, 300000), np.zeros(50)])
    y = _lib.debug_log1p_of_log(q, c)
    x = (q.astype(np.longdouble) * c.astype(np.longdouble))
    ref = np.log1p(x).astype(np.float64)
    assert np.all(np.isfinite(y))
    err = np.abs(y - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= 1e-15, (err.max(), q[err.argmax()], c[err.argmax()])
    # around the two range boundaries
    L = np.concatenate([np.linspace(18.70, 18.73, 20001), np.linspace(-40.01, -39.99, 20001)])
    qb = np.full(L.shape, np.exp(50.0))
    cb = np.exp(L - 50.0)
    yb = _lib.debug_log1p_of_log(qb, cb)
    refb = np.log1p(qb.astype(np.longdouble) * cb.astype(np.longdouble)).astype(np.float64)
    assert (np.abs(yb - refb) / np.maximum(1.0, np.abs(refb))).max() <= 1e-15
    assert np.all(y[-50:] == 0.0)                                  # pi * theta == 0: log 0 = -inf, the term is 0


@pytest.mark.parametrize('fmt', [1, 2])      # option value_format: 1 = fp64 entries, 2 = score codes
@pytest.mark.parametrize('rows,cols,d', [(400_000, 30_000, 40), (300_000, 9_000, 14), (200_000, 50_000, 60)])
def test_lnl_pass_with_log_tables_equals_the_per_entry_logarithm(gpu_device, fmt, rows, cols, d):
    """The dedicated lnl pass (model.py:744-760) with log tables (fused kernel MODE 9) against the same pass with a logarithm per
    entry (MODE 1, fused_dbg bit 13) and against the C oracle — both entry formats; the layouts where the tables do not fit the LDS
    keep MODE 1 (layout_info says which ran)."""
    from oracle import em_fused as oc
    from telescope_amd._lib import EngineError
    vals = {}
    for dbg in (0, 8192):
        tl = _synthetic_tl(rows, cols, d, 'zipf', uniq=0.05, options=(('value_format', fmt), ('fused_dbg', dbg)), opts=Opts(max_iter=6, em_epsilon=0.0))
        tl.em()
        info = tl._eng.layout_info()
        assert info['fused'] == 1 and info['value_bytes'] == (8 if fmt == 1 else 2)
        vals[dbg] = (tl.lnl, info.get('lnl_tables', 0))
        if dbg == 0:
            ip, ix, rw = tl._eng.export_csr()
            ref = oc.em_fused_arrays(ip, ix, rw, cols, 0, 200000, 0.0, 6)
    assert vals[8192][1] == 0
    assert abs(vals[0][0] - ref['lnl']) <= RTOL * abs(ref['lnl'])
    assert abs(vals[0][0] - vals[8192][0]) <= 1e-13 * abs(vals[8192][0]), vals
    if (rows, cols, d) == (400_000, 30_000, 40):
        assert vals[0][1] > 0, 'the bench geometry (40 per row, K = 30k) leaves room for the log tables'


def test_lnl_log_tables_with_dying_and_dead_columns(gpu_device):
    """pi_prior = 0 lets loci die: pi * theta falls through the exact middle range (Q c < 2^27) to 0.  A long run on a matrix with
    many weakly supported loci, checked against the C oracle at several points of the decay (lnl per iteration under
    `use_likelihood` runs the dedicated pass when the carrying layout is not asked for)."""
    from oracle import em_fused as oc
    rows, cols = 60_000, 3_000
    tl = _synthetic_tl(rows, cols, 8, 'zipf', uniq=0.3, options=(('value_format', 1),), opts=Opts(max_iter=400, em_epsilon=0.0, theta_prior=0))
    ip, ix, rw = tl._eng.export_csr()
    for it in (5, 60, 400):
        tl.max_iter = it
        tl._eng.set_params(np.repeat(1. / cols, cols), np.repeat(1. / cols, cols))
        tl.em()
        ref = oc.em_fused_arrays(ip, ix, rw, cols, 0, 0, 0.0, it)
        assert abs(tl.lnl - ref['lnl']) <= RTOL * abs(ref['lnl']), (it, tl.lnl, ref['lnl'])
    assert tl.pi.min() < 1e-30                                       # columns did die
