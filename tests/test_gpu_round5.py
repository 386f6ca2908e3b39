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
