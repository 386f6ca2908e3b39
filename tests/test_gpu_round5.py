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
    if fmt == 2:
        # score codes with the reference's own table: log Q = (code / max) * 100 needs no table (layout_info 'lnl_linear'), so even the
        # layouts whose row slots fill the LDS take the log form; fused_dbg bit 14 forces the table look-up where it fits: same value
        tl = _synthetic_tl(rows, cols, d, 'zipf', uniq=0.05, options=(('value_format', fmt), ('fused_dbg', 16384)), opts=Opts(max_iter=6, em_epsilon=0.0))
        tl.em()
        info = tl._eng.layout_info()
        assert info['lnl_linear'] == 1
        assert abs(tl.lnl - vals[0][0]) <= 1e-13 * abs(vals[0][0])


def test_lnl_log_form_on_a_layout_whose_row_slots_fill_the_lds(gpu_device):
    """K = 30k with 12 entries per row: 1152 row slots take the LDS, no room for a log Q table — with score codes and the reference's
    table the pass needs none (log Q = (code / max) * 100).  Against the per-entry logarithm and the C oracle; and a score table that
    is NOT the reference's (rounded to 11 bits) falls back to the table / the per-entry form with the same results."""
    from oracle import em_fused as oc
    rows, cols, d = 600_000, 30_000, 12
    out = {}
    for dbg in (0, 8192):
        tl = _synthetic_tl(rows, cols, d, 'zipf', uniq=0.05, options=(('fused_dbg', dbg),), opts=Opts(max_iter=5, em_epsilon=0.0))
        tl.em()
        info = tl._eng.layout_info()
        assert info['fused'] == 1 and info['value_bytes'] == 2 and info['geometry'] >= 2, info
        out[dbg] = (tl.lnl, info)
        if dbg == 0:
            ip, ix, rw = tl._eng.export_csr()
            ref = oc.em_fused_arrays(ip, ix, rw, cols, 0, 200000, 0.0, 5)
    assert out[0][1]['lnl_linear'] == 1 and out[0][1]['lnl_tables'] > 0
    assert abs(out[0][0] - ref['lnl']) <= RTOL * abs(ref['lnl']) and abs(out[0][0] - out[8192][0]) <= 1e-13 * abs(out[0][0])
    # a rounded score table: the arithmetic form is refused by the host's check
    from telescope_amd import _lib, synthetic
    from telescope_amd.likelihood import TelescopeLikelihood
    eng = _lib.Engine(0)
    eng.generate(0, 200_000, 5_000, synthetic.poisson_cdf_u32(20), 42, synthetic.DIST_CODE['zipf'], 0.05)
    t2 = TelescopeLikelihood.from_engine(eng, Opts(max_iter=4, em_epsilon=0.0), lut_mantissa_bits=11)
    t2.em()
    i2 = eng.layout_info()
    assert i2['lnl_linear'] == 0 and i2['lnl_tables'] > 0
    eng3 = _lib.Engine(0)
    eng3.set_option('fused_dbg', 8192)
    eng3.generate(0, 200_000, 5_000, synthetic.poisson_cdf_u32(20), 42, synthetic.DIST_CODE['zipf'], 0.05)
    t3 = TelescopeLikelihood.from_engine(eng3, Opts(max_iter=4, em_epsilon=0.0), lut_mantissa_bits=11)
    t3.em()
    assert abs(t2.lnl - t3.lnl) <= 1e-13 * abs(t3.lnl)


def test_lnl_log_tables_with_dying_and_dead_columns(gpu_device):
    """pi_prior = 0 lets loci die: pi * theta falls through the exact middle range (Q c < 2^27) to 0.  A long run on a matrix with
    many weakly supported loci, checked against the C oracle at several points of the decay (lnl per iteration under
    `use_likelihood` runs the dedicated pass when the carrying layout is not asked for)."""
    from oracle import em_fused as oc
    rows, cols = 60_000, 3_000
    # (fused_dbg bit 15: the log form whatever the parameters look like — left alone, the device switches to the per-entry logarithm
    #  once many entries sit in dying columns, test_lnl_pass_picks_its_form_on_the_device)
    tl = _synthetic_tl(rows, cols, 8, 'zipf', uniq=0.3, options=(('value_format', 1), ('fused_dbg', 32768)), opts=Opts(max_iter=400, em_epsilon=0.0, theta_prior=0))
    ip, ix, rw = tl._eng.export_csr()
    for it in (5, 60, 400):
        tl.max_iter = it
        tl._eng.set_params(np.repeat(1. / cols, cols), np.repeat(1. / cols, cols))
        tl.em()
        ref = oc.em_fused_arrays(ip, ix, rw, cols, 0, 0, 0.0, it)
        assert abs(tl.lnl - ref['lnl']) <= RTOL * abs(ref['lnl']), (it, tl.lnl, ref['lnl'])
    assert tl.pi.min() < 1e-30                                       # columns did die


def test_lnl_choice_when_the_arithmetic_log_q_would_force_the_exact_branch(gpu_device):
    """Score codes below 0.38 x max (t < 38) are outside the range where log Q = (code / max) * 100 holds: the log form would send every
    such entry through its exact branch.  Where a log Q table fits the LDS the library takes the look-up instead; on a layout whose row
    slots fill the LDS (K = 30k, 12 per row) every live column counts towards the choice and the per-entry logarithm runs.  Same lnl as
    the forced forms and the C oracle either way."""
    import scipy.sparse as sp
    from oracle import em_fused as oc
    from telescope_amd import synthetic
    from telescope_amd.likelihood import TelescopeLikelihood
    for rows, cols, d, want_linear in ((600_000, 30_000, 12, 1), (200_000, 5_000, 20, 0)):
        ip, ix, rw = synthetic.generate(rows, cols, float(d), seed=42, dist='zipf', uniq_frac=0.05)
        rw = rw.copy()
        rw[::97] = 40                                             # t = 40 / 300 * 100 = 13: far below 38
        raw = sp.csr_matrix((rw, ix, ip), shape=(rows, cols))
        vals = {}
        for dbg in (0, 8192, 32768):
            tl = TelescopeLikelihood(raw, Opts(max_iter=4, em_epsilon=0.0), device=0, engine_options={'fused_dbg': dbg})
            tl.em()
            vals[dbg] = (tl.lnl, tl._eng.layout_info())
        info = vals[0][1]
        assert info['fused'] == 1 and info['lnl_tables'] > 0 and info['lnl_linear'] == want_linear, info
        if want_linear:                                           # no room for the table: every live column counts, MODE 1 runs
            assert info['lnl_mid_entries'] > info['lnl_mid_limit'] > 0, info
        ref = oc.em_fused_arrays(ip, ix, rw, cols, 0, 200000, 0.0, 4)
        assert abs(vals[0][0] - ref['lnl']) <= RTOL * abs(ref['lnl'])
        assert abs(vals[0][0] - vals[8192][0]) <= 1e-13 * abs(vals[0][0]) and abs(vals[32768][0] - vals[8192][0]) <= 1e-13 * abs(vals[0][0]), vals


def test_lnl_pass_picks_its_form_on_the_device(gpu_device):
    """The log form of the lnl pass stalls on its exact branch (pi * theta fetched from global memory) when columns are on their way to
    pi = 0; k_log_tab counts the stored entries of such columns before every pass and the two forms of the pass read the count: one of
    them returns at once.  Fresh parameters: nothing counted, the log form runs.  After a long run with pi_prior = theta_prior = 0:
    more than the limit, the per-entry logarithm runs.  Same lnl either way (both forced forms against each other and the C oracle)."""
    from oracle import em_fused as oc
    rows, cols = 120_000, 6_000
    fresh = _synthetic_tl(rows, cols, 10, 'zipf', uniq=0.3, opts=Opts(max_iter=3, em_epsilon=0.0))      # the reference's default priors
    fresh.em()
    i0 = fresh._eng.layout_info()
    assert i0['lnl_tables'] > 0 and 0 <= i0['lnl_mid_entries'] <= i0['lnl_mid_limit'], i0   # the log form ran
    vals = {}
    for dbg in (0, 8192, 32768):
        tl = _synthetic_tl(rows, cols, 10, 'zipf', uniq=0.3, options=(('fused_dbg', dbg),), opts=Opts(max_iter=300, em_epsilon=0.0, theta_prior=0))
        tl.em()
        vals[dbg] = (tl.lnl, tl._eng.layout_info())
        if dbg == 0:
            ip, ix, rw = tl._eng.export_csr()
    l0, j0 = vals[0]
    assert j0['lnl_mid_entries'] > j0['lnl_mid_limit'] > 0, j0                     # dying columns: the per-entry logarithm took over
    assert vals[8192][1]['lnl_mid_entries'] == -1 and vals[32768][1]['lnl_mid_entries'] == -1   # forced forms: no choice armed
    a, b, c = vals[0][0], vals[8192][0], vals[32768][0]
    assert abs(a - b) <= 1e-13 * abs(b) and abs(c - b) <= 1e-13 * abs(b), (a, b, c)
    ref = oc.em_fused_arrays(ip, ix, rw, cols, 0, 0, 0.0, 300)
    assert abs(l0 - ref['lnl']) <= RTOL * abs(ref['lnl'])
    assert tl.pi.min() < 1e-30


# ---- option drop_csr_indices -----------------------------------------------------------------------------------------

def test_everything_still_works_after_the_csr_column_ids_are_dropped(gpu_device):
    """`drop_csr_indices` = 1 frees the CSR column ids once the blocked layout exists (4 of 14 B per stored entry): the EM and the
    streaming report pass never read them; z export, the generic row passes, `export_csr`, the look-ups and a layout rebuild (the
    fall-back after a time-out, `use_likelihood`) get them back from the 2-byte popularity ids.  Same results as with the ids kept."""
    from telescope_amd._lib import Z_PREV
    res = {}
    for drop in (0, 1):
        tl = _synthetic_tl(200_000, 9_000, 16, 'zipf', uniq=0.05, options=(('drop_csr_indices', drop),), opts=Opts(max_iter=6, em_epsilon=0.0))
        tl.em()
        np.random.seed(5)
        out = dict(lnl=tl.lnl, pi=tl.pi.copy())
        for m in ('exclude', 'average', 'conf', 'unique', 'all', 'choose'):
            out[m] = tl.reassign_colsums(m)                   # ('unique' / 'choose' take the generic row pass)
        out['z'] = tl._eng.export_z(Z_PREV)
        out['csr'] = tl._eng.export_csr()
        out['estep'] = tl.estep(tl.pi, tl.theta).data
        pr, va = tl.lookup(np.arange(0, 200_000, 997), np.zeros(201, np.int64), 'exclude')
        out['lookup'] = (pr, va)
        tl.em(use_likelihood=True)                             # rebuilds the layout for the carrying pass (needs the column ids again)
        out['lnl2'], out['iters2'] = tl.lnl, tl.n_iter
        res[drop] = out
    a, b = res[0], res[1]
    assert abs(a['lnl'] - b['lnl']) <= 1e-12 * abs(a['lnl']) and np.allclose(a['pi'], b['pi'], rtol=1e-11, atol=0)
    for m in ('exclude', 'unique', 'all', 'choose'):
        assert np.array_equal(a[m], b[m]), m
    for m in ('average', 'conf'):
        assert np.allclose(a[m], b[m], rtol=1e-10, atol=1e-10), m
    assert np.allclose(a['z'], b['z'], rtol=1e-10, atol=0) and np.allclose(a['estep'], b['estep'], rtol=1e-10, atol=0)
    for x, y in zip(a['csr'], b['csr']):
        assert np.array_equal(x, y)
    assert np.allclose(a['lookup'][0], b['lookup'][0], rtol=1e-10, atol=0) and np.array_equal(a['lookup'][1], b['lookup'][1])
    assert a['iters2'] == b['iters2'] and abs(a['lnl2'] - b['lnl2']) <= 1e-11 * abs(a['lnl2'])


# ---- BASELINE config 5 at half its size on ONE GPU ---------------------------------------------------------------------

def test_half_of_config5_on_one_gpu(gpu_device):
    """100M fragments x 50k loci x ~100 per row = 1.0e10 stored entries — one half of BASELINE config 5 (200M x 50k x ~100 on 8 GPUs; a
    GPU of that run holds an eighth) — resident on ONE MI355X and run through the properties that do not need a CPU oracle at this
    size: the EM runs on the persistent fused kernel without a fall-back, pi and theta stay distributions, `all` counts every stored
    entry, `exclude` + the tied rows account for every fragment, the streaming report pass equals the generic one on a sample of
    columns, and four iterations on the two-pass kernels (rebuilt layout, fp64 entries) from the same start agree to 1e-10.  Skips when
    less than 200 GB of HBM is free (a shared or smaller device)."""
    from telescope_amd import _lib
    free, total = _lib.device_memory(0)
    if free < 200 * 2 ** 30:
        pytest.skip('needs 200 GB of free HBM, %.0f GB free of %.0f' % (free / 2 ** 30, total / 2 ** 30))
    rows, cols, d, iters = 100_000_000, 50_000, 100, 4
    tl = _synthetic_tl(rows, cols, d, 'zipf', uniq=0.0, opts=Opts(max_iter=iters, em_epsilon=0.0))
    eng = tl._eng
    info = eng.layout_info()
    nnz = eng.dims()[2]
    assert abs(nnz - 1.0e10) < 1e8 and info['fused'] == 1 and info['value_bytes'] == 2 and info['P'] == 8, info
    mem = eng.device_memory()
    assert mem['resident']['csr_indices'] == 0                    # dropped automatically from 4e9 entries on: 10 B per entry resident
    assert sum(mem['resident'].values()) < 11.0 * nnz + 40 * rows, mem
    tl.em()
    assert eng.layout_info()['fallbacks'] == 0 and math.isfinite(tl.lnl)
    pi_f, theta_f, lnl_f = tl.pi.copy(), tl.theta.copy(), tl.lnl
    assert abs(pi_f.sum() - 1.0) < 1e-11 and abs(theta_f.sum() - 1.0) < 1e-11 and (pi_f >= 0).all() and (theta_f >= 0).all()
    assert int(tl.reassign_colsums('all', initial=True).sum()) == nnz
    excl = tl.reassign_colsums('exclude')
    rep = next(iter(tl._report_cache.values()))
    assert int(excl.sum()) + len(rep['rows']) == rows              # every fragment has one best hit, or is in the tie list
    avg = tl.reassign_colsums('average')
    assert abs(avg.sum() - rows) < 1e-6 * rows
    # the same four iterations on the two-pass kernels: the layout is rebuilt with fp64 entries (the column ids come back from the
    # popularity ids first), the parameters restart from 1 / K
    eng.fallback_twopass()
    info2 = eng.layout_info()
    assert info2['fused'] == 0 and info2['value_bytes'] == 8, info2
    eng.set_params(np.repeat(1. / cols, cols), np.repeat(1. / cols, cols))
    tl.em()
    assert np.allclose(tl.pi, pi_f, rtol=1e-10, atol=1e-300) and np.allclose(tl.theta, theta_f, rtol=1e-10, atol=1e-300)
    assert abs(tl.lnl - lnl_f) <= 1e-10 * abs(lnl_f)
    assert np.array_equal(tl.reassign_colsums('exclude'), excl)


# ---- the initial z's report pass on the score codes alone ---------------------------------------------------------------

@pytest.mark.parametrize('rows,cols,d,dist,uniq', [(400_000, 30_000, 40, 'zipf', 0.05), (300_000, 5_000, 9, 'uniform', 0.3),
                                                   (200_000, 50_000, 100, 'zipf', 0.0), (100_000, 2_000, 150, 'uniform', 0.0)])
def test_initial_report_pass_from_the_score_codes(gpu_device, rows, cols, d, dist, uniq):
    """`tsem_report_colsums(initial, thresh < 0)` — no `conf` column wanted, what `output_report` asks of the initial z — finds a row's
    best hits as its largest score codes (k_report_init_codes: integer work on packed 16-bit codes, no score-table look-up, no
    floating point).  `exclude`, the tie list (rows and numbers of best hits) bit for bit, `average` to rounding, against the full pass
    (k_report_rows) and the generic row pass; rows longer than the lane capacity go through the slow kernel in both."""
    from telescope_amd._lib import Z_INITIAL
    tl = _synthetic_tl(rows, cols, d, dist, uniq=uniq, opts=Opts(max_iter=2, em_epsilon=0.0))
    eng = tl._eng
    full, r1, c1 = eng.report_colsums(Z_INITIAL, 0.9)
    fast, r2, c2 = eng.report_colsums(Z_INITIAL, -1.0)
    assert np.array_equal(fast['exclude'], full['exclude']) and np.array_equal(r1, r2) and np.array_equal(c1, c2)
    assert np.allclose(fast['average'], full['average'], rtol=1e-12, atol=1e-9)
    assert len(r1) > 0 and full['exclude'].sum() + len(r1) == rows - np.count_nonzero(np.diff(eng.export_csr()[0]) == 0)
    eng.set_option('report_kernel', 0)
    gen, r3, c3 = eng.report_colsums(Z_INITIAL, 0.9)
    assert np.array_equal(fast['exclude'], gen['exclude']) and np.array_equal(r2, r3) and np.array_equal(c2, c3)
    eng.set_option('report_kernel', 1)
    # through the class: output_report's three columns of the initial z, then `conf` (which runs the full pass after all)
    np.random.seed(3)
    a = [tl.reassign_colsums(m, initial=True) for m in ('exclude', 'choose', 'average')]
    assert next(iter(tl._report_cache))[1] < 0                  # one pass, asked without a conf column
    conf = tl.reassign_colsums('conf', 0.9, initial=True)
    assert np.allclose(conf, full['conf'], rtol=1e-12, atol=1e-9) and np.array_equal(a[0], full['exclude'].astype(np.int64))
    # `choose`: the picked best hits of the tied rows come from the codes-only kernel (k_choose_init_codes); the generic row pass
    # (report_kernel = 0) must give the same column sums for the same picks — with the device's tie list and with a caller's rows
    picks = np.random.RandomState(9).randint(0, c1).astype(np.int32)
    fast_rows = eng.reassign_rows('choose', 0.9, Z_INITIAL, r1, picks)
    fast_list = eng.reassign_rows('choose', 0.9, Z_INITIAL, None, picks, n=len(r1))
    eng.set_option('report_kernel', 0)
    slow_rows = eng.reassign_rows('choose', 0.9, Z_INITIAL, r1, picks)
    eng.set_option('report_kernel', 1)
    assert np.array_equal(fast_rows, slow_rows) and np.array_equal(fast_list, slow_rows) and int(slow_rows.sum()) == len(r1)


def test_initial_report_pass_with_a_stored_zero_score_takes_the_full_kernel(gpu_device):
    """A stored score of 0 (Q = 0: the entry is in z's pattern with z = 0, and a row of zeros ties all its entries) is what the
    codes-only pass cannot tell from its padding: the library then runs the full pass, same results as the generic one."""
    import scipy.sparse as sp
    from telescope_amd._lib import Z_INITIAL
    from telescope_amd.likelihood import TelescopeLikelihood
    rng = np.random.RandomState(4)
    n, k = 20_000, 300
    lens = rng.randint(1, 12, n)
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.concatenate([np.sort(rng.choice(k, l, replace=False)) for l in lens]).astype(np.int32)
    data = rng.randint(0, 40, indptr[-1]).astype(np.uint16)       # zeros included
    data[indptr[5]:indptr[6]] = 0                                  # a row of zeros only
    tl = TelescopeLikelihood(sp.csr_matrix((data, indices, indptr), shape=(n, k)), Opts(max_iter=2, em_epsilon=0.0), device=0)
    eng = tl._eng
    fast, r2, c2 = eng.report_colsums(Z_INITIAL, -1.0)
    eng.set_option('report_kernel', 0)
    gen, r3, c3 = eng.report_colsums(Z_INITIAL, 0.9)
    assert np.array_equal(fast['exclude'], gen['exclude']) and np.array_equal(r2, r3) and np.array_equal(c2, c3)
    assert np.allclose(fast['average'], gen['average'], rtol=1e-12, atol=1e-9)


def _soak_case(fn, seed):
    """One case of tests/fuzz_reports.py, ONE attempt (round 5 retried up to three times around near-ties of the final z; since round 6
    the report and row passes redo such rows in the reference's order of additions — DESIGN 8.8 — and nothing is retried)."""
    res = fn(seed)
    assert res.startswith('ok') or res.startswith('skipped'), res


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [3, 10, 41, 85, 145, 146, 212, 301, 502])
def test_reassign_column_sums_of_random_matrices_against_the_oracle(gpu_device, seed):
    """A slice of tests/fuzz_reports.py (the soak ran 400 seeds: profiles/r05_fuzz_reports.txt): a random small matrix through em()
    and all twelve reassign column sums — six methods x initial / final z, in random order with random thresholds, so that the report
    cache, the codes-only pass of the initial z, the device's tie list and `choose`'s picks meet in every combination — against the
    oracle, integer columns bit for bit.  Seeds 10, 85 and 145 are the two-score matrices whose final masks sit on near-ties:
    the oracle's final z is formed from the engine's parameters, which is what makes them comparable."""
    import fuzz_reports as fuzz
    _soak_case(fuzz.one, seed)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [1, 22, 57, 140, 263])
def test_public_methods_on_random_matrices_against_the_oracle(gpu_device, seed):
    """A slice of `python tests/fuzz_reports.py 0 300 public` (0 failures of 271 cases): estep / mstep / calculate_lnl with caller-supplied
    parameters, a share of them exactly 0 (entries leave z's pattern as in scipy's CSR arithmetic), against the oracle."""
    import fuzz_reports as fuzz
    _soak_case(fuzz.public, seed)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [10, 3, 77, 118])
def test_row_sharded_random_matrices_against_the_oracle(gpu_device, seed):
    """A slice of `python tests/fuzz_reports.py 0 150 sharded` (139 cases, 0 failures): the random matrices row-sharded over 2 or 3
    in-process ranks (the shipped tsem_em_chunk protocol), ranks without rows included; pi / theta / lnl on every rank, bit-identical
    between ranks, all twelve report columns against the oracle.  Seed 10 is the case that exposed FMA-contracted row sums in the
    streaming report kernel (one ulp in 1 / rowsum broke an exact tie of two z values): the report, mask and z kernels are compiled
    with -ffp-contract=off since (telescope_amd/_lib.py build_library)."""
    import fuzz_reports as fuzz
    _soak_case(fuzz.sharded, seed)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [5, 64, 199])
def test_lookups_on_random_matrices_against_the_oracle(gpu_device, seed):
    """A slice of `python tests/fuzz_reports.py 0 300 lookups` (262 cases, 0 failures): tl.lookup's (z, mask) values for random pairs —
    stored, absent, repeated rows — for five methods x initial / final z against the oracle's matrices."""
    import fuzz_reports as fuzz
    _soak_case(fuzz.lookups, seed)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [8, 91, 230])
def test_group_sums_on_random_matrices_against_the_oracle(gpu_device, seed):
    """A slice of `python tests/fuzz_reports.py 0 300 groups` (263 cases, 0 failures): per-group column sums for random groupings
    (partitions, rows in no or several groups or listed twice, empty groups, small group tiles) against the oracle's masks."""
    import fuzz_reports as fuzz
    _soak_case(fuzz.groups, seed)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [2, 37, 120, 201, 288])
def test_convergence_on_random_matrices_against_the_oracle(gpu_device, seed):
    """A slice of `python tests/fuzz_reports.py 0 300 converge` (271 cases, 0 failures, none stopped apart): em() to convergence under
    both of the reference's tests with epsilons that stop early, late or never — iteration count, pi, theta, lnl against the oracle."""
    import fuzz_reports as fuzz
    _soak_case(fuzz.converge, seed)
