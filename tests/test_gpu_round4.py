"""Round 4, `-m gpu`: the log-likelihood carried by the EM pass (`--use_likelihood`, model.py:783-789; fused kernel MODE 4),
row statistics against the goldens, per-barcode sums at scale.  Everything goes through the C ABI (telescope_amd/_lib.py)."""
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


# ---- the lagged log-likelihood ------------------------------------------------------------------------------------------

@pytest.mark.parametrize('name', [n for n in case_names() if n.endswith('_lnl')])
def test_use_likelihood_goldens_run_on_the_carrying_pass(gpu_device, name):
    """The `--use_likelihood` goldens (captured from the reference): same iteration count, lnl, parameters — with the layout that
    lets the EM pass of iteration t+1 sum the lnl of iteration t, whether it was asked for at construction (`opts.use_likelihood`,
    as telescope_assign.py:434-439 has it) or only at `em(use_likelihood=True)` (the layout is rebuilt then)."""
    from telescope_amd.likelihood import TelescopeLikelihood
    c = load_case(name)
    raw = case_matrix(c)
    for hint in (True, False):
        o = Opts(c)
        if hint:
            o.use_likelihood = True
        tl = TelescopeLikelihood(raw, o, device=0)
        assert tl._eng.layout_info()['lnl_fused'] == (1 if hint else 0)
        tl.em(use_likelihood=True)
        info = tl._eng.layout_info()
        assert info['lnl_fused'] == 1 and info['fused'] == 1, info
        assert tl.n_iter == int(c['n_iter']) and tl.converged == bool(c['converged']), (tl.n_iter, int(c['n_iter']))
        assert abs(tl.lnl - float(c['lnl'])) <= RTOL * abs(float(c['lnl']))
        assert np.allclose(tl.pi, c['pi'], rtol=RTOL, atol=1e-300) and np.allclose(tl.theta, c['theta'], rtol=RTOL, atol=1e-300)
        assert np.allclose(tl.pi_init, c['pi_init'], rtol=RTOL, atol=1e-300)
        # z is the E-step before the LAST M-step (model.py:795): the pass that found the run converged committed nothing
        assert np.array_equal(tl.reassign_colsums('exclude'), np.asarray(c['ra_exclude_0_colsum']).astype(np.int64))


@pytest.mark.parametrize('rows,cols,d,chunks', [(1_000_000, 30_000, 20, (4, 3)), (300_000, 12_000, 9, (1, 1, 5)), (400_000, 38_000, 40, (7,))])
def test_lnl_trace_against_the_c_oracle(gpu_device, rows, cols, d, chunks):
    """BASELINE config 2 (and a short-row / a wide matrix) under `--use_likelihood`, fixed iterations: the log-likelihood of EVERY
    iteration to 1e-9 against oracle/em_fused.c, through tsem_em_chunk itself in chunks of several sizes (the last value of a chunk
    arrives as the next chunk's carry, the last of the run is flushed by the dedicated pass)."""
    from oracle import em_fused as oc
    total = sum(chunks)
    tl = _synthetic_tl(rows, cols, d, 'zipf', uniq=0.05, options=(('use_likelihood', 1),), opts=Opts(max_iter=total, em_epsilon=0.0))
    eng = tl._eng
    assert eng.layout_info()['lnl_fused'] == 1
    got, diffs, done = [], [], 0
    for ci, n in enumerate(chunks):
        d_, l_, stopped = eng.em_chunk(n, 0.0, True, first=(ci == 0), last=(ci == len(chunks) - 1))
        assert not stopped and len(d_) == n
        if ci > 0:
            assert math.isnan(got[-1]) and not math.isnan(eng.lnl_carry)
            got[-1] = eng.lnl_carry
        else:
            assert math.isnan(eng.lnl_carry)
        got.extend(float(v) for v in l_)
        diffs.extend(float(v) for v in d_)
        assert math.isnan(got[-1]) == (ci < len(chunks) - 1)
    ip, ix, rw = eng.export_csr()
    ref = oc.em_fused_arrays(ip, ix, rw, cols, 0, 200000, 0.0, total, use_likelihood=True)
    assert np.allclose(got, ref['lnls'], rtol=RTOL, atol=0), (got, ref['lnls'])
    assert np.allclose(diffs, ref['diffs'], rtol=1e-7, atol=1e-15)
    pi, theta = eng.get_params()
    assert np.allclose(pi, ref['pi'], rtol=RTOL, atol=0) and np.allclose(theta, ref['theta'], rtol=RTOL, atol=0)


@pytest.mark.parametrize('rows,d,eps', [(300_000, 20, 3.0), (150_000, 12, 0.5)])
def test_use_likelihood_converges_in_the_oracles_iteration(gpu_device, rows, d, eps):
    """The |lnl_t - lnl_(t-1)| < epsilon test (model.py:785-789) is evaluated by the update kernel of iteration t+1 before it commits:
    the run stops after the oracle's iteration, far below the cap, with that iteration's parameters."""
    from oracle import em_fused as oc
    tl = _synthetic_tl(rows, 30_000, d, 'zipf', uniq=0.05, opts=Opts(max_iter=400, em_epsilon=eps))
    tl.em(use_likelihood=True)
    assert tl._eng.layout_info()['lnl_fused'] == 1
    ip, ix, rw = tl._eng.export_csr()
    ref = oc.em_fused_arrays(ip, ix, rw, 30_000, 0, 200000, eps, 400, use_likelihood=True)
    assert ref['converged'] and 3 < ref['n_iter'] < 300
    assert tl.converged and tl.n_iter == ref['n_iter'], (tl.n_iter, ref['n_iter'])
    assert abs(tl.lnl - ref['lnl']) <= RTOL * abs(ref['lnl'])
    assert np.allclose(tl.pi, ref['pi'], rtol=RTOL, atol=0) and np.allclose(tl.theta, ref['theta'], rtol=RTOL, atol=0)


@pytest.mark.parametrize('when', [0, 3])
def test_time_out_under_the_lagged_scheme(gpu_device, when):
    """A hand-off time-out of the carrying pass (fused_dbg bit 5) in the first iteration of a run or in the middle: nobody commits,
    the layout is rebuilt for the two-pass kernels, the lnl that was owed comes from the dedicated pass — trace and result as without."""
    from oracle import em_fused as oc
    tl = _synthetic_tl(200_000, 9_000, 14, 'zipf', uniq=0.05, options=(('use_likelihood', 1),), opts=Opts(max_iter=7, em_epsilon=0.0))
    eng = tl._eng
    got = []
    if when:
        _, l_, _ = eng.em_chunk(when, 0.0, True, first=True)
        got.extend(float(v) for v in l_)
    eng.set_option('fused_dbg', 32)
    _, l_, stopped = eng.em_chunk(7 - when, 0.0, True, first=(when == 0), last=True)
    if when:
        got[-1] = eng.lnl_carry
    got.extend(float(v) for v in l_)
    info = eng.layout_info()
    assert info['fallbacks'] == 1 and info['fused'] == 0 and not stopped
    ip, ix, rw = eng.export_csr()
    ref = oc.em_fused_arrays(ip, ix, rw, 9_000, 0, 200000, 0.0, 7, use_likelihood=True)
    assert np.allclose(got, ref['lnls'], rtol=RTOL, atol=0), (got, ref['lnls'])
    pi, _ = eng.get_params()
    assert np.allclose(pi, ref['pi'], rtol=RTOL, atol=0)


def test_second_run_under_the_lagged_scheme(gpu_device):
    """A second em(use_likelihood=True) on the same model: its first lnl is compared with the one the first run ended on (model.py:786)."""
    from oracle.telescope_oracle import OracleModel
    c = load_case('bundled_lnl')
    raw = case_matrix(c)
    from telescope_amd.likelihood import TelescopeLikelihood
    o = Opts(c)
    o.max_iter = 6
    tl = TelescopeLikelihood(raw, o, device=0)
    om = OracleModel(raw, o.pi_prior, o.theta_prior)
    for _ in range(2):
        tl.em(use_likelihood=True)
        om.em(o.em_epsilon, o.max_iter, use_likelihood=True)
        assert tl.n_iter == om.n_iter and tl.converged == om.converged
        assert abs(tl.lnl - om.lnl) <= RTOL * abs(om.lnl) and np.allclose(tl.pi, om.pi, rtol=RTOL, atol=1e-300)


# ---- option `reproducible`: a hand-off time-out keeps the fused kernel ------------------------------------

@pytest.mark.parametrize('bit', [32, 64])
def test_reproducible_mode_redoes_a_timed_out_pass_on_the_fused_kernel(gpu_device, bit):
    """The exact sums exist only in the fused kernel, so a hand-off time-out (fused_dbg bit 5: EM pass, bit 6: lnl pass) must not
    send a `reproducible` handle to the two-pass kernels — round 3 tried, failed inside the layout build and left the context
    half torn down.  Now the pass is redone on the same layout: nobody commits the failed pass, the run ends in the SAME BITS as
    a run without the time-out."""
    import scipy.sparse as sp
    from telescope_amd import _lib
    from telescope_amd.likelihood import TelescopeLikelihood, score_lut
    c = load_case('mid_zipf_20k')
    raw = case_matrix(c)
    runs = []
    for dbg in (0, bit):
        eng = _lib.Engine(0)
        eng.set_option('reproducible', 1)
        if dbg:
            eng.set_option('fused_dbg', dbg)
        eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), raw.shape[1], score_lut(int(raw.data.max())))
        tl = TelescopeLikelihood.from_engine(eng, Opts(c))
        tl._raw = sp.csr_matrix(raw)
        tl.em()
        info = eng.layout_info()
        assert info['fused'] == 1 and info['reproducible'] == 1          # still the fused kernel, still exact
        assert info['fallbacks'] == (1 if dbg else 0)
        runs.append((tl.n_iter, tl.lnl, tl.pi.copy(), tl.theta.copy(), tl.reassign_colsums('conf', 0.9)))
    a, b = runs
    assert a[0] == b[0] == int(c['n_iter'])
    assert a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    assert abs(a[1] - float(c['lnl'])) <= RTOL * abs(float(c['lnl']))


# ---- the constructor's products against the goldens (model.py:635-700) -------------------------------------------------------

@pytest.mark.parametrize('name', case_names(full_only=True))
def test_constructor_products_against_the_goldens(gpu_device, name):
    """What `TelescopeLikelihood.__init__` leaves behind (model.py:640-699), read back from the DEVICE and compared with what the
    reference's constructor held when the golden case was recorded (tools/make_golden.py): max_score, Q.data and the per-row Y
    and weights bit for bit (Q comes from a table built with the reference's numpy expression, w is a maximum, Y a count); the
    sums W_tot, W_amb, pisum0 and the prior weights to summation order (1e-12; the device sums in another order than scipy)."""
    from telescope_amd.likelihood import TelescopeLikelihood
    c = load_case(name)
    raw = case_matrix(c)
    tl = TelescopeLikelihood(raw, Opts(c))
    assert tl.max_score == int(c['max_score'])
    assert np.array_equal(tl.Q.data, c['Q_data']) and np.array_equal(tl.Q.indices, raw.indices)
    y, w = tl._eng.row_info()                                       # tsem_export_rowinfo: the k_rowstats outputs every EM pass reads
    assert y.dtype == np.uint8 and np.array_equal(y, c['Y'])
    assert np.array_equal(tl.Y, c['Y'].reshape(-1, 1)) and tl.Y.shape == (raw.shape[0], 1)
    assert np.array_equal(w, c['weights'])
    assert np.array_equal(np.asarray(tl._weights.todense()).ravel(), c['weights'])
    for got, key in ((tl._total_wt, 'total_wt'), (tl._ambig_wt, 'ambig_wt'), (tl._pi_prior_wt, 'pi_prior_wt'),
                     (tl._theta_prior_wt, 'theta_prior_wt')):
        assert abs(got - float(c[key])) <= 1e-12 * abs(float(c[key])), key
    assert np.allclose(tl._pisum0, c['pisum0'], rtol=1e-12, atol=0)
    assert np.array_equal(tl._pisum0 == 0, c['pisum0'] == 0)        # columns without a unique fragment stay exactly 0


# ---- per-barcode sums at the scale BASELINE config 5 names -------------------------------------------------------------------

def test_per_barcode_sums_at_config5_scale(gpu_device):
    """`scTelescope.output_report`'s per-barcode count matrix (model.py:611-625) on a pooled single-cell style shard: 5M fragments x
    50k loci x ~100 per row, 2000 barcodes (an 800 MB count matrix).  `exclude` / `unique` / `all` bit for bit against fancy
    indexing on the ORACLE's assignment matrix over a 200k-row sub-sample, `conf` / `average` to rounding; at full size the
    barcodes' lines add up to the ungrouped column sums (every fragment has exactly one barcode) — integer methods exactly."""
    import scipy.sparse as sp
    from oracle.telescope_oracle import OracleModel
    from telescope_amd._lib import Z_PREV
    rows, cols, n_groups, sub = 5_000_000, 50_000, 2000, 200_000
    tl = _synthetic_tl(rows, cols, 100, 'zipf', uniq=0.05, opts=Opts(max_iter=3, em_epsilon=0.0))
    tl.em()
    eng = tl._eng
    rng = np.random.RandomState(11)
    bc = rng.randint(0, n_groups, rows).astype(np.int32)
    # (a) the sub-sample: barcodes of the first `sub` rows only
    bc_sub = np.full(rows, -1, np.int32)
    bc_sub[:sub] = bc[:sub]
    ip, ix, rw = eng.export_csr()
    raw_sub = sp.csr_matrix((rw[:ip[sub]], ix[:ip[sub]], ip[:sub + 1]), shape=(sub, cols))
    pp, tp = eng.get_params(Z_PREV)
    om = OracleModel(raw_sub, 0, 200000, max_score=tl.max_score)
    om.z = om.estep(pp, tp)                                # z of the last E-step (model.py:795), rows are independent given pi, theta
    member = sp.csr_matrix((np.ones(sub), (bc[:sub], np.arange(sub))), shape=(n_groups, sub))
    for method in ('exclude', 'unique', 'all', 'conf', 'average'):
        eng.set_groups(bc_sub, n_groups)
        got = eng.reassign_groups(method, 0.9, Z_PREV, None, n_groups)
        want = np.asarray((member @ sp.csr_matrix(om.reassign(method, 0.9)).astype(np.float64)).todense())
        if method in ('conf', 'average'):
            assert np.allclose(got, want, rtol=1e-9, atol=1e-12), method
        else:
            assert np.array_equal(got, want), method
    # (b) full size: one pass per method over all barcodes; the lines add up to the ungrouped sums
    eng.set_groups(bc, n_groups)
    out = np.zeros((n_groups, cols))
    for method in ('exclude', 'unique', 'all', 'average', 'conf'):
        eng.reassign_groups(method, 0.9, Z_PREV, None, n_groups, out=out)
        total = out.sum(0)
        ref = tl.reassign_colsums(method, 0.9)
        if method in ('conf', 'average'):
            assert np.allclose(total, ref, rtol=1e-9, atol=1e-9), method
        else:
            assert np.array_equal(total.astype(np.int64), ref), method
    # (c) the same in tiles of 100 MB (8 passes over the matrix): the caller's matrix need not fit a device buffer
    eng.set_option('group_tile_bytes', 100 << 20)
    tiled = eng.reassign_groups('exclude', 0.9, Z_PREV, None, n_groups)
    eng.reassign_groups('exclude', 0.9, Z_PREV, None, n_groups, out=out)
    assert np.array_equal(tiled, out)
    with pytest.raises(Exception):
        eng.set_groups(np.full(rows, n_groups, np.int32), n_groups)      # out of range: refused (checked on the device)


# ---- the SPLIT layout: K beyond 8 x 7680 on the fused kernel ---------------------------------------------

def _engine_with(raw, options):
    from telescope_amd import _lib
    from telescope_amd.likelihood import score_lut
    eng = _lib.Engine(0)
    for k, v in options:
        eng.set_option(k, v)
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), raw.shape[1], score_lut(int(raw.data.max())))
    return eng


@pytest.mark.parametrize('name', ['bundled', 'bundled_lnl', 'tiny_ties', 'tiny_twins', 'tiny_priors', 'tiny_empty_row', 'mid_zipf_20k', 'mid_uniform_200k'])
@pytest.mark.parametrize('fmt,parts', [(0, 5), (1, 8)])
def test_split_layout_forced_on_the_goldens(gpu_device, name, fmt, parts):
    """Option `split` = 1 forces the two-pass-per-iteration form of the fused kernel (row factors through HBM, then
    acc[j] += Q_ij s_i, pi_j theta_j applied by the column reduce; the log-likelihood over two halves of every part's columns)
    on matrices that do not need it: the reference's iteration count, parameters, lnl and integer reassign outputs."""
    import scipy.sparse as sp
    from telescope_amd.likelihood import TelescopeLikelihood
    c = load_case(name)
    raw = case_matrix(c)
    eng = _engine_with(raw, (('split', 1), ('parts', parts), ('value_format', fmt)))
    tl = TelescopeLikelihood.from_engine(eng, Opts(c))
    tl._raw = sp.csr_matrix(raw)
    tl.em(use_likelihood=bool(c['use_likelihood']))
    info = eng.layout_info()
    if raw.shape[0] - int((np.diff(raw.indptr) <= 1).sum()) > 0:
        assert info['split'] == 1 and info['fused'] == 1 and info['P'] == parts
    assert tl.n_iter == int(c['n_iter']) and tl.converged == bool(c['converged'])
    assert abs(tl.lnl - float(c['lnl'])) <= RTOL * abs(float(c['lnl']))
    assert np.allclose(tl.pi, c['pi'], rtol=RTOL, atol=0) and np.allclose(tl.theta, c['theta'], rtol=RTOL, atol=0)
    assert np.allclose(tl.pi_init, c['pi_init'], rtol=RTOL, atol=0)
    if 'ra_exclude_0_colsum' in c:
        assert np.array_equal(tl.reassign_colsums('exclude'), c['ra_exclude_0_colsum'])
        assert np.allclose(tl.reassign_colsums('conf', 0.9), c['ra_conf_0_colsum'], rtol=RTOL, atol=1e-12)


@pytest.mark.parametrize('rows,cols,d,fmt', [(200_000, 100_000, 100, 0), (200_000, 100_000, 100, 1), (300_000, 70_000, 30, 0),
                                             (150_000, 122_000, 60, 0)])
def test_large_K_runs_on_the_fused_kernel_and_matches_the_c_oracle(gpu_device, rows, cols, d, fmt):
    """K > 61 440 (more than 8 parts of 7680 columns): the split layout keeps the fused kernel — parts of up to 15 360 columns, a
    row-sum pass and a scatter pass per iteration — instead of the two-pass kernels.  pi, theta, pi_init and lnl to 1e-9 against
    oracle/em_fused.c over the SAME matrix, the per-locus `exclude` counts bit for bit; use_likelihood (a log-likelihood per
    iteration: three more launches each) ends in the same iteration with the same trace."""
    from oracle import em_fused as oc
    from telescope_amd._lib import Z_PREV
    iters = 6
    tl = _synthetic_tl(rows, cols, d, 'zipf', uniq=0.05, options=(('value_format', fmt),), opts=Opts(max_iter=iters, em_epsilon=0.0))
    info = tl._eng.layout_info()
    assert info['split'] == 1 and info['fused'] == 1 and 5 <= info['P'] <= 8 and info['Kp'] > 7680
    tl.em()
    ip, ix, rw = tl._eng.export_csr()
    ref = oc.em_fused_arrays(ip, ix, rw, cols, 0, 200000, 0.0, iters)
    assert ref['n_iter'] == tl.n_iter == iters
    assert abs(tl.lnl - ref['lnl']) <= RTOL * abs(ref['lnl'])
    assert np.allclose(tl.pi, ref['pi'], rtol=RTOL, atol=0) and np.allclose(tl.theta, ref['theta'], rtol=RTOL, atol=0)
    assert np.allclose(tl.pi_init, ref['pi_init'], rtol=RTOL, atol=0)
    pp, tp = tl._eng.get_params(Z_PREV)
    conf, excl, avg = oc.report_sums(ip, ix, rw, cols, pp, tp, 0.9, False, max_score=tl.max_score)
    assert np.array_equal(tl.reassign_colsums('exclude'), excl)
    assert np.allclose(tl.reassign_colsums('conf', 0.9), conf, rtol=RTOL, atol=1e-9)
    assert tl._eng.layout_info()['fallbacks'] == 0
    tl2 = _synthetic_tl(rows, cols, d, 'zipf', uniq=0.05, options=(('value_format', fmt),), opts=Opts(max_iter=4, em_epsilon=0.0))
    tl2.em(use_likelihood=True)
    ref2 = oc.em_fused_arrays(ip, ix, rw, cols, 0, 200000, 0.0, 4, use_likelihood=True)
    assert tl2.n_iter == ref2['n_iter'] and abs(tl2.lnl - ref2['lnl']) <= RTOL * abs(ref2['lnl'])


def test_split_layout_time_out_falls_back_to_the_two_pass_kernels(gpu_device):
    """A hand-off time-out of the row-sum pass (fused_dbg bit 5) or of the log-likelihood launches (bit 6) on the split layout:
    nobody commits, the handle rebuilds for the two-pass kernels (which take any K up to 64 x 7680) and finishes with the same numbers."""
    from telescope_amd.likelihood import TelescopeLikelihood
    ref = _synthetic_tl(60_000, 70_000, 40, 'zipf', uniq=0.05, opts=Opts(max_iter=4, em_epsilon=0.0))
    ref.em()
    for bit in (32, 64):
        tl = _synthetic_tl(60_000, 70_000, 40, 'zipf', uniq=0.05, options=(('fused_dbg', bit),), opts=Opts(max_iter=4, em_epsilon=0.0))
        assert tl._eng.layout_info()['split'] == 1
        tl.em()
        info = tl._eng.layout_info()
        assert info['fused'] == 0 and info['split'] == 0 and info['fallbacks'] == 1
        assert tl.n_iter == ref.n_iter and abs(tl.lnl - ref.lnl) <= 1e-10 * abs(ref.lnl)
        assert np.allclose(tl.pi, ref.pi, rtol=1e-10, atol=0)


def test_any_number_of_loci(gpu_device):
    """The reference takes any K (model.py:643).  K = 600 000 is beyond the fused kernel (<= 122 880) and the two-pass kernels
    (<= 64 parts of 7680): the EM pass and the log-likelihood run as plain CSR row passes (round 3 refused the matrix) — against the
    oracle: parameters, lnl, integer and float report columns, `use_likelihood`."""
    import scipy.sparse as sp
    from oracle.telescope_oracle import OracleModel
    from telescope_amd.likelihood import TelescopeLikelihood
    rng = np.random.RandomState(11)
    n, k = 40_000, 600_000
    hot = rng.randint(0, k, 3000)                                           # a few thousand loci carry most of the entries
    rows, cols = [], []
    for i in range(n):
        l = 1 if rng.rand() < 0.1 else rng.randint(2, 40)
        c = np.unique(np.where(rng.rand(l) < 0.8, hot[rng.randint(0, 3000, l)], rng.randint(0, k, l)))
        rows.append(np.full(len(c), i)); cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    data = rng.randint(139, 213, len(cols)).astype(np.uint16)
    raw = sp.csr_matrix((data, (rows, cols)), shape=(n, k))
    raw.sort_indices()
    for use_lnl in (False, True):
        o = Opts(max_iter=5, em_epsilon=0.0)
        tl = TelescopeLikelihood(raw, o)
        info = tl._eng.layout_info()
        assert info['row_pass_em'] == 1 and info['fused'] == 0 and info['split'] == 0
        tl.em(use_likelihood=use_lnl)
        om = OracleModel(raw, o.pi_prior, o.theta_prior)
        om.em(0.0, 5, use_lnl)
        assert tl.n_iter == om.n_iter == 5
        assert abs(tl.lnl - om.lnl) <= RTOL * abs(om.lnl)
        assert np.allclose(tl.pi, om.pi, rtol=RTOL, atol=1e-300) and np.allclose(tl.theta, om.theta, rtol=RTOL, atol=1e-300)
        assert np.allclose(tl.pi_init, om.pi_init, rtol=RTOL, atol=1e-300)
    assert np.array_equal(tl.reassign_colsums('exclude'), np.asarray(om.reassign('exclude').sum(0)).ravel())
    assert np.allclose(tl.reassign_colsums('conf', 0.9), np.asarray(om.reassign('conf', 0.9).sum(0)).ravel(), rtol=RTOL, atol=1e-12)
    assert np.array_equal(tl.reassign_colsums('unique'), np.asarray(om.reassign('unique').sum(0)).ravel())


# ---- z / assignment look-ups for update_sam without the N x K matrices ----------------------------------------

@pytest.mark.parametrize('name', case_names(full_only=True))
def test_lookups_for_update_sam_match_the_reference(gpu_device, name):
    """`Telescope.update_sam` reads `tl.z[ridx, fidx]` and `mat[ridx, fidx]` per alignment (model.py:483,508-511).  `tl.lookup`
    answers arrays of such pairs from two device passes over the distinct rows asked for (tsem_rows_lookup) — here EVERY stored
    pair of the golden case plus pairs outside the pattern, all six methods, final and initial z: the posterior against the
    reference's z, the assignment values against the reference's assignment matrices (integer modes exactly; `choose` with the
    reference's RNG stream), and phred(prob) of helpers.py:14-37."""
    import scipy.sparse as sp
    from telescope_amd.likelihood import TelescopeLikelihood
    c = load_case(name)
    raw = case_matrix(c)
    tl = TelescopeLikelihood(raw, Opts(c))
    tl.em(use_likelihood=bool(c['use_likelihood']))
    n, k = raw.shape
    rows, cols = raw.nonzero()
    rng = np.random.RandomState(3)
    xr, xc = rng.randint(0, n, 200), rng.randint(0, k, 200)                      # mostly pairs that are NOT stored
    ridx, fidx = np.concatenate([rows, xr]), np.concatenate([cols, xc])
    perm = rng.permutation(len(ridx))                                            # any order, repeated rows
    ridx, fidx = ridx[perm], fidx[perm]
    zref = sp.csr_matrix((c['z_data'], c['z_indices'], c['z_indptr']), shape=raw.shape)
    want_z = np.asarray(zref[ridx, fidx]).ravel()
    for initial in (False, True):
        for meth in ('exclude', 'choose', 'average', 'conf', 'unique', 'all'):
            np.random.seed(int(c['seed']))
            prob, val = tl.lookup(ridx, fidx, meth, 0.9, initial)
            tag = 'ra_%s_%d_' % (meth, int(initial))
            ref = sp.csr_matrix((c[tag + 'data'], c[tag + 'indices'], c[tag + 'indptr']), shape=raw.shape)
            want = np.asarray(ref[ridx, fidx]).ravel()
            assert np.allclose(prob, want_z, rtol=RTOL, atol=1e-300), (meth, initial)
            if meth in ('average', 'conf'):
                assert np.allclose(val, want, rtol=RTOL, atol=1e-12), (meth, initial)
            else:
                assert np.array_equal(val, want), (meth, initial)
            assert str(val.dtype) == str(c[tag + 'dtype'])
    # phred(prob) (helpers.py:14-37: -10 log10(1 - P)) — compared before the rounding to an integer, where the posterior leaves room for it:
    # for P within 1e-6 of 1 the score hangs on the last bits of P in the reference itself
    sel = want_z < 1.0 - 1e-6
    assert np.allclose(-10 * np.log10(1 - prob[sel]), -10 * np.log10(1 - want_z[sel]), rtol=0, atol=1e-6)
    assert np.array_equal(prob >= 1.0 - 1e-9, want_z >= 1.0 - 1e-9)              # phred 255 territory (P == 1.0 itself is a last-bit matter)
    # the object `reassign` returned and the look-up agree on the picks `choose` drew
    np.random.seed(7)
    a = tl.reassign('choose', 0.9)
    _, val = tl.lookup(ridx, fidx, 'choose', 0.9, assignment=a)
    m = a.tocsr()
    assert np.array_equal(val, np.asarray(m[ridx, fidx]).ravel())


def test_lookups_at_scale(gpu_device):
    """2M fragments x 30k loci: look-ups for 300 000 alignments spread over the matrix against the full export (`tl.z`, the full
    assignment matrix), which this call exists to avoid."""
    tl = _synthetic_tl(2_000_000, 30_000, 20, 'zipf', uniq=0.05, opts=Opts(max_iter=5, em_epsilon=0.0))
    tl.em()
    r = tl._need_raw()
    rng = np.random.RandomState(5)
    e = np.sort(rng.choice(r.nnz, 300_000, replace=False))
    ridx = np.searchsorted(r.indptr, e, side='right') - 1
    fidx = r.indices[e]
    prob, val = tl.lookup(ridx, fidx, 'exclude', 0.9)
    z = tl.z
    assert np.allclose(prob, np.where(z.data[e] < 0, 0, z.data[e]) if z.nnz == r.nnz else np.asarray(z[ridx, fidx]).ravel(), rtol=1e-12, atol=0)
    m = tl.reassign('exclude', 0.9).tocsr()
    assert np.array_equal(val, np.asarray(m[ridx, fidx]).ravel())
    assert val.sum() > 0.5 * len(e) / 20                                         # (a fair share of the sampled entries are best hits)
