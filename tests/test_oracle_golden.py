"""The oracle pinned against the reference: its own known-answer tests, the
README log-likelihood, the bundled golden report, and golden vectors captured
from the imported reference (tools/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLD, Opts, case_matrix, case_names, load_case
from oracle import telescope_oracle as orc


def dense(m):
    return np.asarray(sp.csr_matrix(m).todense())


# --- the reference's own known-answer tests (telescope/tests/test_sparse_plus.py:24-55) ---
M1 = [[1, 0, 2], [0, 0, 3], [4, 5, 6]]


def test_norm_whole_matrix():
    want = np.array(M1) * (1. / 21)
    assert np.array_equal(dense(orc.norm(sp.csr_matrix(M1))), want)


def test_norm_rows():
    want = np.array([[1 * (1. / 3), 0, 2 * (1. / 3)], [0, 0, 1.], [4 * (1. / 15), 5 * (1. / 15), 6 * (1. / 15)]])
    assert np.allclose(dense(orc.norm(sp.csr_matrix(M1), 1)), want, rtol=0, atol=1e-16)


def test_norm_rows_with_zero_row():
    m = sp.csr_matrix([[1, 0, 2], [0, 0, 0], [4, 5, 6]])
    got = dense(orc.norm(m, 1))
    assert np.array_equal(got[1], [0, 0, 0])
    assert np.allclose(got[0], [1. / 3, 0, 2. / 3]) and np.allclose(got[2], [4. / 15, 5. / 15, 6. / 15])


def test_binmax_docstring_example():
    # sparse_plus.py:107-115
    m = sp.csr_matrix([[6, 0, 2], [0, 0, 3], [4, 5, 6]])
    assert np.array_equal(dense(orc.binmax_rows(m)), [[1, 0, 0], [0, 0, 1], [0, 0, 1]])


def test_scale_docstring_example():
    m = sp.csr_matrix([[10, 0, 20], [0, 0, 30], [40, 50, 60]])
    assert np.allclose(dense(orc.scale(m)), np.array([[10, 0, 20], [0, 0, 30], [40, 50, 60]]) / 60.)


# --- README.md:70-71 and telescope/data/telescope_report.tsv ---
def test_bundled_readme_loglikelihood():
    c = load_case('bundled')
    om = orc.OracleModel(case_matrix(c))
    msgs = []
    om.em(1e-7, 100, log=msgs.append)
    assert om.n_iter == 16 and om.converged
    assert msgs[-2] == 'EM converged after 16 iterations.'
    assert msgs[-1] == 'Final log-likelihood: 95252.596293.'


def test_bundled_golden_report_columns():
    """EM-derived columns of the reference's bundled report (v1.0.2 layout)."""
    c = load_case('bundled')
    raw = case_matrix(c)
    names = list(np.load(os.path.join(GOLD, 'bundled_raw_scores.npz'))['feat_names'])
    om = orc.OracleModel(raw)
    om.em(1e-7, 100)
    np.random.seed(int(c['seed']))
    cols = om.report_columns('exclude', 0.9)
    rows = {}
    with open(os.path.join(GOLD, 'reference_telescope_report.tsv')) as fh:
        lines = fh.read().splitlines()
    assert lines[0].startswith('## RunInfo')
    hdr = lines[1].split('\t')
    for ln in lines[2:]:
        f = ln.split('\t')
        rows[f[0]] = dict(zip(hdr, f))
    assert len(rows) == 59
    for j, name in enumerate(names):
        r = rows[name]
        assert int(r['final_count']) == cols['final_count'][j]
        assert abs(float(r['final_conf']) - cols['final_conf'][j]) <= 0.005
        assert abs(float(r['final_prop']) - cols['final_prop'][j]) <= 5.1e-3 * max(cols['final_prop'][j], 1e-300) + 1e-12
        assert int(r['init_aligned']) == cols['init_aligned'][j]
        assert int(r['unique_count']) == cols['unique_count'][j]
        assert int(r['init_best']) == cols['init_best'][j]
        assert int(r['init_best_random']) == cols['init_best_random'][j]
        assert abs(float(r['init_best_avg']) - cols['init_best_avg'][j]) <= 0.005
        assert abs(float(r['init_prop']) - cols['init_prop'][j]) <= 5.1e-3 * cols['init_prop'][j] + 1e-12


# --- golden vectors captured from the imported reference ---
@pytest.mark.parametrize('name', [pytest.param(n, marks=pytest.mark.gpu) if n.startswith('mid_uniform') else n
                                  for n in case_names()])
def test_oracle_equals_reference_vectors(name):
    # (mid_uniform_200k takes the oracle ~40 s: it runs with the -m gpu suite, on the GPU box's host cores)
    c = load_case(name)
    raw = case_matrix(c)
    o = Opts(c)
    om = orc.OracleModel(raw, o.pi_prior, o.theta_prior)
    assert om.max_score == int(c['max_score'])
    assert om.total_wt == float(c['total_wt']) and om.ambig_wt == float(c['ambig_wt'])
    assert np.array_equal(np.asarray(om.pisum0).ravel(), c['pisum0'])
    msgs = []
    trace = om.em(o.em_epsilon, o.max_iter, bool(c['use_likelihood']), log=msgs.append)
    assert list(c['log_lines']) == msgs
    assert om.n_iter == int(c['n_iter']) and om.converged == bool(c['converged'])
    assert om.lnl == float(c['lnl'])
    assert np.array_equal(om.pi, c['pi']) and np.array_equal(om.theta, c['theta'])
    assert np.array_equal(om.pi_init, c['pi_init'])
    assert np.array_equal([t[0] for t in trace], c['diffs'])
    if 'z_data' in c:
        z = sp.csr_matrix(om.z)
        assert np.array_equal(z.data, c['z_data']) and np.array_equal(z.indices, c['z_indices'])
    for initial in (0, 1):
        for meth in orc.REASSIGN_METHODS:
            np.random.seed(int(c['seed']))
            r = om.reassign(meth, 0.9, bool(initial))
            tag = 'ra_%s_%d_' % (meth, initial)
            assert np.array_equal(np.asarray(r.sum(0)).ravel(), c[tag + 'colsum']), tag
            assert str(r.dtype) == str(c[tag + 'dtype'])


def test_reassign_bad_method():
    c = load_case('tiny_ties')
    om = orc.OracleModel(case_matrix(c))
    om.em(1e-7, 3)
    with pytest.raises(ValueError):
        om.reassign('best')


@pytest.mark.parametrize('name', [n for n in case_names()])
def test_c_restatement_matches_reference_goldens(name):
    """oracle/em_fused.c (plain C, row-parallel, fused E+M) against the vectors captured from the
    reference: same iteration count, lnl / pi / theta to rounding."""
    from oracle.em_fused import em_fused
    c = load_case(name)
    ul = bool(c['use_likelihood'])
    r = em_fused(case_matrix(c), float(c['pi_prior']), float(c['theta_prior']), float(c['em_epsilon']), int(c['max_iter']),
                 use_likelihood=ul)
    assert r['n_iter'] == int(c['n_iter']) and r['converged'] == bool(c['converged'])
    assert abs(r['lnl'] - float(c['lnl'])) <= 1e-10 * abs(float(c['lnl']))
    assert np.allclose(r['pi'], c['pi'], rtol=1e-10, atol=0) and np.allclose(r['theta'], c['theta'], rtol=1e-10, atol=0)
    assert np.allclose(r['pi_init'], c['pi_init'], rtol=1e-10, atol=0)
    r1 = em_fused(case_matrix(c), float(c['pi_prior']), float(c['theta_prior']), float(c['em_epsilon']), int(c['max_iter']), nthreads=1,
                  use_likelihood=ul)
    assert np.allclose(r1['pi'], r['pi'], rtol=1e-12, atol=0) and r1['n_iter'] == r['n_iter']


@pytest.mark.parametrize('name', [n for n in case_names(full_only=True)])
def test_c_report_sums_match_reference_goldens(name):
    """oracle_report_sums (em_fused.c: the conf / exclude / average column sums of one z) against the reference's own
    `reassign(...).sum(0)` vectors, for the final z (posteriors of the parameters BEFORE the last M-step, model.py:795) and for
    the initial one (Q.norm(1), model.py:837): integer counts bit for bit, float sums to summation order."""
    from oracle.em_fused import report_sums
    c = load_case(name)
    raw = case_matrix(c)
    o = Opts(c)
    om = orc.OracleModel(raw, o.pi_prior, o.theta_prior)
    if int(c['n_iter']) > 1:
        om.em(0.0, int(c['n_iter']) - 1, bool(c['use_likelihood']))       # the parameters the last E-step saw
    for initial in (0, 1):
        conf, excl, avg = report_sums(raw.indptr, raw.indices, raw.data, raw.shape[1], om.pi, om.theta, 0.9, bool(initial),
                                      max_score=int(c['max_score']))
        assert np.array_equal(excl, c['ra_exclude_%d_colsum' % initial])
        assert np.allclose(conf, c['ra_conf_%d_colsum' % initial], rtol=1e-10, atol=1e-13)
        assert np.allclose(avg, c['ra_average_%d_colsum' % initial], rtol=1e-12, atol=0)
    if 'ra_conf03_0_colsum' in c:                                          # a threshold below 0.5: several confident entries per row
        conf, _, _ = report_sums(raw.indptr, raw.indices, raw.data, raw.shape[1], om.pi, om.theta, 0.3, False,
                                 max_score=int(c['max_score']))
        assert np.allclose(conf, c['ra_conf03_0_colsum'], rtol=1e-10, atol=1e-13)
