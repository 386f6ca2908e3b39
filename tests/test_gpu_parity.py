"""Parity tests proper: the HIP path (through the C ABI) against golden vectors
captured from the reference, against the oracle on seeded inputs, and — at
BASELINE.json's sizes — through size-independent properties.

Tolerances.  north_star: final log-likelihood and per-locus counts within 1e-4
relative, integer reassign outputs bit-exact.  Everything is fp64 on the
device, so the float checks below use RTOL = 1e-9 (five orders tighter than
the bar); integer outputs are compared with array_equal."""
import logging

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import Opts, case_matrix, case_names, load_case

pytestmark = pytest.mark.gpu
RTOL = 1e-9
INT_METHODS = ('exclude', 'choose', 'unique', 'all')
ALL_METHODS = ('exclude', 'choose', 'average', 'conf', 'unique', 'all')


def make_tl(raw, opts, **kw):
    from telescope_amd.likelihood import TelescopeLikelihood
    return TelescopeLikelihood(raw, opts, **kw)


def run_case(name, em_kernel=None, block_rows=None):
    from telescope_amd import _lib
    from telescope_amd.likelihood import TelescopeLikelihood
    c = load_case(name)
    raw = case_matrix(c)
    o = Opts(c)
    if em_kernel is None and block_rows is None:
        tl = TelescopeLikelihood(raw, o)
    else:
        eng = _lib.Engine(0)
        if em_kernel is not None:
            eng.set_option('em_kernel', em_kernel)
        if block_rows is not None:
            eng.set_option('block_rows', block_rows)
        lut_max = int(raw.data.max())
        from telescope_amd.likelihood import score_lut
        eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), raw.shape[1], score_lut(lut_max))
        tl = TelescopeLikelihood.from_engine(eng, o)
        tl._raw = sp.csr_matrix(raw)
    msgs = []

    class H(logging.Handler):
        def emit(self, rec):
            msgs.append(rec.getMessage())
    lg = logging.getLogger(); h = H(); lg.addHandler(h); old = lg.level; lg.setLevel(logging.INFO)
    try:
        tl.em(use_likelihood=bool(c['use_likelihood']), loglev=logging.INFO)
    finally:
        lg.removeHandler(h); lg.setLevel(old)
    return c, raw, tl, msgs


@pytest.mark.parametrize('name', case_names())
def test_em_matches_reference(gpu_device, name):
    c, raw, tl, msgs = run_case(name)
    assert tl.n_iter == int(c['n_iter'])
    assert tl.converged == bool(c['converged'])
    assert abs(tl.lnl - float(c['lnl'])) <= RTOL * abs(float(c['lnl']))
    assert np.allclose(tl.pi, c['pi'], rtol=RTOL, atol=0)
    assert np.allclose(tl.theta, c['theta'], rtol=RTOL, atol=0)
    assert np.allclose(tl.pi_init, c['pi_init'], rtol=RTOL, atol=0)
    assert np.allclose(tl.theta_init, c['theta_init'], rtol=RTOL, atol=0)
    # log lines (model.py:767-768,804-805): same count, same text up to the printed precision
    ref = list(c['log_lines'])
    assert len(msgs) == len(ref)
    assert msgs[-2] == ref[-2]                       # 'EM converged|terminated after N iterations.'
    assert msgs[-1] == ref[-1]                       # 'Final log-likelihood: %f.'
    import re
    num = re.compile(r'[-+]?\d+\.?\d*(?:e[-+]?\d+)?')
    for a, b in zip(msgs[:-2], ref[:-2]):            # 'Iteration N, [lnl= x,] diff=y'
        va, vb = [float(x) for x in num.findall(a)], [float(x) for x in num.findall(b)]
        assert len(va) == len(vb) and va[0] == vb[0]
        # diffs below ~1e-10 are summation-order noise of |pi_hat - pi|_1 in BOTH implementations
        assert np.allclose(va[1:], vb[1:], rtol=2e-4, atol=1e-11), (a, b)


@pytest.mark.parametrize('name', case_names())
def test_reassign_colsums_match_reference(gpu_device, name):
    c, raw, tl, _ = run_case(name)
    for initial in (False, True):
        for meth in ALL_METHODS:
            np.random.seed(int(c['seed']))
            got = tl.reassign_colsums(meth, 0.9, initial)
            ref = c['ra_%s_%d_colsum' % (meth, int(initial))]
            if meth in INT_METHODS:
                assert np.array_equal(got, ref), (meth, initial)
            else:
                assert np.allclose(got, ref, rtol=RTOL, atol=1e-12), (meth, initial)
    got = tl.reassign_colsums('conf', 0.3)
    assert np.allclose(got, c['ra_conf03_0_colsum'], rtol=RTOL, atol=1e-12)


def test_report_columns_in_output_report_order(gpu_device):
    """model.py:432-457: the RNG is consumed by init_best_random BEFORE the final mode."""
    c, raw, tl, _ = run_case('bundled')
    np.random.seed(int(c['seed']))
    got = dict(
        final_conf=tl.reassign_colsums('conf', 0.9),
        init_aligned=tl.reassign_colsums('all', initial=True),
        unique_count=tl.reassign_colsums('unique'),
        init_best=tl.reassign_colsums('exclude', initial=True),
        init_best_random=tl.reassign_colsums('choose', initial=True),
        init_best_avg=tl.reassign_colsums('average', initial=True),
        final_count=tl.reassign_colsums('exclude', 0.9))
    for k, v in got.items():
        assert np.allclose(v, c['report_' + k], rtol=RTOL, atol=1e-12), k
    assert got['final_count'][np.argmax(c['pi'])] == 1000 and got['init_best'][np.argmax(c['pi'])] == 326


@pytest.mark.parametrize('name', case_names(full_only=True))
def test_z_and_masks_match_reference(gpu_device, name):
    c, raw, tl, _ = run_case(name)
    z = sp.csr_matrix(tl.z)
    assert np.array_equal(z.indptr, c['z_indptr']) and np.array_equal(z.indices, c['z_indices'])
    assert np.allclose(z.data, c['z_data'], rtol=RTOL, atol=1e-300)
    for initial in (False, True):
        for meth in ALL_METHODS:
            np.random.seed(int(c['seed']))
            a = tl.reassign(meth, 0.9, initial)
            after = np.random.get_state()[2]                 # `choose` has drawn by now, not later
            colsum = a.sum(0).A1                             # what output_report asks for (model.py:435-457)
            m = a.tocsr()
            assert np.random.get_state()[2] == after
            assert a.shape == raw.shape and sp.issparse(m)
            assert np.allclose(colsum, np.asarray(m.sum(0)).ravel(), rtol=1e-12, atol=1e-12)
            assert colsum.dtype == (np.int64 if meth in INT_METHODS else np.float64)
            assert a[0, int(raw.indices[0])] == m[0, int(raw.indices[0])] and a.nnz == m.nnz
            tag = 'ra_%s_%d_' % (meth, int(initial))
            ref = sp.csr_matrix((c[tag + 'data'], c[tag + 'indices'], c[tag + 'indptr']), shape=raw.shape)
            assert str(m.dtype) == str(c[tag + 'dtype'])
            if meth in INT_METHODS:
                assert (m != ref).nnz == 0, (meth, initial)
            else:
                assert abs(m - ref).max() <= 1e-12 if ref.nnz or m.nnz else True


@pytest.mark.parametrize('name', ['bundled', 'tiny_ties', 'tiny_wide_range', 'tiny_twins', 'tiny_priors'])
def test_public_estep_mstep_lnl(gpu_device, name):
    """estep/mstep/calculate_lnl keep their signatures (model.py:702-760); params with
    exact zeros drop entries from z's pattern like scipy's CSR addition."""
    c = load_case(name)
    raw = case_matrix(c)
    tl = make_tl(raw, Opts(c))
    pi, theta = c['x_pi'], c['x_theta']
    z = sp.csr_matrix(tl.estep(pi, theta))
    assert np.array_equal(z.indptr, c['x_z_indptr']) and np.array_equal(z.indices, c['x_z_indices'])
    assert np.allclose(z.data, c['x_z_data'], rtol=RTOL, atol=1e-300)
    p2, t2 = tl.mstep(z)
    assert np.allclose(p2, c['x_pi2'], rtol=RTOL, atol=0) and np.allclose(t2, c['x_theta2'], rtol=RTOL, atol=0)
    l2 = tl.calculate_lnl(z, p2, t2)
    assert abs(l2 - float(c['x_lnl2'])) <= RTOL * abs(float(c['x_lnl2']))


def test_engine_options_keyword(gpu_device):
    c = load_case('bundled')
    from telescope_amd.likelihood import TelescopeLikelihood
    for fmt, nbytes in ((1, 8), (2, 2)):
        tl = TelescopeLikelihood(case_matrix(c), Opts(c), engine_options={'value_format': fmt})
        tl.em()
        assert tl._eng.layout_info()['value_bytes'] == nbytes
        assert abs(tl.lnl - float(c['lnl'])) <= RTOL * abs(float(c['lnl']))


def test_error_behaviour(gpu_device):
    c = load_case('tiny_ties')
    tl = make_tl(case_matrix(c), Opts(c))
    with pytest.raises(ValueError, match='Argument "method" should be one of'):
        tl.reassign('best')
    with pytest.raises(ValueError):
        tl.reassign('exclude')        # before em(): no posteriors
    with pytest.raises(ValueError):
        make_tl(sp.csr_matrix(np.array([[1.5, 2.0]])), Opts())


def test_csr_matrix_plus_primitives(gpu_device):
    """norm(1) / binmax(1) known answers of telescope/tests/test_sparse_plus.py:24-55."""
    from telescope_amd import _lib
    m = sp.csr_matrix(np.array([[1, 0, 2], [0, 0, 0], [4, 5, 6]], dtype=np.float64))
    out = _lib.csr_norm_rows(m.indptr, m.data)
    assert np.allclose(out, [1 / 3, 2 / 3, 4 / 15, 5 / 15, 6 / 15], rtol=1e-15)
    m = sp.csr_matrix(np.array([[6, 0, 2], [0, 0, 3], [4, 5, 6]], dtype=np.float64))
    assert np.array_equal(_lib.csr_binmax_rows(m.indptr, m.data, 3), [1, 0, 1, 0, 0, 1])


@pytest.mark.parametrize('dist,uniq', [('uniform', 0.0), ('zipf', 0.1), ('family', 0.05)])
def test_device_generator_is_bit_exact(gpu_device, dist, uniq):
    from telescope_amd import _lib, synthetic
    n, k, d = 20000, 3000, 20
    eng = _lib.Engine(0)
    eng.generate(0, n, k, synthetic.poisson_cdf_u32(d), 99, synthetic.DIST_CODE[dist], uniq)
    ip, ix, rw = eng.export_csr()
    ip2, ix2, rw2 = synthetic.generate(n, k, d, seed=99, dist=dist, uniq_frac=uniq)
    assert np.array_equal(ip, ip2) and np.array_equal(ix, ix2) and np.array_equal(rw, rw2)
    eng.generate(5000, 7000, k, synthetic.poisson_cdf_u32(d), 99, synthetic.DIST_CODE[dist], uniq)
    _, ix3, _ = eng.export_csr()
    assert np.array_equal(ix3, ix2[ip2[5000]:ip2[7000]])


def _synthetic_tl(rows, cols, d, dist, seed=42, uniq=0.0, r0=0, r1=None, options=(), opts=None):
    from telescope_amd import _lib, synthetic
    from telescope_amd.likelihood import TelescopeLikelihood
    eng = _lib.Engine(0)
    for k, v in options:
        eng.set_option(k, v)
    eng.set_option('row_offset', r0)
    eng.generate(r0, rows if r1 is None else r1, cols, synthetic.poisson_cdf_u32(d), seed,
                 synthetic.DIST_CODE[dist], uniq)
    return TelescopeLikelihood.from_engine(eng, opts or Opts(max_iter=5, em_epsilon=0.0))


def test_against_oracle_midsize_seeded(gpu_device):
    """200k x 30k (4 column parts), 10 % unique rows, against the oracle."""
    from oracle.telescope_oracle import OracleModel
    tl = _synthetic_tl(200000, 30000, 20, 'zipf', seed=5, uniq=0.1)
    assert tl._eng.layout_info()['P'] == 4
    ip, ix, rw = tl._eng.export_csr()
    om = OracleModel(sp.csr_matrix((rw, ix, ip), shape=(200000, 30000)))
    trace = om.em(0.0, 5)
    tl.em()
    assert abs(tl.lnl - om.lnl) <= RTOL * abs(om.lnl)
    assert np.allclose(tl.pi, om.pi, rtol=RTOL, atol=0) and np.allclose(tl.theta, om.theta, rtol=RTOL, atol=0)
    np.random.seed(3); a = tl.reassign_colsums('choose')
    np.random.seed(3); b = np.asarray(om.reassign('choose').sum(0)).ravel()
    assert np.array_equal(a, b)
    assert np.array_equal(tl.reassign_colsums('exclude'), np.asarray(om.reassign('exclude').sum(0)).ravel())
    assert np.allclose(tl.reassign_colsums('conf'), np.asarray(om.reassign('conf').sum(0)).ravel(), rtol=RTOL)


@pytest.mark.parametrize('dist', ['zipf', 'uniform'])
def test_config3_properties_and_shard_linearity(gpu_device, dist):
    """BASELINE config 3 shape (10M x 30k, ~40 nnz/row): size-independent checks.
       * sum(pi) = sum(theta) = 1: every ambiguous fragment's posteriors sum to 1, so the
         column sums add up to the total weight (a checksum of checksums);
       * the all-ones initial assignment counts every stored entry once;
       * sharding is linear: column sums of two half-shards add up to the full pass."""
    n, k = 10_000_000, 30000
    tl = _synthetic_tl(n, k, 40, dist)
    eng = tl._eng
    _, _, nnz = eng.dims()
    tl.em()
    assert abs(tl.pi.sum() - 1.0) < 1e-9 and abs(tl.theta.sum() - 1.0) < 1e-9
    assert tl.reassign_colsums('all', initial=True).sum() == nnz
    assert np.isfinite(tl.lnl)
    lnl_full, pi_full = tl.lnl, tl.pi.copy()
    eng.set_params(np.repeat(1. / k, k), np.repeat(1. / k, k))
    eng.em_pass()
    full = eng.read_reduce(0, k)
    del tl, eng
    parts = []
    for (a, b) in ((0, n // 2), (n // 2, n)):
        t2 = _synthetic_tl(n, k, 40, dist, r0=a, r1=b)
        t2._eng.em_pass()
        parts.append(t2._eng.read_reduce(0, k))
        del t2
    assert np.allclose(parts[0] + parts[1], full, rtol=1e-9, atol=0)


def test_em_is_reproducible_and_blocksize_independent(gpu_device):
    a = run_case('mid_zipf_20k')[2]
    b = run_case('mid_zipf_20k', block_rows=256)[2]
    assert np.allclose(a.pi, b.pi, rtol=1e-11, atol=0) and abs(a.lnl - b.lnl) <= 1e-11 * abs(a.lnl)
    assert a.n_iter == b.n_iter


def test_rccl_comm_path_single_rank(gpu_device):
    """The N > 1 plumbing on a real GPU with a 1-rank RCCL group (the library's own communicator, created from
    an id shipped over torch.distributed; see also tests/test_gpu_round2.py)."""
    import socket
    import torch
    import torch.distributed as dist
    from telescope_amd.distributed import Comm
    from telescope_amd.likelihood import TelescopeLikelihood
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        comm = Comm(device=0)
        for name in ('bundled', 'tiny_twins', 'mid_zipf_20k'):
            c = load_case(name)
            tl = TelescopeLikelihood(case_matrix(c), Opts(c), comm=comm)
            tl.em()
            assert tl.n_iter == int(c['n_iter'])
            assert abs(tl.lnl - float(c['lnl'])) <= RTOL * abs(float(c['lnl']))
            assert np.allclose(tl.pi, c['pi'], rtol=RTOL, atol=0)
            np.random.seed(int(c['seed']))
            assert np.array_equal(tl.reassign_colsums('choose'), c['ra_choose_0_colsum'])
            assert np.array_equal(tl.reassign_colsums('exclude'), c['ra_exclude_0_colsum'])
        comm.close()
    finally:
        dist.destroy_process_group()


def _oracle_vs_gpu(raw, iters=4, options=(), rtol=RTOL):
    """EM for `iters` fixed iterations on the GPU and in the oracle; returns the layout used."""
    from oracle.telescope_oracle import OracleModel
    from telescope_amd import _lib
    from telescope_amd.likelihood import TelescopeLikelihood, score_lut
    raw = sp.csr_matrix(raw)
    eng = _lib.Engine(0)
    for k, v in options:
        eng.set_option(k, v)
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), raw.shape[1], score_lut(int(raw.data.max())))
    tl = TelescopeLikelihood.from_engine(eng, Opts(max_iter=iters, em_epsilon=0.0))
    tl._raw = raw
    tl.em()
    om = OracleModel(raw)
    om.em(0.0, iters)
    assert abs(tl.lnl - om.lnl) <= rtol * abs(om.lnl)
    assert np.allclose(tl.pi, om.pi, rtol=rtol, atol=0) and np.allclose(tl.theta, om.theta, rtol=rtol, atol=0)
    assert np.array_equal(tl.reassign_colsums('exclude'), np.asarray(om.reassign('exclude').sum(0)).ravel())
    z = sp.csr_matrix(tl.z)
    assert np.allclose(z.data, sp.csr_matrix(om.z).data, rtol=rtol, atol=1e-300)
    return eng.layout_info()


@pytest.mark.parametrize('cols,expect_parts', [(9000, 2), (20000, 3), (30000, 4), (36000, 5), (40000, 6), (50000, 7), (60000, 8)])
def test_column_part_counts(gpu_device, cols, expect_parts):
    """Teams of P = 2, 3, 4 (two exchange waves) and P = 6 (three exchange waves) run the fused kernel."""
    from telescope_amd import synthetic
    ip, ix, rw = synthetic.generate(60000, cols, 24, seed=11, dist='zipf', uniq_frac=0.05)
    info = _oracle_vs_gpu(sp.csr_matrix((rw, ix, ip), shape=(60000, cols)))
    assert info['P'] == expect_parts
    assert info['fused'] == 1


def test_more_than_eight_column_parts_use_the_two_pass_kernels(gpu_device):
    """K = 130 000 -> 17 column parts of 7680: beyond the fused kernel's teams of 8 AND beyond its split layout (8 x 15 360), the
    two-pass form runs; K = 70 000 (round 3: two-pass) now stays on the fused kernel (split layout); option split = 0 gives the
    round-3 behaviour."""
    from telescope_amd import synthetic
    ip, ix, rw = synthetic.generate(30000, 130000, 30, seed=17, dist='zipf', uniq_frac=0.05)
    info = _oracle_vs_gpu(sp.csr_matrix((rw, ix, ip), shape=(30000, 130000)))
    assert info['P'] == 17 and info['fused'] == 0 and info['value_bytes'] == 8
    ip, ix, rw = synthetic.generate(30000, 70000, 30, seed=17, dist='zipf', uniq_frac=0.05)
    info = _oracle_vs_gpu(sp.csr_matrix((rw, ix, ip), shape=(30000, 70000)))
    assert 5 <= info['P'] <= 8 and info['fused'] == 1 and info['split'] == 1
    info = _oracle_vs_gpu(sp.csr_matrix((rw, ix, ip), shape=(30000, 70000)), options=(('split', 0),))
    assert info['P'] == 10 and info['fused'] == 0 and info['value_bytes'] == 8


def test_ragged_rows_and_kernel_variants(gpu_device):
    """Very uneven row lengths (1 .. 3000 entries, a few rows longer than the register tile of a
    fused sub-block) — exercises the block-size retry / two-pass fallback — and both EM kernels on
    the same matrix."""
    rng = np.random.RandomState(5)
    n, k = 30000, 12000
    lens = np.where(rng.rand(n) < 0.002, rng.randint(1500, 3000, n), rng.randint(1, 30, n))
    lens[::97] = 1
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.concatenate([np.sort(rng.choice(k, l, replace=False)) for l in lens]).astype(np.int32)
    data = rng.randint(100, 400, indptr[-1]).astype(np.uint16)
    raw = sp.csr_matrix((data, indices, indptr), shape=(n, k))
    a = _oracle_vs_gpu(raw, options=(('em_kernel', 2),))
    b = _oracle_vs_gpu(raw, options=(('em_kernel', 1),))
    assert b['fused'] == 0 and a['P'] == b['P'] == 2


@pytest.mark.parametrize('method', ['conf', 'all', 'unique', 'exclude', 'choose', 'average'])
def test_per_barcode_count_matrix(gpu_device, method):
    """scTelescope.output_report (model.py:611-625): `_assignments[_rows, :].sum(0).A1` per barcode —
    the segmented device pass against fancy indexing on the ORACLE's assignment matrix (and on the device's)."""
    from oracle.telescope_oracle import OracleModel
    c, raw, tl, _ = run_case('bundled')
    om = OracleModel(raw, float(c['pi_prior']), float(c['theta_prior']))
    om.em(float(c['em_epsilon']), int(c['max_iter']), use_likelihood=bool(c['use_likelihood']))
    rng = np.random.RandomState(3)
    bc = rng.randint(0, 9, tl.N)                         # barcode 8 = reads without a barcode
    groups = [np.flatnonzero(bc == g) for g in range(8)]
    groups[2] = np.concatenate([groups[2], groups[5][:40], groups[2][:3]])   # shared and repeated rows
    groups.append(np.zeros(0, np.int64))                 # an empty barcode
    np.random.seed(int(c['seed']))
    got = tl.reassign_group_sums(method, groups, 0.9)
    np.random.seed(int(c['seed']))
    mat = tl.reassign(method, 0.9)
    want = np.vstack([np.asarray(mat[g, :].sum(0)).ravel() for g in groups])
    np.random.seed(int(c['seed']))
    omat = sp.csr_matrix(om.reassign(method, 0.9))
    owant = np.vstack([np.asarray(omat[g, :].sum(0)).ravel() for g in groups])
    if method in ('conf', 'average'):
        assert np.allclose(got, want, rtol=1e-12, atol=1e-12) and np.allclose(got, owant, rtol=RTOL, atol=1e-12)
    else:
        assert np.array_equal(got, want) and np.array_equal(got, owant)


def test_device_log1p_matches_libm(gpu_device):
    """The lnl passes' log1p (finite x >= 0) against numpy's: <= 2 ulp over 1e-320 .. 1e300."""
    from telescope_amd import _lib
    rng = np.random.RandomState(1)
    x = np.concatenate([10.0 ** rng.uniform(-320, 300, 200000), 10.0 ** rng.uniform(-3, 3, 200000),
                        rng.uniform(0, 3, 100000), [0.0, 5e-324, 2.0 ** -28, 2.0 ** -28 * (1 - 1e-16), 1.0, 2.0 ** 0.5 - 1,
                                                    2.0 ** 53, 1e43, 1.7e308]])
    y = _lib.debug_log1p(x)
    ref = np.log1p(x)
    err = np.abs(y - ref) / np.maximum(np.spacing(ref), 5e-324)
    assert np.all(np.isfinite(y)) and err.max() <= 2.0, (err.max(), x[err.argmax()])


def test_fast_log1p_of_the_fused_lnl_pass(gpu_device):
    """fz_log1p_tab (64-entry table + series): what a sum of z * log1p(x) needs is ABSOLUTE accuracy — <= 4e-16 here, i.e.
    a few ulp of values around 1 — and relative accuracy where log1p is small (first table entry is exact: <= 4 ulp below
    x = 1/64)."""
    from telescope_amd import _lib
    rng = np.random.RandomState(2)
    x = np.concatenate([10.0 ** rng.uniform(-320, 300, 200000), 10.0 ** rng.uniform(-3, 3, 300000), rng.uniform(0, 3, 200000),
                        1.0 + np.arange(64) / 64.0 - 1.0, np.nextafter(1.0 + np.arange(1, 64) / 64.0, 0) - 1.0,
                        [0.0, 5e-324, 2.0 ** -28, 2.0 ** -53, 2.0 ** -52, 1.0, 2.0 ** 0.5 - 1, 2.0 ** 53, 1e43, 1.7e308]])
    y = _lib.debug_log1p(x, table=True)
    ref = np.log1p(x)
    assert np.all(np.isfinite(y))
    abs_err = np.abs(y - ref)
    rel_ulp = abs_err / np.maximum(np.spacing(ref), 5e-324)
    assert (abs_err / np.maximum(1.0, np.abs(ref))).max() <= 4e-16, (abs_err.max(), x[abs_err.argmax()])
    assert rel_ulp.max() <= 64.0, (rel_ulp.max(), x[rel_ulp.argmax()])
    small = x < 1.0 / 64
    assert rel_ulp[small].max() <= 4.0, (rel_ulp[small].max(), x[small][rel_ulp[small].argmax()])


def _random_csr(rng, n, k, max_len, lo, hi, hot_frac=0.0):
    lens = rng.randint(2, max_len, n)
    indptr = np.concatenate([[0], np.cumsum(lens)])
    rows = []
    for l in lens:
        c = rng.choice(k - 1, l, replace=False) + 1
        if rng.rand() < hot_frac:
            c[0] = 0                                      # column 0 (`__no_feature`) in most rows
        rows.append(np.sort(c))
    indices = np.concatenate(rows).astype(np.int32)
    data = rng.randint(lo, hi, indptr[-1]).astype(np.uint16)
    return sp.csr_matrix((data, indices, indptr), shape=(n, k))


@pytest.mark.parametrize('fmt,expect_bytes', [(0, 2), (1, 8), (2, 2)])
def test_entry_formats_agree_with_oracle(gpu_device, fmt, expect_bytes):
    """uint16 score codes + LDS score table (default) and fp64 entries run the same arithmetic."""
    raw = _random_csr(np.random.RandomState(11), 40000, 20000, 40, 120, 330)
    info = _oracle_vs_gpu(raw, options=(('value_format', fmt),))
    assert info['fused'] == 1 and info['value_bytes'] == expect_bytes


def test_large_score_table_falls_back_to_fp64_entries(gpu_device):
    """Raw scores up to 6000 -> a 6001-entry table does not go to LDS: fp64 entries, same results;
    asking for codes explicitly is an error, not a silent change of format."""
    from telescope_amd import _lib
    from telescope_amd.likelihood import TelescopeLikelihood, score_lut
    raw = _random_csr(np.random.RandomState(12), 20000, 9000, 30, 1000, 6000)
    info = _oracle_vs_gpu(raw)
    assert info['fused'] == 1 and info['value_bytes'] == 8
    eng = _lib.Engine(0)
    eng.set_option('value_format', 2)
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), raw.shape[1], score_lut(int(raw.data.max())))
    with pytest.raises(_lib.EngineError):
        TelescopeLikelihood.from_engine(eng, Opts(max_iter=2, em_epsilon=0.0))


@pytest.mark.parametrize('hot_split', [1, 0])
def test_hot_column_gets_several_accumulator_slots(gpu_device, hot_split):
    """A column present in 80 % of the rows is split over several LDS accumulator slots (or not, with
    the option off); the column sums are the same either way."""
    raw = _random_csr(np.random.RandomState(13), 50000, 16000, 24, 139, 300, hot_frac=0.8)
    info = _oracle_vs_gpu(raw, options=(('hot_split', hot_split),))
    assert info['fused'] == 1 and info['P'] == 3
    assert (info['hot_cols'] >= 1) == bool(hot_split)


def test_rows_longer_than_a_wave_of_quads(gpu_device):
    """Rows with hundreds of entries in one column part: the row-sum carry runs across many lanes
    (and across wave boundaries) of the fused kernel's phase 1."""
    rng = np.random.RandomState(14)
    n, k = 6000, 7000                                    # one column part: every row's entries stay together
    lens = np.where(rng.rand(n) < 0.05, rng.randint(300, 900, n), rng.randint(2, 12, n))
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.concatenate([np.sort(rng.choice(k, l, replace=False)) for l in lens]).astype(np.int32)
    data = rng.randint(139, 300, indptr[-1]).astype(np.uint16)
    raw = sp.csr_matrix((data, indices, indptr), shape=(n, k))
    for fmt in (1, 2):
        info = _oracle_vs_gpu(raw, options=(('value_format', fmt),))
        assert info['fused'] == 1 and info['P'] == 1


@pytest.mark.parametrize('seed', list(range(24)) + [100 + s for s in range(12)])
def test_random_shapes_against_oracle(gpu_device, seed):
    """Small random matrices of every shape the layout code branches on: 1..5 column parts, few or
    many blocks (fewer blocks than teams, a single block, blocks of very different fill), rows with
    1..120 entries, 0..60 % unique rows, narrow and wide score ranges, forced block sizes, both
    entry formats, priors on and off."""
    rng = np.random.RandomState(1000 + seed)
    k = int(rng.choice([3, 17, 200, 5000, 9000, 16000, 24000, 33000]))
    n = int(rng.choice([1, 7, 300, 2500, 12000, 40000]))
    max_len = int(min(k, rng.choice([2, 5, 30, 120])))
    uniq = float(rng.choice([0.0, 0.1, 0.6]))
    lens = np.where(rng.rand(n) < uniq, 1, rng.randint(1, max_len + 1, n))
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.concatenate([np.sort(rng.choice(k, l, replace=False)) for l in lens]).astype(np.int32)
    lo, hi = [(139, 212), (1, 5), (100, 1500), (60000, 65535)][int(rng.randint(4))]
    data = rng.randint(lo, hi + 1, indptr[-1]).astype(np.uint16)
    raw = sp.csr_matrix((data, indices, indptr), shape=(n, k))
    options = [('value_format', int(rng.randint(0, 2)))]
    if rng.rand() < 0.3:
        options.append(('block_rows', int(rng.choice([64, 128, 256]))))
    if rng.rand() < 0.2:
        options.append(('em_kernel', 1))
    if seed >= 100:                                          # round 4: the same shapes on the split layout (two light passes per iteration)
        options = [kv for kv in options if kv[0] != 'em_kernel'] + [('split', 1), ('parts', int(rng.randint(5, 9)))]
    from oracle.telescope_oracle import OracleModel
    from telescope_amd import _lib
    from telescope_amd.likelihood import TelescopeLikelihood, score_lut
    o = Opts(max_iter=int(rng.randint(1, 6)), em_epsilon=0.0)
    o.pi_prior, o.theta_prior = [(0, 200000), (0, 0), (5, 1000)][int(rng.randint(3))]
    eng = _lib.Engine(0)
    for key, v in options:
        eng.set_option(key, v)
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), k, score_lut(int(raw.data.max())))
    tl = TelescopeLikelihood.from_engine(eng, o)
    tl._raw = raw
    tl.em()
    om = OracleModel(raw, o.pi_prior, o.theta_prior)
    om.em(0.0, o.max_iter)
    ctx = (seed, n, k, max_len, uniq, (lo, hi), options, eng.layout_info())
    assert abs(tl.lnl - om.lnl) <= RTOL * max(abs(om.lnl), 1e-300), ctx
    assert np.allclose(tl.pi, om.pi, rtol=RTOL, atol=1e-300), ctx
    assert np.allclose(tl.theta, om.theta, rtol=RTOL, atol=1e-300), ctx
    for method in ('exclude', 'all', 'unique'):
        assert np.array_equal(tl.reassign_colsums(method), np.asarray(om.reassign(method).sum(0)).ravel()), (method, ctx)
    from oracle.em_fused import em_fused                  # the independently written C restatement agrees too
    rc = em_fused(raw, o.pi_prior, o.theta_prior, 0.0, o.max_iter)
    assert np.allclose(tl.pi, rc['pi'], rtol=RTOL, atol=1e-300) and abs(tl.lnl - rc['lnl']) <= RTOL * max(abs(rc['lnl']), 1e-300), ctx


def test_more_than_2_31_entries_on_one_gpu(gpu_device):
    """60M x 30k x ~40 = 2.4e9 stored entries (> 2^31): 64-bit offsets everywhere; the fused kernel with
    both entry formats and the two-pass kernels agree to summation-order noise."""
    res = []
    for options in ((('value_format', 1),), (('value_format', 2),), (('em_kernel', 1),)):
        tl = _synthetic_tl(60_000_000, 30000, 40, 'zipf', options=options, opts=Opts(max_iter=3, em_epsilon=0.0))
        tl.em()
        assert tl._eng.dims()[2] > 2 ** 31
        res.append((tl.lnl, tl.pi.copy(), tl.reassign_colsums('exclude'), tl._eng.layout_info()['fused']))
        del tl
    assert [r[3] for r in res] == [1, 1, 0]
    for r in res[1:]:
        assert abs(r[0] - res[0][0]) <= 1e-12 * abs(res[0][0])
        assert np.allclose(r[1], res[0][1], rtol=1e-11, atol=0) and np.array_equal(r[2], res[0][2])


class _ThreadComm(object):
    """Two 'ranks' as two threads of this process, each with its own engine on the SAME GPU (null stream,
    so their kernels serialise): the collectives of telescope_amd.distributed.Comm on shared memory.
    Exercises the whole row-sharded protocol — global score table, setup sums, column signatures, the
    per-iteration all-reduce of the device reduce buffers, lnl, reassign sums, choose picks — on real
    engines, which the 2-rank gloo test can only do with the oracle standing in for the device."""
    import threading
    _lock = threading.Lock()

    def __init__(self, rank, world, shared):
        import threading
        self.rank, self.world, self.sh = rank, world, shared
        self.device = 0

    def _exchange(self, key, value, combine):
        sh = self.sh
        sh['slots'][self.rank] = value
        sh['barrier'].wait()
        if self.rank == 0:
            sh[key] = combine(sh['slots'])
        sh['barrier'].wait()
        out = sh[key]
        sh['barrier'].wait()
        return out

    def max_scalar(self, v):
        return int(self._exchange('r', int(v), lambda xs: max(xs)))

    def sum_array(self, a):
        return self._exchange('r', np.asarray(a, np.float64), lambda xs: np.sum(xs, axis=0)).copy()

    def max_array(self, a):
        return self._exchange('r', np.asarray(a, np.float64), lambda xs: np.max(xs, axis=0)).copy()

    def sum_array_u64(self, a):
        return self._exchange('r', np.asarray(a, np.uint64), lambda xs: np.sum(np.array(xs, dtype=np.uint64), axis=0,
                                                                               dtype=np.uint64)).copy()

    def gather_rows(self, a):
        return self._exchange('r', np.asarray(a), lambda xs: list(xs))

    def scatter_rows(self, parts):
        return self._exchange('r', parts if self.rank == 0 else None, lambda xs: xs[0])[self.rank]

    def barrier(self):
        self.sh['barrier'].wait()

    def attach(self, engine, n_cols):
        import torch
        self._red = torch.zeros(n_cols + 2, dtype=torch.float64, device='cuda:0')
        self.sh['red'][self.rank] = self._red
        engine.bind_reduce_buffer(self._red.data_ptr(), n_cols + 2)

    def allreduce_device(self, engine, offset=0, count=None):
        import torch
        sh = self.sh
        sh['barrier'].wait()
        if self.rank == 0:
            torch.cuda.synchronize()
            tot = sh['red'][0] + sh['red'][1]
            for r in sh['red']:
                r.copy_(tot)
            torch.cuda.synchronize()
        sh['barrier'].wait()


@pytest.mark.parametrize('fmt', [1, 2])
def test_two_row_shards_on_one_gpu(gpu_device, fmt):
    """Row-sharded EM with two real engines (rows split 45 / 55 %) against one engine on all rows."""
    import threading
    from telescope_amd import _lib, synthetic
    from telescope_amd.likelihood import TelescopeLikelihood
    N, K = 300000, 30000
    cdf = synthetic.poisson_cdf_u32(24)
    o = Opts(max_iter=6, em_epsilon=0.0)

    def make(r0, r1, comm):
        eng = _lib.Engine(0)
        eng.set_option('value_format', fmt)
        eng.set_option('row_offset', r0)
        eng.generate(r0, r1, K, cdf, 9, synthetic.DIST_CODE['zipf'], 0.08)
        return TelescopeLikelihood.from_engine(eng, o, comm)

    ref = make(0, N, None)
    ref.em()
    np.random.seed(4); ref_choose = ref.reassign_colsums('choose')
    ref_excl, ref_conf = ref.reassign_colsums('exclude'), ref.reassign_colsums('conf')
    shared = dict(slots=[None, None], barrier=threading.Barrier(2), red=[None, None])
    cut = int(N * 0.45)
    out, errs = [None, None], []

    def worker(rank):
        try:
            comm = _ThreadComm(rank, 2, shared)
            tl = make(0 if rank == 0 else cut, cut if rank == 0 else N, comm)
            tl.em()
            if rank == 0:
                np.random.seed(4)                          # only rank 0 draws (it holds the RNG stream)
            out[rank] = dict(pi=tl.pi.copy(), theta=tl.theta.copy(), lnl=tl.lnl, n=tl.N,
                             choose=tl.reassign_colsums('choose'), excl=tl.reassign_colsums('exclude'),
                             conf=tl.reassign_colsums('conf'))
        except Exception as e:   # noqa: BLE001
            errs.append(e)
            shared['barrier'].abort()
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errs, errs
    assert out[0]['n'] + out[1]['n'] == N
    for r in out:                                          # every rank holds the global result
        assert abs(r['lnl'] - ref.lnl) <= 1e-12 * abs(ref.lnl)
        assert np.allclose(r['pi'], ref.pi, rtol=1e-11, atol=0) and np.allclose(r['theta'], ref.theta, rtol=1e-11, atol=0)
        assert np.array_equal(r['excl'], ref_excl) and np.array_equal(r['choose'], ref_choose)
        assert np.allclose(r['conf'], ref_conf, rtol=1e-10, atol=1e-12)
    assert np.array_equal(out[0]['pi'], out[1]['pi'])


def test_config4_full_size_properties(gpu_device):
    """BASELINE config 4 (50M x 30k x ~40, 5 % unique rows) through size-independent properties:
    pi and theta are distributions; `all` counts every stored entry, `unique` every unique row, `choose`
    and `average` one per fragment; `exclude` never exceeds `choose` on any locus."""
    tl = _synthetic_tl(50_000_000, 30000, 40, 'zipf', uniq=0.05, opts=Opts(max_iter=3, em_epsilon=0.0))
    tl.em()
    n, k, nnz = tl._eng.dims()
    info = tl._eng.layout_info()
    assert abs(tl.pi.sum() - 1.0) <= 1e-12 and abs(tl.theta.sum() - 1.0) <= 1e-12
    assert np.all(tl.pi > 0) and np.isfinite(tl.lnl)
    assert int(tl.reassign_colsums('all').sum()) == nnz
    assert int(tl.reassign_colsums('unique').sum()) == info['N_uni'] == n - info['N_amb']
    np.random.seed(2)
    choose, excl = tl.reassign_colsums('choose'), tl.reassign_colsums('exclude')
    assert int(choose.sum()) == n and np.all(excl <= choose) and int(excl.sum()) <= n
    assert abs(tl.reassign_colsums('average').sum() - n) <= 1e-9 * n
    conf = tl.reassign_colsums('conf', 0.9)
    assert 0 < conf.sum() <= n * (1 + 1e-12)


def test_config3_entry_formats_agree(gpu_device):
    """BASELINE config 3 (10M x 30k x ~40) asks for a precision sweep of the stored values.  No reduced
    precision is offered: 2-byte score codes + the fp64 score table are SMALLER than fp32 values and give
    the same fp64 numbers — the two layouts must agree to summation-order noise at full size."""
    res = []
    for fmt in (1, 2):
        tl = _synthetic_tl(10_000_000, 30000, 40, 'zipf', options=(('value_format', fmt),),
                           opts=Opts(max_iter=8, em_epsilon=0.0))
        tl.em()
        res.append((tl.pi.copy(), tl.theta.copy(), tl.lnl, tl._eng.layout_info()['value_bytes'],
                    tl.reassign_colsums('exclude')))
    assert (res[0][3], res[1][3]) == (8, 2)
    assert np.allclose(res[0][0], res[1][0], rtol=1e-11, atol=0) and np.allclose(res[0][1], res[1][1], rtol=1e-11, atol=0)
    assert abs(res[0][2] - res[1][2]) <= 1e-12 * abs(res[0][2])
    assert np.array_equal(res[0][4], res[1][4])


def test_fused_and_twopass_agree_at_scale(gpu_device):
    """5M x 30k x 40: the two EM kernels give the same parameters (summation order aside)."""
    res = []
    for kern in (1, 2):
        tl = _synthetic_tl(5_000_000, 30000, 40, 'zipf', options=(('em_kernel', kern),))
        tl.em()
        res.append((tl.pi.copy(), tl.lnl, tl._eng.layout_info()['fused']))
    assert res[0][2] == 0 and res[1][2] == 1
    assert np.allclose(res[0][0], res[1][0], rtol=1e-10, atol=0)
    assert abs(res[0][1] - res[1][1]) <= 1e-11 * abs(res[0][1])
