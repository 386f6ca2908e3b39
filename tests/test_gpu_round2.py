"""GPU tests added in round 2: the chunked EM loop (device-side convergence), the library's own RCCL
communicator, recovery from a hand-off time-out of the persistent kernel, reassign() on a caller-assigned
z, config-5-shaped matrices (K = 50 000, ~100 entries per row), a full-size (50M rows) comparison against
the C oracle, and run-to-run determinism of the integer outputs.

Tolerances as in test_gpu_parity.py: north_star's bar is 1e-4 relative on lnl / final counts and bit-exact
integer outputs; float checks here use RTOL = 1e-9, integer outputs array_equal."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import Opts, case_matrix, load_case

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _engine_for(raw, options=()):
    from telescope_amd import _lib
    from telescope_amd.likelihood import score_lut
    raw = sp.csr_matrix(raw)
    eng = _lib.Engine(0)
    for k, v in options:
        eng.set_option(k, v)
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), raw.shape[1], score_lut(int(raw.data.max())))
    return eng


def _tl_for(raw, opts, options=(), comm=None):
    from telescope_amd.likelihood import TelescopeLikelihood
    tl = TelescopeLikelihood.from_engine(_engine_for(raw, options), opts, comm)
    tl._raw = sp.csr_matrix(raw)
    return tl


def _synthetic_tl(rows, cols, d, dist, seed=42, uniq=0.0, r0=0, r1=None, options=(), opts=None, comm=None):
    from telescope_amd import _lib, synthetic
    from telescope_amd.likelihood import TelescopeLikelihood
    eng = _lib.Engine(0)
    for k, v in options:
        eng.set_option(k, v)
    eng.set_option('row_offset', r0)
    eng.generate(r0, rows if r1 is None else r1, cols, synthetic.poisson_cdf_u32(d), seed,
                 synthetic.DIST_CODE[dist], uniq)
    return TelescopeLikelihood.from_engine(eng, opts or Opts(max_iter=5, em_epsilon=0.0), comm)


# ---------------------------------------------------------------------------------------------------
# chunked loop == one iteration per host round trip
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name,use_lnl', [('mid_zipf_20k', False), ('bundled', False), ('bundled', True), ('tiny_twins', False)])
def test_chunked_loop_equals_stepwise(gpu_device, name, use_lnl):
    """`Engine.em_chunk` (convergence decided on the device, kernels behind the converging iteration return
    at once) leaves exactly the state of the host-driven loop: iteration count, every diff_est, parameters."""
    from telescope_amd._lib import Z_CUR, Z_FIRST, Z_PREV
    c = load_case(name)
    raw = case_matrix(c)
    o = Opts(c)
    a, b = _engine_for(raw), _engine_for(raw)
    from telescope_amd.likelihood import TelescopeLikelihood
    ta, tb = TelescopeLikelihood.from_engine(a, o), TelescopeLikelihood.from_engine(b, o)
    # host-driven reference loop on engine a (model.py:771-797)
    diffs_a, lnls_a, lnl_prev, inum, conv = [], [], float('inf'), 0, False
    while not (conv or inum >= o.max_iter):
        a.em_pass(); d = a.em_update(); inum += 1
        diffs_a.append(d)
        if inum == 1:
            first_a = a.get_params(Z_CUR)
        if use_lnl:
            a.lnl_pass(); l = float(a.read_reduce(a.dims()[1], 1)[0]); lnls_a.append(l)
            conv = abs(l - lnl_prev) < o.em_epsilon; lnl_prev = l
        else:
            conv = d < o.em_epsilon
    # chunked on engine b, deliberately over-asking (chunk of 13 > remaining iterations at the end)
    diffs_b, lnls_b, stopped, first = [], [], False, True
    while not stopped and len(diffs_b) < o.max_iter:
        d, l, stopped = b.em_chunk(min(13, o.max_iter - len(diffs_b)), o.em_epsilon, use_lnl, first=first)
        first = False
        diffs_b += list(d); lnls_b += list(l) if use_lnl else []
    assert len(diffs_b) == inum
    if bool(c['use_likelihood']) == use_lnl:
        assert inum == int(c['n_iter'])
    assert np.allclose(diffs_a, diffs_b, rtol=1e-9, atol=1e-12)     # diff_est of late iterations is ~1e-8: atomics-order noise
    if use_lnl:
        assert np.allclose(lnls_a, lnls_b, rtol=1e-12, atol=0)
    assert stopped == conv
    for which in (Z_CUR, Z_PREV):
        pa, ta_ = a.get_params(which); pb, tb_ = b.get_params(which)
        assert np.allclose(pa, pb, rtol=1e-10, atol=0) and np.allclose(ta_, tb_, rtol=1e-10, atol=0)
    pf, tf = b.get_params(Z_FIRST)
    assert np.allclose(pf, first_a[0], rtol=1e-12, atol=0) and np.allclose(tf, first_a[1], rtol=1e-12, atol=0)
    if not use_lnl and not bool(c['use_likelihood']):
        assert abs(b.final_lnl() - float(c['lnl'])) <= RTOL * abs(float(c['lnl']))


def test_overshoot_costs_nothing(gpu_device):
    """Asking for 64 iterations when the run converges after a few: the iterations enqueued behind the
    converging one must not change anything (n_iter, pi, lnl as in the golden run)."""
    c = load_case('bundled')
    eng = _engine_for(case_matrix(c))
    from telescope_amd.likelihood import TelescopeLikelihood
    tl = TelescopeLikelihood.from_engine(eng, Opts(c))
    diffs, _, stopped = eng.em_chunk(64, 1e-7, False, first=True)
    assert stopped and len(diffs) == int(c['n_iter']) == 16
    pi, _ = eng.get_params(1)
    assert np.allclose(pi, c['pi'], rtol=RTOL, atol=0)
    assert abs(eng.final_lnl() - 95252.596293) < 1e-5


# ---------------------------------------------------------------------------------------------------
# the library's own RCCL communicator (1 rank on the test box)
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def rccl_comm(gpu_device):
    import socket
    import torch
    import torch.distributed as dist
    from telescope_amd.distributed import Comm
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    comm = Comm(device=0)
    yield comm
    comm.close()
    dist.destroy_process_group()


def test_library_communicator_single_rank(rccl_comm):
    """N > 1 code path on a real GPU with a 1-rank group: the library creates its own RCCL communicator from an
    id shipped over torch.distributed, em() runs chunked with ncclAllReduce on the engine's stream, the setup /
    reassign sums go through the same communicator, and the public mstep(z) / calculate_lnl are sharded."""
    comm = rccl_comm
    assert comm.lib is not None
    assert np.array_equal(comm.sum_array(np.arange(5.0)), np.arange(5.0))
    assert comm.max_scalar(7) == 7
    big = np.array([2 ** 63 + 5, 3], np.uint64)
    assert np.array_equal(comm.sum_array_u64(big), big)
    for name in ('bundled', 'tiny_twins', 'mid_zipf_20k'):
        c = load_case(name)
        tl = _tl_for(case_matrix(c), Opts(c), comm=comm)
        assert comm.in_library
        tl.em(use_likelihood=bool(c['use_likelihood']))
        assert tl.n_iter == int(c['n_iter'])
        assert abs(tl.lnl - float(c['lnl'])) <= RTOL * abs(float(c['lnl']))
        assert np.allclose(tl.pi, c['pi'], rtol=RTOL, atol=0) and np.allclose(tl.pi_init, c['pi_init'], rtol=RTOL, atol=0)
        np.random.seed(int(c['seed']))
        assert np.array_equal(tl.reassign_colsums('choose'), c['ra_choose_0_colsum'])
        assert np.array_equal(tl.reassign_colsums('exclude'), c['ra_exclude_0_colsum'])
        z = tl.z
        pi_hat, theta_hat = tl.mstep(z)                     # sharded public M-step (all-reduce inside the library)
        assert np.allclose(pi_hat, tl.pi, rtol=RTOL, atol=0) and np.allclose(theta_hat, tl.theta, rtol=RTOL, atol=0)
        assert abs(tl.calculate_lnl(z, tl.pi, tl.theta) - tl.lnl) <= RTOL * abs(tl.lnl)


def test_use_likelihood_through_the_communicator(rccl_comm):
    from oracle.telescope_oracle import OracleModel
    from telescope_amd import synthetic
    ip, ix, rw = synthetic.generate(30000, 9000, 12, seed=3, dist='zipf', uniq_frac=0.1)
    raw = sp.csr_matrix((rw, ix, ip), shape=(30000, 9000))
    o = Opts(max_iter=40, em_epsilon=1e-3)
    tl = _tl_for(raw, o, comm=rccl_comm)
    tl.em(use_likelihood=True)
    om = OracleModel(raw)
    om.em(o.em_epsilon, o.max_iter, use_likelihood=True)
    assert tl.n_iter == om.n_iter and abs(tl.lnl - om.lnl) <= RTOL * abs(om.lnl)
    assert np.allclose(tl.pi, om.pi, rtol=RTOL, atol=0)


# ---------------------------------------------------------------------------------------------------
# hand-off time-out of the persistent kernel -> two-pass kernels, run continues
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('bit,use_lnl', [(32, False), (64, False), (32, True), (64, True)])
def test_timeout_recovery_switches_to_two_pass(gpu_device, bit, use_lnl):
    """fused_dbg bit 5 / 6 makes the fused EM / lnl pass behave like a watchdog time-out (error word set, column
    sums not written).  The update kernel must not commit, the handle must rebuild its layout for the two-pass
    kernels, redo the step and finish with the golden result."""
    c = load_case('mid_zipf_20k')
    raw = case_matrix(c)
    o = Opts(c, max_iter=12) if use_lnl else Opts(c)
    ref = _tl_for(raw, o)
    ref.em(use_likelihood=use_lnl)
    tl = _tl_for(raw, o, options=(('fused_dbg', bit),))
    assert tl._eng.layout_info()['fused'] == 1
    tl.em(use_likelihood=use_lnl)
    assert tl._eng.layout_info()['fused'] == 0              # fell back
    assert tl.n_iter == ref.n_iter
    assert abs(tl.lnl - ref.lnl) <= 1e-11 * abs(ref.lnl)
    assert np.allclose(tl.pi, ref.pi, rtol=1e-10, atol=0) and np.allclose(tl.pi_init, ref.pi_init, rtol=1e-10, atol=0)
    assert np.array_equal(tl.reassign_colsums('exclude'), ref.reassign_colsums('exclude'))
    if not use_lnl:
        assert abs(tl.lnl - float(c['lnl'])) <= RTOL * abs(float(c['lnl']))


def test_timeout_in_the_stepwise_api_leaves_parameters_untouched(gpu_device):
    """Hosts that drive pass / update themselves: tsem_em_update reports the time-out WITHOUT committing;
    tsem_recover_timeout switches the failing handle; the redone step gives the regular result."""
    from telescope_amd import _lib
    c = load_case('mid_zipf_20k')
    raw = case_matrix(c)
    good = _tl_for(raw, Opts(c))
    good._eng.em_pass(); d_good = good._eng.em_update()
    tl = _tl_for(raw, Opts(c), options=(('fused_dbg', 32),))
    eng = tl._eng
    before = eng.get_params(_lib.Z_CUR)
    eng.em_pass()
    with pytest.raises(_lib.EngineError) as ei:
        eng.em_update()
    assert ei.value.code == _lib.ERR_TIMEOUT
    after = eng.get_params(_lib.Z_CUR)
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    assert eng.recover_timeout() is True and eng.layout_info()['fused'] == 0
    eng.em_pass(); d = eng.em_update()
    assert abs(d - d_good) <= 1e-10 * d_good
    assert np.allclose(eng.get_params(_lib.Z_CUR)[0], good._eng.get_params(_lib.Z_CUR)[0], rtol=1e-10, atol=0)


def test_two_engines_on_two_streams(gpu_device):
    """two handles on two streams of one GPU.  Their persistent kernels cannot both be
    resident; whichever way the hardware schedules them (one after the other, or a time-out followed by the
    two-pass fallback) both runs must finish with the right answer."""
    import threading
    import torch
    c = load_case('mid_zipf_20k')
    raw = case_matrix(c)
    out, errs = [None, None], []

    def worker(i):
        try:
            st = torch.cuda.Stream(device=0)
            tl = _tl_for(raw, Opts(c))
            tl._eng.set_stream(st.cuda_stream)
            tl.em()
            out[i] = (tl.n_iter, tl.lnl, tl.pi.copy())
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errs, errs
    for n, lnl, pi in out:
        assert n == int(c['n_iter']) and abs(lnl - float(c['lnl'])) <= RTOL * abs(float(c['lnl']))
        assert np.allclose(pi, c['pi'], rtol=RTOL, atol=0)


# ---------------------------------------------------------------------------------------------------
# reassign() on a caller-assigned z (model.py:837 reads whatever self.z holds)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['bundled', 'tiny_ties', 'mid_zipf_20k'])
def test_reassign_on_caller_assigned_z(gpu_device, name):
    from oracle.telescope_oracle import OracleModel
    c = load_case(name)
    raw = case_matrix(c)
    tl = _tl_for(raw, Opts(c))
    tl.em()
    om = OracleModel(raw, float(c['pi_prior']), float(c['theta_prior']))
    # a z the engine never produced: the E-step of perturbed parameters, with a few entries removed
    rng = np.random.RandomState(5)
    pi = om.pi * rng.uniform(0.5, 1.5, om.K); pi /= pi.sum()
    z = sp.csr_matrix(om.estep(pi, om.theta))
    z.data[rng.rand(z.nnz) < 0.02] = 0.0
    z.eliminate_zeros()
    tl.z = z
    om.z = z
    for meth in ('exclude', 'choose', 'average', 'conf', 'unique', 'all'):
        np.random.seed(11); got = tl.reassign(meth, 0.9).sum(0).A1
        np.random.seed(11); want = np.asarray(om.reassign(meth, 0.9).sum(0)).ravel()
        if meth in ('average', 'conf'):
            assert np.allclose(got, want, rtol=RTOL, atol=1e-12), meth
        else:
            assert np.array_equal(got, want), meth
    np.random.seed(11); m = sp.csr_matrix(tl.reassign('choose', 0.9).tocsr())
    np.random.seed(11); w = sp.csr_matrix(om.reassign('choose', 0.9))
    assert (m != w).nnz == 0
    # update_sam's consumer side (model.py:479-521): scalar lookups tl.z[ridx, fidx] and mat[ridx, fidx]
    r, f = int(z.nonzero()[0][7]), int(z.nonzero()[1][7])
    assert tl.z[r, f] == z[r, f]


def test_update_sam_lookups_after_em(gpu_device):
    """model.py:483,508-511: `tl.reassign(...)[ridx, fidx]` and `tl.z[ridx, fidx]` per alignment; phred(prob)
    of helpers.py:14-37 on those values."""
    from oracle.telescope_oracle import OracleModel
    c = load_case('bundled')
    raw = case_matrix(c)
    tl = _tl_for(raw, Opts(c))
    tl.em()
    om = OracleModel(raw)
    om.em(1e-7, 100)
    mat = tl.reassign('exclude', 0.9)
    omat = sp.csr_matrix(om.reassign('exclude', 0.9))
    zz = sp.csr_matrix(om.z)
    rows, cols = raw.nonzero()
    for k in range(0, len(rows), 397):
        r, f = int(rows[k]), int(cols[k])
        assert mat[r, f] == omat[r, f]
        assert abs(tl.z[r, f] - zz[r, f]) <= RTOL * max(zz[r, f], 1e-300)
        prob = tl.z[r, f]
        phred = 255 if prob >= 1 else int(round(-10 * np.log10(1 - prob))) if prob < 1 else 255   # helpers.py:14-37
        oprob = zz[r, f]
        ophred = 255 if oprob >= 1 else int(round(-10 * np.log10(1 - oprob)))
        assert phred == ophred


# ---------------------------------------------------------------------------------------------------
# config 5 shape: K = 50 000 loci, ~100 stored entries per row
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('fmt', [0, 1, 2])
@pytest.mark.parametrize('parts', [0, 7, 8])
def test_config5_shape_against_oracle(gpu_device, fmt, parts):
    """100k x 50k x ~100 against the oracle: every entry format, teams of 7 and 8 column parts (auto = 8)."""
    from oracle.telescope_oracle import OracleModel
    from telescope_amd import synthetic
    n, k = 100_000, 50_000
    ip, ix, rw = synthetic.generate(n, k, 100, seed=21, dist='zipf', uniq_frac=0.02)
    raw = sp.csr_matrix((rw, ix, ip), shape=(n, k))
    opts = (('value_format', fmt),) + ((('parts', parts),) if parts else ())
    tl = _tl_for(raw, Opts(max_iter=4, em_epsilon=0.0), options=opts)
    info = tl._eng.layout_info()
    assert info['fused'] == 1 and info['P'] == (parts or 8)
    assert info['value_bytes'] == (8 if fmt == 1 else 2)
    tl.em()
    om = OracleModel(raw)
    om.em(0.0, 4)
    assert abs(tl.lnl - om.lnl) <= RTOL * abs(om.lnl)
    assert np.allclose(tl.pi, om.pi, rtol=RTOL, atol=0) and np.allclose(tl.theta, om.theta, rtol=RTOL, atol=0)
    assert np.array_equal(tl.reassign_colsums('exclude'), np.asarray(om.reassign('exclude').sum(0)).ravel())
    assert np.allclose(tl.reassign_colsums('conf'), np.asarray(om.reassign('conf').sum(0)).ravel(), rtol=RTOL, atol=1e-12)


def test_config5_per_gpu_shard_properties(gpu_device):
    """The per-GPU shard of config 5 (25M x 50k x ~100 = 2.5e9 stored entries): size-independent properties
    and agreement of the fused kernel (both entry formats) with the two-pass kernels."""
    res = []
    for options in ((('value_format', 2),), (('value_format', 1),), (('em_kernel', 1),)):
        tl = _synthetic_tl(25_000_000, 50_000, 100, 'zipf', uniq=0.02, options=options, opts=Opts(max_iter=3, em_epsilon=0.0))
        tl.em()
        n, k, nnz = tl._eng.dims()
        info = tl._eng.layout_info()
        assert abs(tl.pi.sum() - 1.0) <= 1e-12 and abs(tl.theta.sum() - 1.0) <= 1e-12 and np.isfinite(tl.lnl)
        if not res:
            assert nnz > 2 ** 31 and info['P'] == 8
            assert int(tl.reassign_colsums('all').sum()) == nnz
            assert int(tl.reassign_colsums('unique').sum()) == info['N_uni']
        res.append((tl.lnl, tl.pi.copy(), tl.reassign_colsums('exclude'), info['fused']))
        del tl
    assert [r[3] for r in res] == [1, 1, 0]
    for r in res[1:]:
        assert abs(r[0] - res[0][0]) <= 1e-11 * abs(res[0][0])
        assert np.allclose(r[1], res[0][1], rtol=1e-10, atol=0) and np.array_equal(r[2], res[0][2])


def test_config5_per_gpu_shard_against_the_c_oracle(gpu_device):
    """Round 4: the same shard (25M x 50k x ~100, 2.5e9 stored entries — more than 2^31) against oracle/em_fused.c on all host cores
    over the SAME matrix: pi, theta, pi_init and lnl to 1e-9, the per-locus `exclude` counts bit for bit, `conf` and `average` to
    summation order — the last BASELINE configuration that had only properties at full size."""
    from oracle import em_fused as oc
    from telescope_amd._lib import Z_PREV
    n, k = 25_000_000, 50_000
    tl = _synthetic_tl(n, k, 100, 'zipf', uniq=0.02, opts=Opts(max_iter=3, em_epsilon=0.0))
    tl.em()
    ip, ix, rw = tl._eng.export_csr()
    assert len(ix) > 2 ** 31
    ref = oc.em_fused_arrays(ip, ix, rw, k, 0, 200000, 0.0, 3)
    assert ref['n_iter'] == tl.n_iter == 3
    assert abs(tl.lnl - ref['lnl']) <= RTOL * abs(ref['lnl'])
    assert np.allclose(tl.pi, ref['pi'], rtol=RTOL, atol=0) and np.allclose(tl.theta, ref['theta'], rtol=RTOL, atol=0)
    assert np.allclose(tl.pi_init, ref['pi_init'], rtol=RTOL, atol=0)
    pp, tp = tl._eng.get_params(Z_PREV)
    conf, excl, avg = oc.report_sums(ip, ix, rw, k, pp, tp, 0.9, False, max_score=tl.max_score)
    assert np.array_equal(tl.reassign_colsums('exclude'), excl)
    assert np.allclose(tl.reassign_colsums('conf', 0.9), conf, rtol=RTOL, atol=1e-9)
    assert np.allclose(tl.reassign_colsums('average'), avg, rtol=RTOL, atol=1e-9)


# ---------------------------------------------------------------------------------------------------
# config 4 at FULL size against the C oracle
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('fmt', [1, 0])
def test_config4_full_size_against_the_c_oracle(gpu_device, fmt):
    """50M x 30k x ~40 (2e9 stored entries), 3 EM iterations: pi, theta and lnl against oracle/em_fused.c run on
    all host cores over the SAME matrix (exported once), and the per-locus `exclude` counts — computed by the
    oracle from the device's own z parameters — bit for bit."""
    from oracle import em_fused as oc
    from telescope_amd._lib import Z_PREV
    n, k = 50_000_000, 30_000
    tl = _synthetic_tl(n, k, 40, 'zipf', uniq=0.05, options=(('value_format', fmt),), opts=Opts(max_iter=3, em_epsilon=0.0))
    tl.em()
    ip, ix, rw = tl._eng.export_csr()
    assert len(ix) > 1.9e9
    ref = oc.em_fused_arrays(ip, ix, rw, k, 0, 200000, 0.0, 3)
    assert ref['n_iter'] == tl.n_iter == 3
    assert abs(tl.lnl - ref['lnl']) <= RTOL * abs(ref['lnl'])
    assert np.allclose(tl.pi, ref['pi'], rtol=RTOL, atol=0) and np.allclose(tl.theta, ref['theta'], rtol=RTOL, atol=0)
    assert np.allclose(tl.pi_init, ref['pi_init'], rtol=RTOL, atol=0)
    pp, tp = tl._eng.get_params(Z_PREV)
    want = oc.exclude_counts(ip, ix, rw, k, pp, tp, max_score=tl.max_score)
    assert np.array_equal(tl.reassign_colsums('exclude'), want)
    conf, excl, avg = oc.report_sums(ip, ix, rw, k, pp, tp, 0.9, False, max_score=tl.max_score)    # conf / average at full size too
    assert np.array_equal(excl, want)
    assert np.allclose(tl.reassign_colsums('conf', 0.9), conf, rtol=RTOL, atol=1e-9)
    assert np.allclose(tl.reassign_colsums('average'), avg, rtol=RTOL, atol=1e-9)


# ---------------------------------------------------------------------------------------------------
# run-to-run determinism of the integer outputs
# ---------------------------------------------------------------------------------------------------
def test_integer_outputs_identical_across_runs(gpu_device):
    """Column sums are accumulated with unordered LDS atomics, so pi may differ in the last bits from run to
    run; the integer reassign outputs (which hinge on exact ties) must not."""
    c = load_case('mid_zipf_20k')
    raw = case_matrix(c)
    runs = []
    for _ in range(3):
        tl = _tl_for(raw, Opts(c))
        tl.em()
        np.random.seed(1)
        runs.append((tl.n_iter, [tl.reassign_colsums(m) for m in ('exclude', 'choose', 'unique', 'all')], tl.pi.copy(),
                     [tl.reassign_colsums(m) for m in ('average', 'conf')]))
    for r in runs[1:]:
        assert r[0] == runs[0][0]
        for a, b in zip(r[1], runs[0][1]):
            assert np.array_equal(a, b)
        assert np.allclose(r[2], runs[0][2], rtol=1e-12, atol=0)
        for a, b in zip(r[3], runs[0][3]):                       # float-valued modes: unordered fp64 atomics, documented
            assert np.allclose(a, b, rtol=1e-12, atol=1e-9)      # tolerance (profiles/HISTORY.md 5)
    big = []
    for _ in range(2):
        tl = _synthetic_tl(5_000_000, 30000, 40, 'zipf', uniq=0.05, opts=Opts(max_iter=10, em_epsilon=0.0))
        tl.em()
        np.random.seed(1)
        big.append([tl.reassign_colsums(m) for m in ('exclude', 'choose', 'unique', 'all')])
    for a, b in zip(*big):
        assert np.array_equal(a, b)


# ---------------------------------------------------------------------------------------------------
# BASELINE config 3: fp32-vs-fp64 tolerance sweep (tools/precision_sweep.py), at a size the test box runs in seconds
# ---------------------------------------------------------------------------------------------------
def test_precision_sweep_legs(gpu_device):
    """Score codes are EXACT (same fp64 numbers as the fp64 layout); fp32-rounded stored values with fp64 sums stay
    five orders inside north_star's 1e-4 bar; fp32 arithmetic and sums (diagnostic kernel) are finite, close, and
    visibly worse — the measured reason the product accumulates in fp64.  Full-size table: profiles/HISTORY.md 9."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import precision_sweep as ps
    res = ps.sweep(rows=400_000, cols=30_000, nnz_row=40.0, iters=12)
    legs = res['legs']
    assert legs['code16']['lnl_rel'] <= 1e-13 and legs['code16']['pi_max_rel'] <= 1e-11
    assert legs['code16']['final_count_loci_differing'] == 0
    assert legs['store_f32']['lnl_rel'] <= 1e-8 and legs['store_f32']['pi_max_rel'] <= 1e-5
    assert legs['store_f32']['final_count_total_moved'] <= 2
    assert legs['store_bf16m']['pi_max_rel'] > legs['store_f32']['pi_max_rel']
    a = legs['accum_f32']
    assert np.isfinite(a['lnl_rel']) and a['lnl_rel'] < 1e-3 and a['pi_max_rel'] < 0.1
    assert a['pi_max_rel'] > 10 * legs['store_f32']['pi_max_rel']


def test_conflict_aware_row_order_option(gpu_device):
    """Option `deconflict`: the entries of each row are re-ordered (a permutation INSIDE rows, so every result stays
    the same to summation-order noise) so that fewer lanes of a 16-lane group hit accumulator slots that agree
    modulo 16 (the granularity at which ds_add_f64 serialises)."""
    def worst_class_multiplicity(eng):
        info, out = eng.layout_info(), []
        for b in range(2, 12):
            for p in range(info['P']):
                w = eng.debug_subblock(b, p)
                g = (w[:len(w) // 64 * 64] & 0xFFFF).reshape(-1, 16, 4)
                out += [np.bincount(g[x, :, j] & 15, minlength=16).max() for x in range(g.shape[0]) for j in range(4)]
        return float(np.mean(out))
    res = []
    for dc in (0, 1):
        tl = _synthetic_tl(400_000, 30000, 40, 'zipf', uniq=0.05, options=(('value_format', 2), ('deconflict', dc)),
                           opts=Opts(max_iter=6, em_epsilon=0.0))
        info = tl._eng.layout_info()
        assert info['row_order'] == 1 and info['value_bytes'] == 2
        tl.em()
        res.append((tl.lnl, tl.pi.copy(), tl.reassign_colsums('exclude'), worst_class_multiplicity(tl._eng),
                    np.sort(tl._eng.debug_subblock(3, 1))))
    assert abs(res[0][0] - res[1][0]) <= 1e-12 * abs(res[0][0])
    assert np.allclose(res[0][1], res[1][1], rtol=1e-11, atol=0) and np.array_equal(res[0][2], res[1][2])
    assert np.array_equal(res[0][4], res[1][4])             # the same entries, another order
    assert res[1][3] < res[0][3] - 0.5, (res[0][3], res[1][3])


# ---------------------------------------------------------------------------------------------------
# report passes: shortcuts from the setup counts, tie compaction for `choose`
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['bundled', 'tiny_ties', 'tiny_empty_row', 'mid_zipf_20k'])
def test_report_shortcuts_equal_the_row_pass(gpu_device, name):
    """`all` (initial) and `unique` are answered from counts taken at setup (tsem_reassign, option report_shortcuts);
    the generic row pass must give the same numbers — before and after EM, and a stored score of 0 or parameters set
    by the caller must switch the shortcut off."""
    from conftest import case_names
    assert name in case_names(), 'no such golden case: %s (a typo here once skipped the empty-row case silently)' % name
    c = load_case(name)
    raw = case_matrix(c)
    fast, slow = _tl_for(raw, Opts(c)), _tl_for(raw, Opts(c), options=(('report_shortcuts', 0),))
    for tl in (fast, slow):
        tl.em()
    for meth, init in (('all', True), ('unique', False), ('unique', True), ('all', False)):
        a, b = fast.reassign_colsums(meth, 0.9, init), slow.reassign_colsums(meth, 0.9, init)
        assert np.array_equal(a, b), (meth, init)
        assert np.array_equal(a, c['ra_%s_%d_colsum' % (meth, int(init))]), (meth, init)
    # caller-set parameters with an exact zero: the unique rows of that locus drop out of z's pattern (model.py:720)
    pi, theta = fast.pi.copy(), fast.theta.copy()
    j = int(np.argmax(fast.reassign_colsums('unique')))
    pi[j] = 0.0
    for tl in (fast, slow):
        tl._eng.set_params(pi, theta)
    from telescope_amd._lib import Z_CUR
    a, _ = fast._eng.reassign('unique', 0.9, Z_CUR)
    b, _ = slow._eng.reassign('unique', 0.9, Z_CUR)
    assert np.array_equal(a, b) and a[j] == 0


def test_report_shortcut_off_when_a_stored_score_is_zero(gpu_device):
    from oracle.telescope_oracle import OracleModel
    rng = np.random.RandomState(2)
    dense = (rng.rand(300, 40) < 0.15) * rng.randint(1, 200, (300, 40))
    dense[:, 0] = np.where(dense.sum(1) == 0, 7, dense[:, 0])
    raw = sp.csr_matrix(dense.astype(np.uint16))
    raw.data[5] = 0                                        # an explicitly stored zero score (Q = 0)
    raw.data[11] = 0
    tl = _tl_for(raw, Opts(max_iter=5, em_epsilon=0.0))
    tl.em()
    om = OracleModel(raw)
    om.em(0.0, 5)
    for meth, init in (('all', True), ('unique', False), ('all', False)):
        want = np.asarray(om.reassign(meth, 0.9, initial=init).sum(0)).ravel()
        assert np.array_equal(tl.reassign_colsums(meth, 0.9, init), want), (meth, init)


def test_choose_ties_are_compacted_on_the_device(gpu_device):
    """tsem_best_ties returns exactly the rows np.flatnonzero(best_counts > 1) and their counts, in row order, so the
    legacy RNG stream is consumed as before (golden `choose` columns are checked by the reference-parity tests)."""
    from telescope_amd._lib import Z_INITIAL, Z_PREV
    c = load_case('tiny_ties')
    tl = _tl_for(case_matrix(c), Opts(c))
    tl.em()
    for which in (Z_INITIAL, Z_PREV):
        nb = tl._eng.best_counts(which)
        rows, counts = tl._eng.best_ties(which)
        assert np.array_equal(rows, np.flatnonzero(nb > 1)) and np.array_equal(counts, nb[nb > 1])
    assert len(tl._eng.best_ties(Z_INITIAL)[0]) > 0
    from telescope_amd import synthetic
    ip, ix, rw = synthetic.generate(200000, 3000, 6, seed=4, dist='uniform', uniq_frac=0.0)
    rw[:] = 150                                            # every row is one big tie under the initial z
    t2 = _tl_for(sp.csr_matrix((rw, ix, ip), shape=(200000, 3000)), Opts(max_iter=2, em_epsilon=0.0))
    rows, counts = t2._eng.best_ties(Z_INITIAL)            # more than the first buffer of 65536 holds
    nb = t2._eng.best_counts(Z_INITIAL)
    assert len(rows) > 65536 and np.array_equal(rows, np.flatnonzero(nb > 1)) and np.array_equal(counts, nb[rows])


def test_load_scores_validates_on_the_device(gpu_device):
    """tsem_load_scores checks the caller's arrays after the copy, on the device: every violation is TSEM_ERR_ARG with
    the same message the host loop of round 1 gave, and leaves the handle without a matrix."""
    from telescope_amd._lib import Engine, EngineError
    from telescope_amd.likelihood import score_lut
    lut = score_lut(300)
    indptr = np.array([0, 2, 3], np.int64); idx = np.array([0, 2, 1], np.int32); raw = np.array([150, 200, 300], np.uint16)
    eng = Engine(0)
    eng.load_scores(indptr, idx, raw, 3, lut)
    assert eng.dims() == (2, 3, 3)
    for bad_ptr, bad_idx, bad_raw, msg in (
            (np.array([0, 3, 2], np.int64), idx, raw, 'non-decreasing'),
            (np.array([0, 2, 3], np.int64), np.array([0, 3, 1], np.int32), raw, 'column index out of range'),
            (np.array([0, 2, 3], np.int64), np.array([0, -1, 1], np.int32), raw, 'column index out of range'),
            (np.array([0, 2, 3], np.int64), idx, np.array([150, 301, 300], np.uint16), 'exceeds lookup table'),
            (np.array([0, 2, 3], np.int64), np.array([2, 0, 1], np.int32), raw, 'canonical'),
            (np.array([0, 2, 3], np.int64), np.array([2, 2, 1], np.int32), raw, 'canonical')):
        with pytest.raises(EngineError, match=msg):
            eng.load_scores(bad_ptr, bad_idx, bad_raw, 3, lut)
    eng.load_scores(indptr, idx, raw, 3, lut)               # the handle is usable afterwards


def test_constructor_canonicalises_what_the_device_rejects(gpu_device):
    """A matrix with unsorted column ids (or duplicates) is found by the device-side check of tsem_load_scores; the
    constructor then canonicalises it like `csr_matrix.sum_duplicates` and the run equals the canonical matrix's."""
    c = load_case('bundled')
    m = case_matrix(c)
    rng = np.random.default_rng(3)
    data, idx = m.data.copy(), m.indices.copy()
    for i in range(m.shape[0]):                                # shuffle every row's entries
        a, b = m.indptr[i], m.indptr[i + 1]
        p = rng.permutation(b - a)
        data[a:b], idx[a:b] = data[a:b][p], idx[a:b][p]
    shuffled = sp.csr_matrix((data, idx, m.indptr.copy()), shape=m.shape)
    assert not shuffled.has_sorted_indices
    from telescope_amd.likelihood import TelescopeLikelihood
    t1, t2 = TelescopeLikelihood(m, Opts(c)), TelescopeLikelihood(shuffled, Opts(c))
    t1.em(); t2.em()
    assert np.isclose(t1.lnl, t2.lnl, rtol=1e-13, atol=0) and np.allclose(t1.pi, t2.pi, rtol=1e-11, atol=0)   # (LDS atomics: summation order)
    assert np.array_equal(t1.reassign('exclude').sum(0).A1, t2.reassign('exclude').sum(0).A1)


# ---------------------------------------------------------------------------------------------------
# one pass for the report's column sums (tsem_report_colsums / tsem_reassign_rows)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['bundled', 'tiny_ties', 'mid_zipf_20k'])
def test_report_pass_equals_the_single_mode_passes(gpu_device, name):
    """conf | exclude | average from ONE pass, and choose = exclude + the picked entries of the tied rows, equal what
    the one-mode-per-pass kernels give (which the golden tests pin against the reference), for the final and the
    initial z; the tie list equals best_ties."""
    from telescope_amd._lib import Z_INITIAL, Z_PREV
    c = load_case(name)
    tl = _tl_for(case_matrix(c), Opts(c))
    tl.em()
    eng = tl._eng
    for which in (Z_PREV, Z_INITIAL):
        for thresh in (0.9, 0.3):
            sums, rows, counts = eng.report_colsums(which, thresh)
            for m in ('conf', 'exclude', 'average'):
                want, _ = eng.reassign(m, thresh, which)
                if m == 'exclude':
                    assert np.array_equal(sums[m], want)
                else:
                    assert np.allclose(sums[m], want, rtol=1e-12, atol=1e-12)
            r2, c2 = eng.best_ties(which)
            assert np.array_equal(rows, r2) and np.array_equal(counts, c2)
            rng = np.random.default_rng(7)
            picks = (rng.integers(0, 1 << 30, len(rows)) % np.maximum(counts, 1)).astype(np.int32)
            dense = np.zeros(tl.N, np.int32); dense[rows] = picks
            want, _ = eng.reassign('choose', thresh, which, dense)
            got = sums['exclude'] + eng.reassign_rows('choose', thresh, which, rows, picks)
            assert np.array_equal(got, want)
            got2 = sums['exclude'] + eng.reassign_rows('choose', thresh, which, None, picks, n=len(rows))   # rows left on the device
            assert np.array_equal(got2, want)


def test_report_order_consumes_the_rng_like_the_reference(gpu_device):
    """output_report's sequence (model.py:432-457) through the cached report passes gives the same columns as
    uncached single-mode calls with the same RNG seed; em() and `tl.z = ...` drop the cache."""
    c = load_case('mid_zipf_20k')
    raw = case_matrix(c)
    seq = (('conf', False), ('all', True), ('unique', False), ('exclude', True), ('choose', True), ('average', True), ('exclude', False),
           ('choose', False), ('average', False))
    tl = _tl_for(raw, Opts(c)); tl.em()
    np.random.seed(11)
    got = [tl.reassign(m, 0.9, init).sum(0).A1 for m, init in seq]
    assert len(tl._report_cache) == 2                                    # one pass per z
    tl2 = _tl_for(raw, Opts(c)); tl2.em()
    np.random.seed(11)
    for (m, init), g in zip(seq, got):
        which = tl2._which(init)
        sp_picks = tl2._picks(which) if m == 'choose' else None
        want, _ = tl2._eng.reassign(m, 0.9, which, tl2._dense_picks(sp_picks))
        if m in ('conf', 'average'):
            assert np.allclose(g, want, rtol=1e-12, atol=1e-12)
        else:
            assert np.array_equal(g, np.rint(want).astype(np.int64))
    tl.em()
    assert tl._report_cache == {}


def _check_report(tl, whiches, threshes=(0.9,)):
    eng = tl._eng
    for which in whiches:
        for thresh in threshes:
            sums, rows, counts = eng.report_colsums(which, thresh)
            for m in ('conf', 'exclude', 'average'):
                want, _ = eng.reassign(m, thresh, which)
                assert np.allclose(sums[m], want, rtol=1e-12, atol=1e-12), (which, thresh, m)
            r2, c2 = eng.best_ties(which)
            assert np.array_equal(rows, r2) and np.array_equal(counts, c2)
            picks = (np.arange(len(rows)) % np.maximum(counts, 1)).astype(np.int32)
            dense = np.zeros(tl.N, np.int32); dense[rows] = picks
            want, _ = eng.reassign('choose', thresh, which, dense)
            assert np.array_equal(sums['exclude'] + eng.reassign_rows('choose', thresh, which, rows, picks), want)


def test_report_pass_on_the_layouts_the_row_pass_branches_on(gpu_device):
    """RP_REPORT on: rows of hundreds of entries (the sweep path of the row pass), more than eight column parts (the
    two-pass layout), and a caller-assigned z (TSEM_Z_USER)."""
    from telescope_amd._lib import Z_INITIAL, Z_PREV, Z_USER
    from telescope_amd.likelihood import TelescopeLikelihood
    rng = np.random.RandomState(15)
    n, k = 5000, 7000
    lens = np.where(rng.rand(n) < 0.05, rng.randint(100, 600, n), rng.randint(1, 12, n))
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.concatenate([np.sort(rng.choice(k, l, replace=False)) for l in lens]).astype(np.int32)
    data = rng.randint(139, 160, indptr[-1]).astype(np.uint16)            # few score levels: many ties
    raw = sp.csr_matrix((data, indices, indptr), shape=(n, k))
    tl = TelescopeLikelihood(raw, Opts(max_iter=6, em_epsilon=0.0)); tl.em()
    _check_report(tl, (Z_PREV, Z_INITIAL), (0.9, 0.2))
    z = tl.z.copy(); z.data[::7] = 0.0; z.eliminate_zeros()               # a caller's z with a different pattern, used as is
    tl.z = z
    assert tl._which(False) == Z_USER
    _check_report(tl, (Z_USER,))
    np.random.seed(3)
    a = tl.reassign('choose').sum(0).A1
    np.random.seed(3)
    want, _ = tl._eng.reassign('choose', 0.9, Z_USER, tl._dense_picks(tl._picks(Z_USER)))
    assert np.array_equal(a, np.rint(want).astype(np.int64))
    big = _synthetic_tl(300_000, 70_000, 30, 'zipf', uniq=0.05, options=(('split', 0),))   # P = 10 column parts: two-pass layout
    big.em()
    assert big._eng.layout_info()['P'] > 8
    _check_report(big, (Z_PREV, Z_INITIAL))
    big = _synthetic_tl(300_000, 70_000, 30, 'zipf', uniq=0.05)            # round 4: the same matrix on the split layout of the fused kernel
    big.em()
    assert big._eng.layout_info()['split'] == 1
    _check_report(big, (Z_PREV, Z_INITIAL))


def test_torch_transport_fallback(gpu_device):
    """If the library's RCCL communicator cannot be created, the run ends with the library's message (EngineError on every rank)
    unless the caller opted in with TSEM_ALLOW_TORCH_COLLECTIVES=1; then Comm uses torch.distributed collectives on the engine's
    reduce buffer (one host round trip per iteration).  The failure is made on purpose with TSEM_TORCH_COLLECTIVES=1 in a 1-rank
    group of its own process: first the default (raises), then the opt-in — same iteration count, parameters and integer report
    columns as the goldens."""
    import subprocess, sys, os, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import os, sys, warnings
        sys.path[:0] = [%r, %r]
        import numpy as np, scipy.sparse as sp
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29617', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                          TSEM_TORCH_COLLECTIVES='1')
        from conftest import Opts, case_matrix, load_case
        from telescope_amd.distributed import init_from_env
        from telescope_amd.likelihood import TelescopeLikelihood
        from telescope_amd._lib import EngineError
        try:
            init_from_env(force=True)
            raise SystemExit('the default must not fall back by itself')
        except EngineError as e:
            assert 'TSEM_ALLOW_TORCH_COLLECTIVES' in str(e) and 'TSEM_TORCH_COLLECTIVES=1' in str(e), str(e)
        os.environ['TSEM_ALLOW_TORCH_COLLECTIVES'] = '1'
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            comm = init_from_env(force=True)
        assert comm.lib is None and any('torch.distributed' in str(x.message) for x in w)
        for name in ('bundled', 'mid_zipf_20k'):
            c = load_case(name)
            tl = TelescopeLikelihood(case_matrix(c), Opts(c), comm=comm)
            assert not comm.in_library
            tl.em(use_likelihood=bool(c['use_likelihood']))
            assert tl.n_iter == int(c['n_iter'])
            assert abs(tl.lnl - float(c['lnl'])) <= 1e-9 * abs(float(c['lnl']))
            assert np.allclose(tl.pi, c['pi'], rtol=1e-9, atol=0) and np.allclose(tl.pi_init, c['pi_init'], rtol=1e-9, atol=0)
            np.random.seed(int(c['seed']))
            assert np.array_equal(tl.reassign_colsums('choose'), c['ra_choose_0_colsum'])
            assert np.array_equal(tl.reassign_colsums('exclude'), c['ra_exclude_0_colsum'])
        print('FALLBACK-OK')
    """) % (root, os.path.join(root, 'tests'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert 'FALLBACK-OK' in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
