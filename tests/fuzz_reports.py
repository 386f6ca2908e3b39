"""Random small matrices through em() and EVERY reassign column sum (6 methods x initial / final z, in random order, so that the
report cache, the codes-only pass of the initial z, the tie list on the device and `choose`'s picks meet in every combination)
against the oracle: integer columns bit for bit, conf / average to 1e-9.  A soak (the oracle is the checker, hence its place under tests/); tests/test_gpu_round5.py runs a slice of it:

    python tests/fuzz_reports.py [first_seed=0] [n_seeds=200] [public | sharded | lookups | groups | converge | own]      (`sharded`: 2-3 in-process ranks; `public`: estep / mstep / calculate_lnl with caller-supplied parameters; `own`: the END-TO-END comparison — the oracle's final z from the oracle's OWN parameters — which counts the rows whose masks differ instead of asserting, see `own`)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import scipy.sparse as sp
from oracle.telescope_oracle import OracleModel
from telescope_amd import _lib
from telescope_amd.likelihood import TelescopeLikelihood, score_lut


class Opts(object):
    def __init__(self, **kw):
        self.em_epsilon, self.max_iter, self.pi_prior, self.theta_prior = 1e-7, 100, 0, 200000
        self.__dict__.update(kw)


def make_case(seed):
    """(rng, raw matrix or None when empty, engine options, shape description) of a seed."""
    rng = np.random.RandomState(77000 + seed)
    k = int(rng.choice([3, 17, 200, 5000, 9000, 24000, 33000]))
    n = int(rng.choice([1, 7, 300, 2500, 12000, 30000]))
    max_len = int(min(k, rng.choice([2, 5, 30, 120])))
    uniq = float(rng.choice([0.0, 0.1, 0.6]))
    lens = np.where(rng.rand(n) < uniq, 1, rng.randint(1, max_len + 1, n))
    if rng.rand() < 0.2:
        lens[rng.randint(0, n, max(1, n // 50))] = 0                          # empty rows
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.concatenate([np.sort(rng.choice(k, l, replace=False)) for l in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    lo, hi = [(139, 212), (1, 5), (1, 2), (100, 1500), (60000, 65535), (0, 6)][int(rng.randint(6))]   # (few distinct scores: many ties; (0, 6): stored zeros)
    data = rng.randint(lo, hi + 1, indptr[-1]).astype(np.uint16)
    if indptr[-1] == 0 or data.max() == 0:
        return rng, None, [], None
    raw = sp.csr_matrix((data, indices, indptr), shape=(n, k))
    options = [('value_format', int(rng.randint(0, 3)) if hi <= 1500 else int(rng.randint(0, 2)))]   # (2 = codes forced: needs a table of at most 2048 entries)
    if rng.rand() < 0.3:
        options.append(('block_rows', int(rng.choice([64, 128, 256]))))
    if rng.rand() < 0.3:
        options.append(('drop_csr_indices', 1))
    if rng.rand() < 0.15:
        options.append(('em_kernel', 1))
    return rng, raw, options, (n, k, max_len, uniq, (lo, hi))


def one(seed):
    rng, raw, options, shape = make_case(seed)
    if raw is None:
        return 'skipped (empty)'
    n, k, max_len, uniq, (lo, hi) = shape
    o = Opts(max_iter=int(rng.randint(1, 5)), em_epsilon=0.0)
    o.pi_prior, o.theta_prior = [(0, 200000), (0, 0), (5, 1000)][int(rng.randint(3))]
    eng = _lib.Engine(0)
    for key, v in options:
        eng.set_option(key, v)
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), k, score_lut(int(raw.data.max())))
    try:
        tl = TelescopeLikelihood.from_engine(eng, o)
    except _lib.EngineError as e:
        if 'value_format=codes needs' in str(e):              # (codes forced where the layout cannot have them: a designed, loud refusal)
            return 'skipped (%s)' % e
        raise
    tl._raw = raw
    tl.em()
    om = OracleModel(raw, o.pi_prior, o.theta_prior)
    om.em(0.0, o.max_iter)
    ctx = (seed, n, k, max_len, uniq, (lo, hi), options)
    if not np.isfinite(om.lnl):
        # no ambiguous row and theta_prior = 0: the reference divides 0 by 0 and spreads the NaN over the unique rows through 0 * NaN
        # (model.py:706-714 multiplies Q * Y by pi * theta); the engine keeps unique rows out of theta and stays finite there
        return 'skipped (the reference yields NaN: %s)' % (ctx,)
    assert abs(tl.lnl - om.lnl) <= 1e-9 * max(abs(om.lnl), 1e-300), ('lnl', tl.lnl, om.lnl, ctx)
    assert np.allclose(tl.pi, om.pi, rtol=1e-9, atol=1e-300) and np.allclose(tl.theta, om.theta, rtol=1e-9, atol=1e-300), ('pi / theta', ctx)
    # The masks of the FINAL z hang on exact ties between products of rounded sums: the engine's column sums (atomics) and the
    # reference's (row order) differ in the last bit of ~5 % of the columns, which flips near-ties on matrices made of two scores
    # (seeds 10, 85, 145 of the first run).  What is checked here is the report pass: the oracle's z is therefore formed from the
    # ENGINE's parameters of the last E-step, bit for bit the same inputs on both sides.
    p_prev, t_prev = eng.get_params(_lib.Z_PREV)
    om.z = om.estep(p_prev, t_prev)
    asks = [(m, ini) for m in ('exclude', 'choose', 'average', 'conf', 'unique', 'all') for ini in (False, True)]
    for i in rng.permutation(len(asks)):
        m, ini = asks[i]
        thresh = float(rng.choice([0.9, 0.6, 0.99]))
        np.random.seed(1234 + seed)
        got = tl.reassign_colsums(m, thresh, initial=ini)
        np.random.seed(1234 + seed)
        want = np.asarray(om.reassign(m, thresh, initial=ini, rng=np.random).sum(0)).ravel()
        if m in ('conf', 'average'):
            assert np.allclose(got, want, rtol=1e-9, atol=1e-9), (m, ini, thresh, np.abs(got - want).max(), ctx)
        else:
            assert np.array_equal(np.asarray(got, np.int64), np.rint(want).astype(np.int64)), (m, ini, thresh, int(np.abs(got - want).sum()), ctx)
    eng.close()
    return 'ok %s' % (ctx,)


def own(seed):
    """End to end: the oracle's final z from the ORACLE'S OWN parameters.  The engine's column sums (atomics) and scipy's (a column's
    terms in row order) differ in the last bit of a few per cent of the columns after an iteration, so two z values that tie exactly on one
    side may be an ulp apart on the other: nothing the report pass can repair — it is exact for the parameters it is given (`one`) — and the
    reference's own result there depends on scipy's order of additions.  This leg MEASURES it: rows of the final z whose `exclude` /
    `average` masks differ, per seed.  Returns 'ok own: <rows> rows differ ...' (never fails on a mask; lnl / pi / theta still assert)."""
    rng, raw, options, shape = make_case(seed)
    if raw is None:
        return 'skipped (empty)'
    n, k, max_len, uniq, (lo, hi) = shape
    o = Opts(max_iter=int(rng.randint(1, 5)), em_epsilon=0.0)
    o.pi_prior, o.theta_prior = [(0, 200000), (0, 0), (5, 1000)][int(rng.randint(3))]
    eng = _lib.Engine(0)
    for key, v in options:
        eng.set_option(key, v)
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), k, score_lut(int(raw.data.max())))
    try:
        tl = TelescopeLikelihood.from_engine(eng, o)
    except _lib.EngineError as e:
        if 'value_format=codes needs' in str(e):
            return 'skipped (%s)' % e
        raise
    tl._raw = raw
    tl.em()
    om = OracleModel(raw, o.pi_prior, o.theta_prior)
    om.em(0.0, o.max_iter)
    ctx = (seed, n, k, max_len, uniq, (lo, hi), options)
    if not np.isfinite(om.lnl):
        return 'skipped (the reference yields NaN: %s)' % (ctx,)
    assert abs(tl.lnl - om.lnl) <= 1e-9 * max(abs(om.lnl), 1e-300), ('lnl', tl.lnl, om.lnl, ctx)
    assert np.allclose(tl.pi, om.pi, rtol=1e-9, atol=1e-300) and np.allclose(tl.theta, om.theta, rtol=1e-9, atol=1e-300), ('pi / theta', ctx)
    rows = set()
    for m in ('exclude', 'average'):
        got = sp.csr_matrix(tl.reassign(m)).astype(np.float64)
        want = sp.csr_matrix(om.reassign(m)).astype(np.float64)
        d = (got - want).tocsr()
        d.eliminate_zeros()
        rows.update(np.flatnonzero(np.diff(d.indptr)).tolist())
    near = eng.layout_info()['near_tie_rows']
    eng.close()
    return 'ok own: %d rows differ of %d (engine redid %d near-tie rows in numpy order) %s' % (len(rows), n, near, ctx)


def public(seed):
    """estep / mstep / calculate_lnl (model.py:702-760) with caller-supplied parameters — a share of them exactly 0, which drops
    entries from z's pattern like scipy's CSR arithmetic does — against the oracle: pattern equal, values to 1e-9."""
    rng, raw, options, shape = make_case(seed)
    if raw is None:
        return 'skipped (empty)'
    n, k = raw.shape
    o = Opts(max_iter=1, em_epsilon=0.0)
    o.pi_prior, o.theta_prior = [(0, 200000), (0, 0), (5, 1000)][int(rng.randint(3))]
    eng = _lib.Engine(0)
    for key, v in options:
        eng.set_option(key, v)
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), k, score_lut(int(raw.data.max())))
    try:
        tl = TelescopeLikelihood.from_engine(eng, o)
    except _lib.EngineError as e:
        if 'value_format=codes needs' in str(e):
            return 'skipped (%s)' % e
        raise
    tl._raw = raw
    om = OracleModel(raw, o.pi_prior, o.theta_prior)
    pi, theta = rng.dirichlet(np.full(k, 0.3)), rng.dirichlet(np.full(k, 0.3))
    zf = float(rng.choice([0.0, 0.05, 0.3]))
    pi[rng.rand(k) < zf] = 0.0
    theta[rng.rand(k) < zf] = 0.0
    ctx = (seed, shape, options, zf)
    with np.errstate(all='ignore'):
        zo = sp.csr_matrix(om.estep(pi, theta))
        zg = sp.csr_matrix(tl.estep(pi, theta))
        zo.sort_indices(); zg.sort_indices()
        if not (np.all(np.isfinite(zo.data))):
            return 'skipped (the reference yields NaN / inf: %s)' % (ctx,)
        assert np.array_equal(zo.indptr, zg.indptr) and np.array_equal(zo.indices, zg.indices), ('estep pattern', zo.nnz, zg.nnz, ctx)
        assert np.allclose(zg.data, zo.data, rtol=1e-9, atol=1e-300), ('estep values', ctx)
        po, to = om.mstep(zo)
        pg, tg = tl.mstep(zg)
        if np.all(np.isfinite(po)) and np.all(np.isfinite(to)):
            assert np.allclose(pg, po, rtol=1e-9, atol=1e-300) and np.allclose(tg, to, rtol=1e-9, atol=1e-300), ('mstep', ctx)
            lo_, lg_ = om.calculate_lnl(zo, po, to), tl.calculate_lnl(zg, pg, tg)
            if np.isfinite(lo_):
                assert abs(lg_ - lo_) <= 1e-9 * max(abs(lo_), 1e-300), ('calculate_lnl', lg_, lo_, ctx)
    eng.close()
    return 'ok %s' % (ctx,)


def sharded(seed):
    """The same random matrices ROW-SHARDED over 2 or 3 ranks (host threads on one GPU, the library's in-process transport: the shipped
    tsem_em_chunk protocol, one all-reduce per iteration) — ranks without rows, with unique rows only, with a single row included —
    against the oracle on the whole matrix: pi / theta / lnl to 1e-9 on every rank and bit-identical between the ranks, the integer
    report columns bit for bit (the final z's from the engine's own parameters, as in `one`)."""
    import threading
    from telescope_amd.distributed import ThreadGroup, shard_bounds
    rng, raw, options, shape = make_case(seed)
    if raw is None:
        return 'skipped (empty)'
    n, k = raw.shape
    o = Opts(max_iter=int(rng.randint(1, 5)), em_epsilon=0.0)
    o.pi_prior, o.theta_prior = [(0, 200000), (0, 0), (5, 1000)][int(rng.randint(3))]
    world = int(rng.choice([2, 3]))
    options = [kv for kv in options if kv[0] != 'em_kernel']        # (every rank on the fused kernel: what a row-sharded run uses)
    om = OracleModel(raw, o.pi_prior, o.theta_prior)
    om.em(0.0, o.max_iter)
    ctx = (seed, shape, options, world)
    if not np.isfinite(om.lnl):
        return 'skipped (the reference yields NaN: %s)' % (ctx,)
    cuts = shard_bounds(n, world, indptr=raw.indptr)
    group = ThreadGroup(0, world)
    out, errs = [None] * world, []
    asks = [(m, ini) for m in ('exclude', 'choose', 'average', 'conf', 'unique', 'all') for ini in (False, True)]
    order = rng.permutation(len(asks))

    def rank_main(rank):
        comm = group.comm(rank)
        r0, r1 = cuts[rank], cuts[rank + 1]
        eo = dict(options); eo['row_offset'] = r0
        tl = TelescopeLikelihood(raw[r0:r1], o, device=0, comm=comm, engine_options=eo)
        tl.em()
        res = dict(lnl=tl.lnl, pi=tl.pi.copy(), theta=tl.theta.copy(), prev=tl._eng.get_params(_lib.Z_PREV), cols={})
        for i in order:
            m, ini = asks[i]
            if rank == 0:
                np.random.seed(1234 + seed)
            res['cols'][(m, ini)] = tl.reassign_colsums(m, 0.9, initial=ini)
        comm.close()
        return res

    def work(r):
        try:
            out[r] = rank_main(r)
        except BaseException as e:   # noqa: BLE001
            errs.append((r, e))
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    hung = any(t.is_alive() for t in ts)
    try:
        group.close()
    except Exception:    # noqa: BLE001
        pass
    assert not hung, ('a rank hangs', ctx)
    if errs:
        if 'value_format=codes needs' in str(errs[0][1]):
            return 'skipped (%s)' % errs[0][1]
        raise AssertionError(('rank %d: %r' % errs[0], ctx))
    for r in out:
        assert abs(r['lnl'] - om.lnl) <= 1e-9 * max(abs(om.lnl), 1e-300), ('lnl', r['lnl'], om.lnl, ctx)
        assert np.allclose(r['pi'], om.pi, rtol=1e-9, atol=1e-300) and np.allclose(r['theta'], om.theta, rtol=1e-9, atol=1e-300), ('pi / theta', ctx)
        assert np.array_equal(r['pi'], out[0]['pi']) and np.array_equal(r['theta'], out[0]['theta']) and r['lnl'] == out[0]['lnl'], ('ranks differ', ctx)
    om.z = om.estep(*out[0]['prev'])
    for (m, ini), got in out[0]['cols'].items():
        np.random.seed(1234 + seed)
        want = np.asarray(om.reassign(m, 0.9, initial=ini, rng=np.random).sum(0)).ravel()
        if m in ('conf', 'average'):
            assert np.allclose(got, want, rtol=1e-9, atol=1e-9), (m, ini, ctx)
        else:
            assert np.array_equal(np.asarray(got, np.int64), np.rint(want).astype(np.int64)), (m, ini, int(np.abs(got - want).sum()), ctx)
    return 'ok %s' % (ctx,)


def lookups(seed):
    """`tl.lookup(ridx, fidx, method)` = `(tl.z[ridx, fidx], tl.reassign(method)[ridx, fidx])` (what update_sam reads per alignment,
    model.py:483,508-511) for random pairs — stored entries, pairs outside the pattern, repeated rows — against the oracle's matrices;
    the final z from the engine's parameters, as in `one`."""
    rng, raw, options, shape = make_case(seed)
    if raw is None:
        return 'skipped (empty)'
    n, k = raw.shape
    o = Opts(max_iter=int(rng.randint(1, 5)), em_epsilon=0.0)
    o.pi_prior, o.theta_prior = [(0, 200000), (0, 0), (5, 1000)][int(rng.randint(3))]
    eng = _lib.Engine(0)
    for key, v in options:
        eng.set_option(key, v)
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), k, score_lut(int(raw.data.max())))
    try:
        tl = TelescopeLikelihood.from_engine(eng, o)
    except _lib.EngineError as e:
        if 'value_format=codes needs' in str(e):
            return 'skipped (%s)' % e
        raise
    tl._raw = raw
    tl.em()
    om = OracleModel(raw, o.pi_prior, o.theta_prior)
    om.em(0.0, o.max_iter)
    ctx = (seed, shape, options)
    if not np.isfinite(om.lnl):
        return 'skipped (the reference yields NaN: %s)' % (ctx,)
    om.z = om.estep(*eng.get_params(_lib.Z_PREV))
    q = int(rng.choice([1, 50, 3000]))
    coo = raw.tocoo()
    take = rng.randint(0, coo.nnz, q)
    ridx, fidx = coo.row[take].astype(np.int64), coo.col[take].astype(np.int64)
    miss = rng.rand(q) < 0.2                                  # pairs that are (mostly) not stored
    fidx[miss] = rng.randint(0, k, int(miss.sum()))
    zo = sp.csr_matrix(om.z)
    for method in ('exclude', 'average', 'conf', 'unique', 'all'):
        for ini in (False, True):
            thresh = float(rng.choice([0.9, 0.6]))
            pz, pm = tl.lookup(ridx, fidx, method, thresh, initial=ini)
            mo = sp.csr_matrix(om.reassign(method, thresh, initial=ini)).astype(np.float64)
            wz = np.asarray(zo[ridx, fidx]).ravel()
            wm = np.asarray(mo[ridx, fidx]).ravel()
            assert np.allclose(pz, wz, rtol=1e-9, atol=1e-300), ('z', method, ini, ctx)
            assert np.allclose(np.asarray(pm, np.float64), wm, rtol=1e-9, atol=1e-12), ('mask', method, ini, thresh, ctx)
    eng.close()
    return 'ok %s' % (ctx,)


def groups(seed):
    """`tl.reassign_group_sums(method, group_rows)` — per-barcode column sums (model.py:523-555 sums the mask's rows per cell) — for
    random groupings (rows in no group, in several, listed twice; empty groups; a small group tile) against the oracle's masks."""
    rng, raw, options, shape = make_case(seed)
    if raw is None:
        return 'skipped (empty)'
    n, k = raw.shape
    o = Opts(max_iter=int(rng.randint(1, 4)), em_epsilon=0.0)
    o.pi_prior, o.theta_prior = [(0, 200000), (0, 0), (5, 1000)][int(rng.randint(3))]
    eng = _lib.Engine(0)
    for key, v in options:
        eng.set_option(key, v)
    if rng.rand() < 0.4:
        eng.set_option('group_tile_bytes', int(rng.choice([1 << 16, 1 << 20])))       # (bytes: forces several tiles)
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), k, score_lut(int(raw.data.max())))
    try:
        tl = TelescopeLikelihood.from_engine(eng, o)
    except _lib.EngineError as e:
        if 'value_format=codes needs' in str(e):
            return 'skipped (%s)' % e
        raise
    tl._raw = raw
    tl.em()
    om = OracleModel(raw, o.pi_prior, o.theta_prior)
    om.em(0.0, o.max_iter)
    ctx = (seed, shape, options)
    if not np.isfinite(om.lnl):
        return 'skipped (the reference yields NaN: %s)' % (ctx,)
    om.z = om.estep(*eng.get_params(_lib.Z_PREV))
    ng = int(rng.choice([1, 3, 40]))
    style = int(rng.randint(3))
    if style == 0:                                              # a partition (what barcodes are)
        lab = rng.randint(0, ng, n)
        grp = [np.nonzero(lab == g)[0] for g in range(ng)]
    elif style == 1:                                            # some rows nowhere, some in two groups, some listed twice
        grp = [rng.randint(0, n, rng.randint(0, max(1, n // 2) + 1)) for _ in range(ng)]
    else:                                                       # python lists, an empty group
        grp = [list(map(int, rng.randint(0, n, rng.randint(0, 20)))) for _ in range(ng)] + [[]]
    for method in ('exclude', 'average', 'conf', 'unique', 'all'):
        ini = bool(rng.randint(2))
        thresh = float(rng.choice([0.9, 0.6]))
        got = tl.reassign_group_sums(method, grp, thresh, initial=ini)
        mo = sp.csr_matrix(om.reassign(method, thresh, initial=ini)).astype(np.float64)
        for gi, g in enumerate(grp):
            g = np.asarray(g, np.int64)
            want = np.asarray(mo[g].sum(0)).ravel() if len(g) else np.zeros(k)
            assert np.allclose(got[gi], want, rtol=1e-9, atol=1e-9), (method, ini, thresh, gi, float(np.abs(got[gi] - want).max()), ctx)
    eng.close()
    return 'ok %s' % (ctx,)


def converge(seed):
    """em() run to CONVERGENCE on the random matrices — both tests of the reference (model.py:771-797: sum |pi - pi_prev| < epsilon, or
    under use_likelihood |lnl - lnl_prev| < epsilon), epsilons that stop the loop early, late or never — against the oracle: the same
    iteration count (unless the oracle's own deciding quantity sits within 1e-6 of epsilon: then either side may stop one later),
    pi / theta / lnl to 1e-9.  Covers the device-side stop flag, the lagged test of the carried lnl (MODE 4) and the per-iteration lnl pass."""
    rng, raw, options, shape = make_case(seed)
    if raw is None:
        return 'skipped (empty)'
    n, k = raw.shape
    use_lnl = bool(rng.randint(2))
    o = Opts(max_iter=int(rng.choice([3, 40, 120])), em_epsilon=float(rng.choice([1e-7, 1e-4, 1e-2, 1.0])))
    o.pi_prior, o.theta_prior = [(0, 200000), (5, 1000)][int(rng.randint(2))]
    eng = _lib.Engine(0)
    for key, v in options:
        eng.set_option(key, v)
    if use_lnl and rng.rand() < 0.7:
        eng.set_option('use_likelihood', 1)                  # lay the matrix out for the carried lnl where the geometry allows
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), k, score_lut(int(raw.data.max())))
    try:
        tl = TelescopeLikelihood.from_engine(eng, o)
    except _lib.EngineError as e:
        if 'value_format=codes needs' in str(e):
            return 'skipped (%s)' % e
        raise
    tl._raw = raw
    tl.em(use_likelihood=use_lnl)
    om = OracleModel(raw, o.pi_prior, o.theta_prior)
    trace = om.em(o.em_epsilon, o.max_iter, use_likelihood=use_lnl)
    ctx = (seed, shape, options, use_lnl, o.em_epsilon, o.max_iter, eng.layout_info()['lnl_fused'])
    if not np.isfinite(om.lnl):
        return 'skipped (the reference yields NaN: %s)' % (ctx,)
    n_ref = len(trace)
    if tl.n_iter != n_ref:
        # the deciding quantity of the oracle's last (or the engine's last) iteration within 1e-6 of epsilon?
        def decider(i):
            if i < 1 or i > len(trace):
                return None
            if use_lnl:
                return abs(trace[i - 1][1] - trace[i - 2][1]) if i >= 2 else None
            return trace[i - 1][0]
        near = [d for d in (decider(min(tl.n_iter, n_ref)), decider(max(tl.n_iter, n_ref))) if d is not None]
        close = any(abs(d - o.em_epsilon) <= 1e-6 * max(o.em_epsilon, d) for d in near)
        assert close and abs(tl.n_iter - n_ref) == 1, ('iterations', tl.n_iter, n_ref, near, ctx)
        return 'ok (stopped one apart on a threshold tie) %s' % (ctx,)
    assert abs(tl.lnl - om.lnl) <= 1e-9 * max(abs(om.lnl), 1e-300), ('lnl', tl.lnl, om.lnl, ctx)
    assert np.allclose(tl.pi, om.pi, rtol=1e-9, atol=1e-300) and np.allclose(tl.theta, om.theta, rtol=1e-9, atol=1e-300), ('pi / theta', ctx)
    eng.close()
    return 'ok %s' % (ctx,)


if __name__ == '__main__':
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    fn = {'public': public, 'sharded': sharded, 'lookups': lookups, 'groups': groups, 'converge': converge, 'own': own}.get(sys.argv[3] if len(sys.argv) > 3 else '', one)
    bad = 0
    flip_rows = flip_seeds = ran = 0
    for s in range(first, first + count):
        try:
            r = fn(s)
        except AssertionError as e:
            bad += 1
            r = 'FAILED %s' % (e,)
        if r.startswith('ok own:'):
            ran += 1
            d = int(r.split()[2])
            flip_rows += d
            flip_seeds += d > 0
        print(s, r, flush=True)
    if fn is own:
        print('end to end (oracle own parameters): %d rows in %d of %d cases have a different exclude / average mask' % (flip_rows, flip_seeds, ran))
    print('failures: %d of %d' % (bad, count))
    sys.exit(1 if bad else 0)
