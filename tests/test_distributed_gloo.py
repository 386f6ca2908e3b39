"""world_size-2 gloo run of the row-sharded EM host path (CPU, no GPU).

Each rank wraps its row shard in the TEST-ONLY oracle engine; everything above
it — shard bounds, the setup all-reduces (global max score, weights, pisum0,
twin signatures), the per-iteration all-reduce of the column sums inside
TelescopeLikelihood.em(), the lnl reduction and the rank-0 RNG draw for
`choose` — is the product code that runs on RCCL at N > 1."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, Opts, case_matrix, load_case


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, name, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    from _oracle_engine import OracleShardEngine
    from telescope_amd.distributed import Comm, shard_bounds
    from telescope_amd.likelihood import TelescopeLikelihood
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        c = load_case(name)
        raw = case_matrix(c).tocsr()
        comm = Comm()
        r0, r1 = shard_bounds(raw.shape[0], world, rank, indptr=raw.indptr)
        eng = OracleShardEngine(raw[r0:r1], raw.shape[1], row_offset=r0)
        tl = TelescopeLikelihood.from_engine(eng, Opts(c), comm=comm)
        tl.em(use_likelihood=bool(c['use_likelihood']))
        out = dict(n_iter=tl.n_iter, lnl=tl.lnl, pi=tl.pi, theta=tl.theta, pi_init=tl.pi_init,
                   max_score=tl.max_score, rows=(r0, r1))
        for meth in ('exclude', 'choose', 'average', 'conf', 'unique', 'all'):
            np.random.seed(int(c['seed']))
            out['ra_' + meth] = tl.reassign_colsums(meth, 0.9, False)
        np.random.seed(int(c['seed']))
        out['ra_choose_init'] = tl.reassign_colsums('choose', 0.9, True)
        np.random.seed(int(c['seed']))
        a = tl.reassign('choose', 0.9)                     # the reference's call pattern: reassign(...).sum(0).A1
        out['ra_assignment_choose'] = a.sum(0).A1
        out['ra_assignment_shape'] = a.shape
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name', ['tiny_twins', 'tiny_ties', 'bundled', 'mid_zipf_20k'])
def test_two_rank_gloo_matches_reference(name):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c = load_case(name)
    a, b = res[0], res[1]
    assert a['rows'][1] == b['rows'][0] and a['rows'][0] == 0
    for r in (a, b):   # every rank holds the same global result
        assert r['n_iter'] == int(c['n_iter'])
        assert r['max_score'] == int(c['max_score'])
        assert abs(r['lnl'] - float(c['lnl'])) <= 1e-10 * abs(float(c['lnl']))
        assert np.allclose(r['pi'], c['pi'], rtol=1e-10, atol=0)
        assert np.allclose(r['theta'], c['theta'], rtol=1e-10, atol=0)
        assert np.allclose(r['pi_init'], c['pi_init'], rtol=1e-12, atol=0)
        for meth in ('exclude', 'choose', 'unique', 'all'):
            assert np.array_equal(r['ra_' + meth], c['ra_%s_0_colsum' % meth]), meth
        for meth in ('average', 'conf'):
            assert np.allclose(r['ra_' + meth], c['ra_%s_0_colsum' % meth], rtol=1e-9, atol=1e-12)
        assert np.array_equal(r['ra_choose_init'], c['ra_choose_1_colsum'])
        assert np.array_equal(r['ra_assignment_choose'], c['ra_choose_0_colsum'])
        assert r['ra_assignment_shape'] == (r['rows'][1] - r['rows'][0], len(c['pi']))
    assert np.array_equal(a['pi'], b['pi'])
