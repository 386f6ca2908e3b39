import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_case(name):
    return dict(np.load(os.path.join(GOLD, 'case_%s.npz' % name), allow_pickle=False))


def case_names(full_only=False):
    names = sorted(f[5:-4] for f in os.listdir(GOLD) if f.startswith('case_'))
    if full_only:
        names = [n for n in names if not n.startswith('mid_')]
    return names


def case_matrix(c):
    """Raw uint16 CSR of a golden case (mid-size cases are regenerated)."""
    import scipy.sparse as sp
    if 'raw_data' in c:
        return sp.csr_matrix((c['raw_data'], c['raw_indices'], c['raw_indptr']),
                             shape=tuple(int(x) for x in c['shape']))
    from telescope_amd import synthetic
    ip, ix, rw = synthetic.generate(int(c['gen_n']), int(c['gen_k']), float(c['gen_d']),
                                    seed=int(c['gen_seed']), dist=str(c['gen_dist']),
                                    uniq_frac=float(c['gen_uniq']))
    chk = int(np.sum(ix.astype(np.uint64) * np.uint64(2654435761) + rw.astype(np.uint64), dtype=np.uint64))
    assert chk == int(c['gen_checksum']), 'synthetic generator changed: regenerate goldens'
    return sp.csr_matrix((rw, ix, ip), shape=(int(c['gen_n']), int(c['gen_k'])))


class Opts(object):
    def __init__(self, c=None, **kw):
        self.em_epsilon, self.max_iter, self.pi_prior, self.theta_prior = 1e-7, 100, 0, 200000
        if c is not None:
            self.em_epsilon = float(c['em_epsilon']); self.max_iter = int(c['max_iter'])
            self.pi_prior = float(c['pi_prior']); self.theta_prior = float(c['theta_prior'])
            if self.pi_prior == int(self.pi_prior): self.pi_prior = int(self.pi_prior)
            if self.theta_prior == int(self.theta_prior): self.theta_prior = int(self.theta_prior)
        self.__dict__.update(kw)


@pytest.fixture(scope='session')
def gpu_device():
    from telescope_amd import _lib
    try:
        e = _lib.Engine(0)
    except _lib.EngineError as exc:
        pytest.fail('GPU test selected but the HIP engine is unusable: %s' % exc)
    e.close()
    return 0
