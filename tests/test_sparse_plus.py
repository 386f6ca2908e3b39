"""csr_matrix_plus mirror: the reference's own tests (telescope/tests/test_sparse_plus.py:15-66)
restated against the GPU-backed class, its docstring examples that are right
(sparse_plus.py:34-41, 77-85, 107-115), and the structural methods."""
from tempfile import TemporaryFile

import numpy as np
import pytest

from telescope_amd.sparse_plus import csr_matrix_plus

M = [[1, 0, 2], [0, 0, 3], [4, 5, 6]]


def sparse_equal(m1, m2):
    return m1.shape == m2.shape and (m1 != m2).nnz == 0


def test_identity_and_structure():
    m1 = csr_matrix_plus(M)
    assert (m1[0, 0], m1[0, 2], m1[1, 2], m1[2, 0], m1[2, 1], m1[2, 2]) == (1, 2, 3, 4, 5, 6)
    assert np.array_equal(m1.count(1), [[2], [1], [3]])
    assert m1.check_equal(csr_matrix_plus(M)) and not m1.check_equal(csr_matrix_plus([[1, 0, 2], [0, 0, 3], [4, 5, 7]]))
    with pytest.raises(NotImplementedError):
        m1.count(0)


def test_save_load():
    m1 = csr_matrix_plus(M)
    out = TemporaryFile()
    m1.save(out)
    out.seek(0)
    assert sparse_equal(m1, csr_matrix_plus.load(out))


def test_choose_random_consumes_legacy_rng_like_the_reference():
    m = csr_matrix_plus(np.ones((4, 5), dtype=np.int8))
    np.random.seed(3)
    got = m.choose_random(1)
    np.random.seed(3)
    want = [np.random.choice(range(5)) for _ in range(4)]
    assert got.nnz == 4 and list(got.indices) == want


def test_apply_func_keeps_pattern():
    m = csr_matrix_plus(np.array(M, dtype=np.float64))
    r = m.apply_func(lambda x: x if x >= 3 else 0)
    assert r.nnz == m.nnz and np.array_equal(r.toarray(), [[0, 0, 0], [0, 0, 3], [4, 5, 6]])


@pytest.mark.gpu
def test_norm(gpu_device):
    a_none = csr_matrix_plus([[(1. / 21), 0, (2. / 21)], [0, 0, (3. / 21)], [(4. / 21), (5. / 21), (6. / 21)]])
    assert sparse_equal(csr_matrix_plus(M).norm(), a_none)


@pytest.mark.gpu
def test_norm_row(gpu_device):
    got = csr_matrix_plus(M).norm(1).toarray()
    want = np.array([[1 * (1. / 3), 0, 2 * (1. / 3)], [0, 0, 1.], [4 * (1. / 15), 5 * (1. / 15), 6 * (1. / 15)]])
    assert np.allclose(got, want, rtol=1e-15, atol=0)


@pytest.mark.gpu
def test_norm_row_withzero(gpu_device):
    got = csr_matrix_plus([[1, 0, 2], [0, 0, 0], [4, 5, 6]]).norm(1).toarray()
    assert np.array_equal(got[1], [0, 0, 0]) and np.allclose(got[0], [1. / 3, 0, 2. / 3]) \
        and np.allclose(got[2], [4. / 15, 5. / 15, 6. / 15])
    with pytest.raises(NotImplementedError):
        csr_matrix_plus(M).norm(0)


@pytest.mark.gpu
def test_scale_and_binmax(gpu_device):
    m = csr_matrix_plus([[10, 0, 20], [0, 0, 30], [40, 50, 60]])
    assert np.allclose(m.scale().toarray(), np.array([[10, 0, 20], [0, 0, 30], [40, 50, 60]]) / 60.)
    assert np.allclose(m.scale(1).toarray(), [[0.5, 0, 1.], [0, 0, 1.], [40 / 60., 50 / 60., 1.]])
    b = csr_matrix_plus([[6, 0, 2], [0, 0, 3], [4, 5, 6]]).binmax(1)
    assert b.dtype == np.int8 and np.array_equal(b.toarray(), [[1, 0, 0], [0, 0, 1], [0, 0, 1]])
    with pytest.raises(NotImplementedError):
        m.binmax()
