"""`csr_matrix_plus` on the MI355X engine.

Host mirror of /root/reference/telescope/utils/sparse_plus.py:24-174: a
`scipy.sparse.csr_matrix` subclass with `norm`, `scale`, `binmax`, `count`,
`choose_random`, `apply_func`, `check_equal`, `save`, `load` — same names,
argument meaning and `NotImplementedError` behaviour.  The numeric methods run
on the GPU through the C ABI (`tsem_csr_norm_rows`, `tsem_csr_scale`,
`tsem_csr_binmax_rows`); there is no CPU fallback for them.  `count`,
`check_equal`, `save`/`load` are structural, `choose_random` only consumes the
caller's legacy RNG stream and edits the pattern, `apply_func` calls the
caller's Python function per stored element exactly like the reference.
"""
import numpy as np
import scipy.sparse

from . import _lib


class csr_matrix_plus(scipy.sparse.csr_matrix):

    def _f64(self):
        m = scipy.sparse.csr_matrix(self)
        if not m.has_canonical_format:
            m = m.copy()
            m.sum_duplicates()
        return m, np.ascontiguousarray(m.data, dtype=np.float64)

    def _with(self, m, data):
        return type(self)((data, m.indices.copy(), m.indptr.copy()), shape=m.shape)

    def norm(self, axis=None):
        """sparse_plus.py:26-52."""
        m, d = self._f64()
        if axis is None:
            return self._with(m, _lib.csr_scale(0, m.indptr, d, m.shape[1]))
        if axis == 1:
            return self._with(m, _lib.csr_norm_rows(m.indptr, d))
        raise NotImplementedError

    def scale(self, axis=None):
        """sparse_plus.py:69-97."""
        m, d = self._f64()
        if axis is None:
            return self._with(m, _lib.csr_scale(1, m.indptr, d, m.shape[1]))
        if axis == 1:
            return self._with(m, _lib.csr_scale(2, m.indptr, d, m.shape[1]))
        raise NotImplementedError

    def binmax(self, axis=None):
        """sparse_plus.py:99-129."""
        if axis != 1:
            raise NotImplementedError
        m, d = self._f64()
        marks = _lib.csr_binmax_rows(m.indptr, d, m.shape[1])
        ret = type(self)((marks, m.indices.copy(), m.indptr.copy()), shape=m.shape)
        ret.eliminate_zeros()
        return ret

    def count(self, axis=None):
        """sparse_plus.py:131-138."""
        if axis != 1:
            raise NotImplementedError
        return np.array(self.indptr[1:] - self.indptr[:-1], ndmin=2).T

    def choose_random(self, axis=None):
        """sparse_plus.py:140-154 — one legacy-RNG draw per row with more than one stored entry,
        in row order (a vectorised `randint` consumes the stream exactly like the reference's loop)."""
        if axis != 1:
            raise NotImplementedError
        ret = self.copy()
        lens = np.diff(ret.indptr)
        multi = np.nonzero(lens > 1)[0]
        if multi.size:
            picks = ret.indptr[multi] + np.random.randint(0, lens[multi])
            keep = np.ones(ret.nnz, dtype=bool)
            keep[np.repeat(lens > 1, lens)] = False
            keep[picks] = True
            ret.data = np.where(keep, ret.data, 0).astype(ret.data.dtype)
        ret.eliminate_zeros()
        return ret

    def check_equal(self, other):
        """sparse_plus.py:156-159."""
        if self.shape != other.shape:
            return False
        return (self != other).nnz == 0

    def apply_func(self, func):
        """sparse_plus.py:161-165."""
        ret = self.copy()
        ret.data = np.fromiter((func(v) for v in self.data), self.data.dtype, count=len(self.data))
        return ret

    def save(self, filename):
        """sparse_plus.py:167-169."""
        np.savez(filename, data=self.data, indices=self.indices, indptr=self.indptr, shape=self.shape)

    @classmethod
    def load(cls, filename):
        """sparse_plus.py:170-174."""
        z = np.load(filename)
        return cls((z['data'], z['indices'], z['indptr']), shape=z['shape'])
