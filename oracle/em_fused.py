"""TEST INFRASTRUCTURE — ctypes loader of oracle/em_fused.c (the plain-C, OpenMP restatement of the EM
loop).  Used by tests/ (a second oracle beside telescope_oracle.py) and by bench.py's cpu_baseline leg
(a strong all-cores CPU number beside the reference-equivalent scipy one).  Never imported by telescope_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libem_fused.so')
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.run(['make', '-s', '-C', HERE], check=True)
        L = C.CDLL(LIB)
        vp, dbl = C.c_void_p, C.c_double
        L.oracle_em_fused.restype = C.c_int
        L.oracle_em_fused.argtypes = [C.c_int64, C.c_int32, vp, vp, vp, vp, dbl, dbl, dbl, C.c_int32, C.c_int32,
                                      vp, vp, vp, vp, vp, vp]
        L.oracle_em_fused2.restype = C.c_int
        L.oracle_em_fused2.argtypes = [C.c_int64, C.c_int32, vp, vp, vp, vp, dbl, dbl, dbl, C.c_int32, C.c_int32, C.c_int32,
                                       vp, vp, vp, vp, vp, vp]
        L.oracle_em_fused3.restype = C.c_int
        L.oracle_em_fused3.argtypes = [C.c_int64, C.c_int32, vp, vp, vp, vp, dbl, dbl, dbl, C.c_int32, C.c_int32, C.c_int32,
                                       vp, vp, vp, vp, vp, vp, vp]
        L.oracle_exclude_counts.restype = C.c_int
        L.oracle_exclude_counts.argtypes = [C.c_int64, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
        L.oracle_report_sums.restype = C.c_int
        L.oracle_report_sums.argtypes = [C.c_int64, C.c_int32, vp, vp, vp, vp, vp, vp, C.c_int32, dbl, C.c_int32, vp, vp, vp]
        _lib = L
    return _lib


def score_lut(max_score, scale=100.):
    """Q for every raw score 0..max, with the reference's numpy expression (model.py:653, sparse_plus.py:89-91)."""
    return np.expm1((np.arange(max_score + 1, dtype=np.uint16) * (1. / max_score)) * scale)


def em_fused(raw, pi_prior=0, theta_prior=200000, epsilon=1e-7, max_iter=100, nthreads=0, use_likelihood=False):
    """EM on a scipy CSR of integer raw scores; returns dict(pi, theta, pi_init, lnl, n_iter, converged, diffs)."""
    raw = raw.tocsr()
    return em_fused_arrays(raw.indptr, raw.indices, raw.data, raw.shape[1], pi_prior, theta_prior, epsilon, max_iter, nthreads,
                           use_likelihood)


def exclude_counts(indptr, indices, data, k, pi, theta, max_score=None):
    """reassign('exclude').sum(0) for z = estep(pi, theta) (oracle_exclude_counts in em_fused.c)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    data = np.ascontiguousarray(data, dtype=np.uint16)
    lut = np.ascontiguousarray(score_lut(int(max_score if max_score is not None else data.max())))
    pi = np.ascontiguousarray(pi, dtype=np.float64)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    counts = np.zeros(k, np.int64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().oracle_exclude_counts(len(indptr) - 1, int(k), p(indptr), p(indices), p(data), p(lut), p(pi), p(theta), p(counts))
    return counts


def report_sums(indptr, indices, data, k, pi, theta, thresh=0.9, initial=False, max_score=None, nthreads=0):
    """(conf, exclude, average): the column sums of reassign('conf', thresh) / ('exclude') / ('average') for
    z = estep(pi, theta), or z = Q.norm(1) with initial=True (oracle_report_sums in em_fused.c)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    data = np.ascontiguousarray(data, dtype=np.uint16)
    lut = np.ascontiguousarray(score_lut(int(max_score if max_score is not None else data.max())))
    pi = np.ascontiguousarray(pi, dtype=np.float64)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    conf, excl, avg = np.zeros(k), np.zeros(k, np.int64), np.zeros(k)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().oracle_report_sums(len(indptr) - 1, int(k), p(indptr), p(indices), p(data), p(lut), p(pi), p(theta),
                                  1 if initial else 0, float(thresh), int(nthreads), p(conf), p(excl), p(avg))
    if rc:
        raise MemoryError('oracle_report_sums')
    return conf, excl, avg


def em_fused_arrays(indptr, indices, data, k, pi_prior=0, theta_prior=200000, epsilon=1e-7, max_iter=100, nthreads=0,
                    use_likelihood=False):
    """The same on raw CSR arrays (no scipy object, no copies when the dtypes already match)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    data = np.ascontiguousarray(data, dtype=np.uint16)
    n = len(indptr) - 1
    lut = np.ascontiguousarray(score_lut(int(data.max())) if data.size else np.zeros(1))
    pi, theta, pi_init = np.zeros(k), np.zeros(k), np.zeros(k)
    lnl, conv = C.c_double(), C.c_int32()
    diffs, lnls = np.zeros(max(1, max_iter)), np.zeros(max(1, max_iter))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    it = lib().oracle_em_fused3(n, k, p(indptr), p(indices), p(data), p(lut), float(pi_prior), float(theta_prior),
                                float(epsilon), int(max_iter), 1 if use_likelihood else 0, int(nthreads), p(pi), p(theta),
                                p(pi_init), C.addressof(lnl), C.addressof(conv), p(diffs), p(lnls))
    if it < 0:
        raise MemoryError('oracle_em_fused')
    return dict(pi=pi, theta=theta, pi_init=pi_init, lnl=lnl.value, n_iter=it, converged=bool(conv.value), diffs=diffs[:it],
                lnls=lnls[:it] if use_likelihood else None)
