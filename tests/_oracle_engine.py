"""TEST-ONLY stand-in for telescope_amd._lib.Engine built on the oracle.

Lets the multi-rank host logic (telescope_amd/distributed.py + the em() loop of
telescope_amd/likelihood.py) run on CPU under the gloo backend: every rank
wraps ITS row shard in this engine; the reduce buffer is a CPU tensor owned by
Comm.  Lives under tests/ — the product never imports it.
"""
import ctypes

import numpy as np
import scipy.sparse as sp

from oracle import telescope_oracle as orc
from telescope_amd import synthetic

Z_PREV, Z_CUR, Z_INITIAL = 0, 1, 2


class OracleShardEngine(object):
    def __init__(self, raw_local, n_cols, row_offset=0):
        self.raw = sp.csr_matrix(raw_local)
        self.K = n_cols
        self.N = self.raw.shape[0]
        self.row_offset = row_offset
        self.lut = None
        self.red = None

    # --- plumbing ---
    def dims(self):
        return self.N, self.K, self.raw.nnz

    def set_stream(self, s):
        pass

    def max_score(self):
        return int(self.raw.data.max()) if self.raw.nnz else 0

    def set_lut(self, lut):
        self.lut = np.asarray(lut)
        om = orc.OracleModel.__new__(orc.OracleModel)
        om.N, om.K = self.N, self.K
        om.Q = sp.csr_matrix((self.lut[self.raw.data], self.raw.indices, self.raw.indptr), shape=self.raw.shape)
        om.Y = (orc.count_rows(om.Q) > 1).astype(np.uint8)
        om.weights = om.Q.max(1)
        self.om = om

    def bind_reduce_buffer(self, ptr, count):
        buf = (ctypes.c_double * count).from_address(ptr)
        self.red = np.ctypeslib.as_array(buf)

    def export_csr(self):
        return self.raw.indptr.astype(np.int64), self.raw.indices.astype(np.int32), self.raw.data.astype(np.uint16)

    # --- model ---
    def row_info(self):
        w = np.asarray(self.om.weights.todense()).ravel() if self.N else np.zeros(0)
        return self.om.Y.ravel().astype(np.uint8), w

    def rowstats(self):
        om = self.om
        w = np.asarray(om.weights.todense()).ravel() if self.N else np.zeros(0)
        Y = om.Y.ravel()
        stats = np.array([w.sum(), (w * Y).sum(), w.max() if self.N else 0.0])
        pisum0 = np.asarray(om.Q.multiply(1 - om.Y).sum(0)).ravel()
        rows = np.repeat(np.arange(self.N, dtype=np.int64), np.diff(self.raw.indptr)) + self.row_offset
        hv = synthetic.hash3(0x7715, rows, self.raw.data.astype(np.uint64))
        cnt = np.bincount(self.raw.indices, minlength=self.K).astype(np.uint64)
        hsh = np.zeros(self.K, np.uint64)
        with np.errstate(over='ignore'):
            np.add.at(hsh, self.raw.indices, hv)
        return stats, pisum0, cnt, hsh

    def set_model(self, stats, pisum0, cnt, hsh, pi_prior, theta_prior):
        om = self.om
        om.total_wt, om.ambig_wt = stats[0], stats[1]
        om.pi_prior_wt, om.theta_prior_wt = pi_prior * stats[2], theta_prior * stats[2]
        self.pisum0 = np.asarray(pisum0)
        self.pi = np.repeat(1. / self.K, self.K)
        self.theta = np.repeat(1. / self.K, self.K)
        self.pi_prev, self.theta_prev = self.pi, self.theta

    def get_params(self, which=Z_CUR):
        return (self.pi_prev, self.theta_prev) if which == Z_PREV else (self.pi, self.theta)

    # --- EM ---
    def em_pass(self):
        om = self.om
        z = om.estep(self.pi, self.theta)
        ts = z.multiply(om.weights).multiply(om.Y).sum(0)
        self.red[:self.K] = np.asarray(ts).ravel()
        self.red[self.K:] = 0

    def em_update(self, want_diff=True):
        om, ts = self.om, self.red[:self.K].copy()
        theta_hat = (ts + om.theta_prior_wt) / (om.ambig_wt + om.theta_prior_wt * self.K)
        pi_hat = ((self.pisum0 + ts) + om.pi_prior_wt) / (om.total_wt + om.pi_prior_wt * self.K)
        diff = np.abs(pi_hat - self.pi).sum()
        self.pi_prev, self.theta_prev = self.pi, self.theta
        self.pi, self.theta = pi_hat, theta_hat
        return diff

    def lnl_pass(self):
        z = self.om.estep(self.pi_prev, self.theta_prev)
        self.red[self.K] = self.om.calculate_lnl(z, self.pi, self.theta)

    def read_reduce(self, offset, count):
        return self.red[offset:offset + count].copy()

    # --- results ---
    def _z(self, which):
        if which == Z_INITIAL:
            return orc.norm(self.om.Q, 1)
        p, t = self.get_params(which)
        return self.om.estep(p, t)

    def best_counts(self, which):
        v = orc.binmax_rows(self._z(which))
        return np.diff(v.indptr).astype(np.int32)

    def report_colsums(self, which, thresh):
        """Engine.report_colsums: conf | exclude | average of one z and the rows with several best hits."""
        self.om.z = self._z(which)
        sums = {m: np.asarray(self.om.reassign(m, thresh, initial=False).sum(0)).ravel().astype(np.float64)
                for m in ('conf', 'exclude', 'average')}
        nbest = self.best_counts(which)
        rows = np.flatnonzero(nbest > 1).astype(np.int32)
        self._ties = rows
        return sums, rows, nbest[rows].astype(np.int32)

    def reassign_rows(self, method, thresh, which, rows, picks, n=None):
        """Engine.reassign_rows for `choose`: the picked best hit of every listed (tied) row."""
        assert method == 'choose'
        rows = self._ties if rows is None else np.asarray(rows)
        v = orc.binmax_rows(self._z(which))
        cs = np.zeros(self.K)
        for r, pk in zip(rows, np.asarray(picks)):
            cs[v.indices[v.indptr[r] + int(pk)]] += 1.0
        return cs

    def reassign(self, method, thresh, which, picks=None, want_mask=False):
        self.om.z = self._z(which)
        if method == 'choose':
            v = orc.binmax_rows(self.om.z)
            keep = np.ones(v.nnz, bool)
            lens = np.diff(v.indptr)
            for i in np.nonzero(lens > 1)[0]:
                keep[v.indptr[i]:v.indptr[i + 1]] = False
                keep[v.indptr[i] + picks[i]] = True
            v.data = np.where(keep, v.data, 0).astype(v.data.dtype)
            v.eliminate_zeros()
            out = v
        else:
            out = self.om.reassign(method, thresh, initial=False)
        return np.asarray(out.sum(0)).ravel().astype(np.float64), None
