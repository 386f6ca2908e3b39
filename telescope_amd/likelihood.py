"""`TelescopeLikelihood` on the MI355X engine.

Host-side mirror of the reference's operator interface
(/root/reference/telescope/utils/model.py:631-865): same constructor
(`TelescopeLikelihood(score_matrix, opts)`), same methods (`estep`, `mstep`,
`calculate_lnl`, `em`, `reassign`), same attributes (`pi`, `theta`, `pi_init`,
`theta_init`, `z`, `lnl`, `Q`, `Y`, `N`, `K`, `raw_scores`, ...), same log
lines and the same `ValueError` for a bad reassign method — so it drops in for
the scipy.sparse path behind `telescope assign` / `telescope resume`.

All arithmetic runs in libtelescope_em.so (hand-written HIP for gfx950) through
the C ABI in include/telescope_em.h.  There is no CPU fallback: constructing
the class without a usable GPU raises `EngineError`.

Multi-GPU: one process per GPU; pass `comm=` (see telescope_amd/distributed.py).
Each rank constructs the class on ITS row range of the score matrix; the global
maximum score, the setup sums and, every iteration, the per-locus column sums
are all-reduced over RCCL.
"""
import logging as lg

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import Engine, EngineError, Z_CUR, Z_FIRST, Z_INITIAL, Z_PREV, Z_USER  # noqa: F401

REASSIGN_METHODS = ('exclude', 'choose', 'average', 'conf', 'unique', 'all')
_MASK_DTYPE = {'exclude': np.int8, 'choose': np.int8, 'average': np.float64,
               'conf': np.float64, 'unique': np.uint8, 'all': np.uint8}


def score_lut(max_score, scale_factor=100., mantissa_bits=53):
    """Q for every raw score 0..max_score with the reference's numpy expression
    (model.py:653 via sparse_plus.py:89-91: `raw.multiply(1./max).multiply(100.).expm1()`),
    so Q is bit-identical to the reference's.

    `mantissa_bits` < 53 rounds every Q to that many significant bits (24 = what an fp32 value format
    would store, 11 = fp16, 8 = bf16; the exponent range stays fp64's, i.e. a power-of-two scale is
    assumed): the STORAGE-precision leg of BASELINE config 3's tolerance sweep (tools/precision_sweep.py).
    The product default is 53: exact."""
    r = np.arange(int(max_score) + 1, dtype=np.uint16)
    q = np.expm1((r * (1. / max_score)) * scale_factor)
    if mantissa_bits < 53:
        m, e = np.frexp(q)
        q = np.ldexp(np.round(np.ldexp(m, mantissa_bits)), e - mantissa_bits)
    return q


EM_CHUNK = 32     # iterations enqueued per host synchronisation (Engine.em_chunk decides convergence on the device; kernels behind the converging
                  # iteration return at once: ~4 us each).  8 -> 32 in round 4: a synchronisation costs ~0.1-0.2 ms of host time, 2-3 % of a
                  # 20-iteration run of the 0.55 ms 8-GPU shard; a run that stops early pays at most 31 no-op iterations (~0.4 ms)


class _NullComm(object):
    """Single-rank stand-in for distributed.Comm."""
    rank, world = 0, 1
    in_library = True     # nothing to exchange: the engine runs whole chunks of iterations by itself

    def max_scalar(self, v):
        return v

    def sum_array(self, a):
        return a

    def max_array(self, a):
        return a

    def sum_array_u64(self, a):
        return a

    def attach(self, engine, n_cols):
        pass

    def allreduce_device(self, engine, offset=0, count=None):
        pass

    def gather_rows(self, a):
        return [a]

    def scatter_rows(self, parts):
        return parts[0]


class Assignment(object):
    """What `TelescopeLikelihood.reassign` returns: a stand-in for the reference's N x K
    `csr_matrix_plus` that answers `.sum(0)` from the device and turns into the real scipy matrix
    (rows of this rank) when anything else is asked of it."""

    def __init__(self, tl, method, thresh, which, picks):
        self._tl, self._args, self._mat, self._colsum = tl, (method, thresh, which, picks), None, None
        self.shape = (tl.N, tl.K)

    def tocsr(self):
        if self._mat is None:
            self._mat = self._tl._assignment_matrix(*self._args)
        return self._mat

    def sum(self, axis=None, **kw):
        if axis in (0, -2) and not kw and self._mat is None:
            if self._colsum is None:
                self._colsum = self._tl._colsums(*self._args)      # (picks: the tied rows only, sparse)
            return np.asarray(self._colsum).reshape(1, -1).view(np.matrix)   # scipy returns a 1 x K np.matrix: `.A1` works
        return self.tocsr().sum(axis, **kw)

    def __getitem__(self, key):
        return self.tocsr()[key]

    def __getattr__(self, name):                         # data, indices, indptr, multiply, nnz, toarray, ...
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        return getattr(self.tocsr(), name)

    def __array__(self, *a, **kw):
        return self.tocsr().toarray()


class TelescopeLikelihood(object):
    """EM model over a fragments x loci score matrix (model.py:631-865)."""

    def __init__(self, score_matrix, opts, device=None, comm=None, engine=None, engine_options=None):
        """`score_matrix`, `opts` as in the reference (model.py:635-662).  Extra keywords: `device`
        (GPU index), `comm` (row-sharded runs, telescope_amd.distributed), `engine_options` — a dict for
        `tsem_set_option`, e.g. {'value_format': 1} for fp64 entries (include/telescope_em.h)."""
        self.comm = comm if comm is not None else _NullComm()
        # (a csr_matrix is taken as it is: re-wrapping one scans its index arrays, 0.8 s at 2e9 entries)
        raw = score_matrix if isinstance(score_matrix, sp.csr_matrix) else sp.csr_matrix(score_matrix)
        if engine is not None and not raw.has_canonical_format:      # (a fresh engine checks that on the device, below)
            raw = raw.copy()
            raw.sum_duplicates()
        data = raw.data
        if data.dtype != np.uint16 and raw.nnz:                      # (uint16 — what Telescope.raw_scores holds, model.py:300 — needs no check)
            if data.min() < 0 or data.max() > 65535 or \
                    (data.dtype.kind not in 'ui' and not np.all(data == np.floor(data))):
                raise ValueError('score matrix must hold integer alignment scores in [0, 65535]')
        self.raw_scores = score_matrix
        self._raw = raw
        self.N, self.K = raw.shape                                   # model.py:643
        self.scale_factor = 100.                                     # model.py:652
        engine_options = dict(engine_options or {})
        self._mantissa_bits = int(engine_options.pop('lut_mantissa_bits', 53))   # host-side knob, see score_lut
        if device is None:
            device = getattr(self.comm, 'device', 0)
        self._eng = engine if engine is not None else Engine(device)
        for key, value in engine_options.items():
            self._eng.set_option(key, int(value))
        if engine is None:
            # the matrix goes to the device first and the maximum is taken THERE (k_max_u16, 2 ms at 2e9 entries; numpy
            # needs 0.3 s, scipy's .max() 4 s); the score table follows once the global maximum is known
            try:
                self._eng.load_scores(raw.indptr, raw.indices, data.astype(np.uint16, copy=False), self.K, None)
            except EngineError as e:
                if 'canonical' not in str(e):
                    raise
                raw = raw.copy()                                     # unsorted or duplicate column ids (the device found
                raw.sum_duplicates()                                 # them): canonicalise like scipy would, load again
                if raw.nnz and raw.data.dtype != np.uint16 and raw.data.max() > 65535:
                    raise ValueError('score matrix must hold integer alignment scores in [0, 65535]')
                self._raw = raw
                self._eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16, copy=False), self.K, None)
            local_max = self._eng.max_score()
        else:
            local_max = int(data.max()) if raw.nnz else 0
        self.max_score = self.comm.max_scalar(local_max)             # model.py:640 (global)
        self._lut = score_lut(self.max_score, self.scale_factor, self._mantissa_bits) if self.max_score > 0 \
            else np.zeros(1)
        if engine is None:
            self._eng.set_lut(self._lut)

        self.epsilon = opts.em_epsilon                               # model.py:661-662
        self.max_iter = opts.max_iter
        self.pi_prior = opts.pi_prior                                # model.py:686-687
        self.theta_prior = opts.theta_prior
        if getattr(opts, 'use_likelihood', False):                   # (telescope_assign.py:439 passes it to em(): lay the matrix out for it now)
            self._eng.set_option('use_likelihood', 1)

        self._setup_model()

    @classmethod
    def from_engine(cls, engine, opts, comm=None, lut_mantissa_bits=53):
        """Wrap an engine whose score matrix is already resident (device-generated)."""
        self = cls.__new__(cls)
        self._mantissa_bits = int(lut_mantissa_bits)
        self.comm = comm if comm is not None else _NullComm()
        self.raw_scores = self._raw = None
        self._eng = engine
        self.N, self.K, _ = engine.dims()
        self.max_score = self.comm.max_scalar(engine.max_score())   # model.py:640 (global)
        self.scale_factor = 100.
        self._lut = score_lut(self.max_score, mantissa_bits=self._mantissa_bits) if self.max_score > 0 else np.zeros(1)
        engine.set_lut(self._lut)
        self.epsilon, self.max_iter = opts.em_epsilon, opts.max_iter
        self.pi_prior, self.theta_prior = opts.pi_prior, opts.theta_prior
        if getattr(opts, 'use_likelihood', False):
            engine.set_option('use_likelihood', 1)
        self._setup_model()
        return self

    def _setup_model(self):
        self.comm.attach(self._eng, self.K)
        stats, pisum0, cnt, hsh = self._eng.rowstats()               # model.py:679-699 (local)
        sums = self.comm.sum_array(np.concatenate([stats[:2], pisum0]))
        wmax = self.comm.max_array(stats[2:3])
        sig = self.comm.sum_array_u64(np.concatenate([cnt, hsh]))    # wrap-around integer sums
        cnt, hsh = sig[:self.K], sig[self.K:]
        self._total_wt, self._ambig_wt = float(sums[0]), float(sums[1])
        self._max_wt = float(wmax[0])
        self._pisum0 = np.asarray(sums[2:])
        self._pi_prior_wt = self.pi_prior * self._max_wt             # model.py:696-697
        self._theta_prior_wt = self.theta_prior * self._max_wt
        self._eng.set_model(np.array([self._total_wt, self._ambig_wt, self._max_wt]),
                            self._pisum0, cnt, hsh, self.pi_prior, self.theta_prior)
        self.pi = np.repeat(1. / self.K, self.K)                     # model.py:667
        self.theta = np.repeat(1. / self.K, self.K)                  # model.py:673
        self.pi_init = self.theta_init = None
        self.lnl = float('inf')                                      # model.py:683
        self._z = None
        self._z_which = None
        self._report_cache = {}
        self.n_iter, self.converged = 0, False

    # ---- lazily materialised compat attributes --------------------------------
    @property
    def Q(self):
        """model.py:653 — materialised on the host only when asked for."""
        r = self._need_raw()
        return sp.csr_matrix((self._lut[r.data], r.indices, r.indptr), shape=r.shape)

    @property
    def Y(self):
        """model.py:679 — N x 1 uint8 ambiguity indicator, as the device holds it (tsem_export_rowinfo)."""
        return self._eng.row_info()[0].reshape(-1, 1)

    @property
    def _weights(self):
        """model.py:690 — `Q.max(1)`: the N x 1 sparse column of row weights w_i, read back from the device."""
        w = self._eng.row_info()[1]
        return sp.coo_matrix(w.reshape(-1, 1))

    @property
    def z(self):
        """self.z (model.py:795): posteriors of the last E-step, on demand."""
        if self._z is None and self._z_which is not None:
            self._z = self._z_matrix(self._eng.export_z(self._z_which))
        return self._z

    @z.setter
    def z(self, value):
        """A caller-assigned z (the reference's `reassign` reads whatever `self.z` holds, model.py:837)."""
        self._z, self._z_which = value, None
        self._user_z_loaded = False
        self._report_cache = {}

    def _need_raw(self):
        if self._raw is None:
            ip, ix, rw = self._eng.export_csr()
            self._raw = sp.csr_matrix((rw, ix, ip), shape=(self.N, self.K))
        return self._raw

    def _z_matrix(self, zdata):
        """Device z (aligned to Q's pattern, -1 = not in z's pattern) -> CSR."""
        r = self._need_raw()
        keep = zdata >= 0
        if keep.all():
            return sp.csr_matrix((zdata, r.indices.copy(), r.indptr.copy()), shape=r.shape)
        rows = np.repeat(np.arange(self.N), np.diff(r.indptr))[keep]
        indptr = np.zeros(self.N + 1, dtype=r.indptr.dtype)
        np.cumsum(np.bincount(rows, minlength=self.N), out=indptr[1:])
        return sp.csr_matrix((zdata[keep], r.indices[keep], indptr), shape=r.shape)

    def _align(self, z, fill=0.0):
        """Values of sparse z laid out on Q's CSR pattern (`fill` where z has no entry)."""
        r = self._need_raw()
        z = sp.csr_matrix(z)
        if z.shape != r.shape:
            raise ValueError('z has shape %s, expected %s' % (z.shape, r.shape))
        if z.nnz == r.nnz and np.array_equal(z.indptr, r.indptr) and np.array_equal(z.indices, r.indices):
            return np.ascontiguousarray(z.data, dtype=np.float64)
        if not z.has_canonical_format:
            z = z.copy(); z.sum_duplicates()
        kq = np.repeat(np.arange(self.N, dtype=np.int64), np.diff(r.indptr)) * self.K + r.indices
        kz = np.repeat(np.arange(self.N, dtype=np.int64), np.diff(z.indptr)) * self.K + z.indices
        pos = np.searchsorted(kq, kz)
        if np.any(pos >= len(kq)) or np.any(kq[np.minimum(pos, len(kq) - 1)] != kz):
            raise ValueError('z has entries outside the score matrix pattern')
        out = np.full(r.nnz, fill, dtype=np.float64)
        out[pos] = z.data
        return out

    # ---- E / M / lnl on explicit arguments (public API) -----------------------------
    def estep(self, pi, theta):
        """model.py:702-722."""
        lg.debug('started e-step')
        return self._z_matrix(self._eng.estep(pi, theta))

    def mstep(self, z):
        """model.py:724-742."""
        lg.debug('started m-step')
        if self.comm.world > 1 and not getattr(self.comm, 'in_library', False):
            raise NotImplementedError('sharded mstep(z) needs the library communicator (nccl backend)')
        return self._eng.mstep(self._align(z))    # row-sharded: the library sums thetasum over the ranks

    def calculate_lnl(self, z, pi, theta):
        """model.py:744-760."""
        lg.debug('started lnl')
        cur = self._eng.calc_lnl(self._align(z), pi, theta)
        if self.comm.world > 1 and not getattr(self.comm, 'in_library', False):
            cur = float(self.comm.sum_array(np.array([cur]))[0])
        lg.debug('completed lnl')
        return cur

    # ---- EM loop ---------------------------------------------------------------------
    def em(self, use_likelihood=False, loglev=lg.WARNING, save_memory=True, final_lnl=True):
        """model.py:762-806 — same control flow, log lines and final state.  (`final_lnl=False`, not in the reference: skip the
        log-likelihood pass after the loop, model.py:800-801 — bench.py times the EM iterations proper with it.)

        One fused E+M device pass per iteration.  The engine runs CHUNKS of iterations without a host
        round trip (`Engine.em_chunk`): pass, all-reduce of the per-locus column sums over the library's
        RCCL communicator (row-sharded runs), update, and — decided on the device — the convergence
        test; kernels enqueued behind the converging iteration return at once, so the state after the
        call is the reference's.  The log lines of a chunk are emitted when it returns.
        """
        inum, converged, reached_max = 0, False, False
        msgD = 'Iteration {:d}, diff={:.5g}'
        msgL = 'Iteration {:d}, lnl= {:.5e}, diff={:.5g}'
        from time import perf_counter
        eng, comm, K = self._eng, self.comm, self.K
        chunked = getattr(comm, 'in_library', False)          # the engine all-reduces by itself: whole chunks per host round trip
        timeouts = 0
        timing_was = None
        if chunked and not getattr(self, 'keep_kernel_timing', False):
            timing_was = getattr(eng, 'options', {}).get('kernel_timing', 1)
            eng.set_option('kernel_timing', 0)  # per-pass HIP events are for benchmarks (Engine.kernel_stats), not for em()
        if chunked:
            eng.set_prev_lnl(self.lnl)          # model.py:786: the first lnl is compared with what the last run left (inf at first)
            if use_likelihood:
                eng.prepare_likelihood()        # the EM pass of iteration t+1 sums the lnl of iteration t (no lnl pass per iteration)
        owed = None                             # (iteration, diff) whose lnl comes with the next chunk
        while not (converged or reached_max):
            xtime = perf_counter()
            if chunked:
                want = max(1, min(EM_CHUNK, self.max_iter - inum))
                diffs, lnls, converged = eng.em_chunk(want, self.epsilon, use_likelihood, first=(inum == 0),
                                                      last=(inum + want >= self.max_iter))
                if owed is not None:
                    self.lnl = float(eng.lnl_carry)
                    lg.log(loglev, msgL.format(owed[0], self.lnl, owed[1]))
                    owed = None
                for i, diff_est in enumerate(diffs):
                    inum += 1
                    if use_likelihood:
                        if lnls[i] != lnls[i]:                 # NaN: the chunk's last iteration, summed by the next chunk's first pass
                            owed = (inum, diff_est)
                            continue
                        self.lnl = float(lnls[i])
                        lg.log(loglev, msgL.format(inum, self.lnl, diff_est))
                    else:
                        lg.log(loglev, msgD.format(inum, diff_est))
            else:
                # host-driven exchange (gloo / in-process communicators): one iteration per round trip
                eng.em_pass()                       # estep + mstep column sums (local rows)
                comm.allreduce_device(eng, 0, K + 1)   # sum over ranks; slot K carries the time-out flag
                try:
                    diff_est = eng.em_update()      # theta_hat, pi_hat, |pi_hat - pi|_1
                except EngineError as exc:          # a rank's persistent kernel timed out: nobody committed
                    timeouts += 1
                    if exc.code != _lib.ERR_TIMEOUT or timeouts > 3:
                        raise
                    eng.recover_timeout()           # that rank switches to the two-pass kernels
                    continue
                inum += 1
                if inum == 1:
                    self.pi_init, self.theta_init = eng.get_params(Z_CUR)
                if use_likelihood:
                    _lnl = self._device_lnl()
                    diff_lnl = abs(_lnl - self.lnl)
                    lg.log(loglev, msgL.format(inum, _lnl, diff_est))
                    converged = diff_lnl < self.epsilon
                    self.lnl = _lnl
                else:
                    lg.log(loglev, msgD.format(inum, diff_est))
                    converged = diff_est < self.epsilon
            reached_max = inum >= self.max_iter
            lg.debug("time: {}".format(perf_counter() - xtime))
        if timing_was is not None:
            eng.set_option('kernel_timing', timing_was)
        # A fall-back to the two-pass kernels (the persistent kernel could not keep its workgroups co-resident: a GPU shared with another
        # process, a CU mask) costs 3-6 x per iteration from then on: say so where the caller's log goes, not only on the library's stderr
        try:
            fb = int(eng.layout_info().get('fallbacks', 0))
        except Exception:                                   # noqa: BLE001 — (a tests-only engine without layout_info)
            fb = 0
        if fb > getattr(self, '_fallbacks_seen', 0):
            lg.warning('telescope_amd: the fused EM kernel fell back to the two-pass kernels (%d time(s)): this GPU is shared or masked; '
                       'iterations run 3-6x slower from here on' % fb)
            self._fallbacks_seen = fb
        if chunked:
            self.pi_init, self.theta_init = eng.get_params(Z_FIRST)
        self.pi, self.theta = eng.get_params(Z_CUR)
        self._z, self._z_which = None, Z_PREV   # z of the last E-step, exported on demand
        self._report_cache = {}
        _con = 'converged' if converged else 'terminated'
        if not use_likelihood:
            if final_lnl:
                self.lnl = eng.final_lnl() if chunked else self._device_lnl()
            else:
                self.lnl = float('nan')          # not computed (never a stale value: the next em() would seed its lnl test with it)
        self.n_iter, self.converged = inum, converged
        lg.log(loglev, 'EM {:s} after {:d} iterations.'.format(_con, inum))
        if use_likelihood or final_lnl:
            lg.log(loglev, 'Final log-likelihood: {:f}.'.format(self.lnl))
        return

    def _device_lnl(self):
        """calculate_lnl(z(prev params), cur params) without moving z (model.py:785,801)."""
        self._eng.lnl_pass()
        self.comm.allreduce_device(self._eng, self.K, 1)
        return float(self._eng.read_reduce(self.K, 1)[0])

    # ---- reassign -------------------------------------------------------------------------
    def _which(self, initial):
        if initial:
            return Z_INITIAL
        if self._z_which is None:
            if self._z is None:
                raise ValueError('reassign() before em(): no posteriors yet')
            if not getattr(self, '_user_z_loaded', False):   # a caller assigned tl.z: the device reads it as is
                self._eng.set_user_z(self._align(self._z, fill=np.nan))
                self._user_z_loaded = True
            return Z_USER
        return self._z_which

    def _report(self, which, thresh):
        """The column sums `output_report` takes from one z (model.py:432-457) — conf, exclude, average — and the rows
        with several best hits, from ONE device pass, kept until z changes (`em()`, `tl.z = ...`): the reference's
        report asks for three modes of the initial z and two of the final one, one after the other."""
        key = (which, float(thresh))
        cache = self.__dict__.setdefault('_report_cache', {})
        if key not in cache:
            other = [k for k in cache if k[0] == which]
            sums, rows, counts = self._eng.report_colsums(which, thresh)
            self._dev_ties = (which, len(rows))                  # the device keeps this pass's tie rows until the next one
            flat = self.comm.sum_array(np.concatenate([sums['conf'], sums['exclude'], sums['average']]))
            K = self.K
            rep = {'conf': flat[:K] if thresh >= 0 else None, 'exclude': np.rint(flat[K:2 * K]).astype(np.int64), 'average': flat[2 * K:],
                   'rows': rows, 'counts': counts}
            for k in other:                                        # same z, other threshold: only `conf` differs
                del cache[k]
            cache[key] = rep
        return cache[key]

    def _draw_picks(self, counts):
        """Random picks for `choose`, drawn exactly like sparse_plus.py:140-154: one draw per row with >1 best
        hits, in (global) row order, on numpy's legacy global RandomState (seeded by the caller,
        telescope_assign.py:429-431).  `counts`: this rank's tied rows' numbers of best hits, in row order."""
        parts = self.comm.gather_rows(counts)                 # rank order == global row order
        if self.comm.rank == 0:
            allc = parts[0] if len(parts) == 1 else np.concatenate(parts)
            draws = _lib.legacy_randint(allc) if allc.size else np.zeros(0, np.int32)   # == np.random.randint(0, allc), same stream
            cuts = np.cumsum([len(p) for p in parts])[:-1]
            parts = np.split(np.asarray(draws, dtype=np.int32), cuts)
        mine = self.comm.scatter_rows(parts)
        return np.zeros(0, np.int32) if mine is None else np.asarray(mine, dtype=np.int32)

    def _picks(self, which):
        """(rows, picks) of the tied rows for `choose` (sparse); consumes the caller's RNG stream NOW."""
        rep = self._report(which, self._cached_thresh(which, 0.9))   # compacted on the device: only the tied rows travel
        rows, counts = rep['rows'], rep['counts']
        return rows, self._draw_picks(counts)

    def _dense_picks(self, sparse):
        if sparse is None or len(sparse[0]) == 0:
            return None
        picks = np.zeros(self.N, dtype=np.int32)
        picks[sparse[0]] = sparse[1]
        return picks

    def reassign_colsums(self, method, thresh=0.9, initial=False):
        """`reassign(...).sum(0).A1` (model.py:435-457) without materialising the mask."""
        if method not in REASSIGN_METHODS:
            raise ValueError('Argument "method" should be one of (exclude, choose, average, conf, unique, all)')
        which = self._which(initial)
        return self._colsums(method, thresh, which, self._picks(which) if method == 'choose' else None)

    def _colsums(self, method, thresh, which, sparse_picks):
        eng = self._eng
        if method in ('conf', 'exclude', 'average', 'choose') and which != Z_USER:
            if method == 'choose':                               # = exclude + the picked entries of the tied rows
                rep = self._report(which, self._cached_thresh(which, 0.9))
                rows, picks = sparse_picks
                on_dev = getattr(self, '_dev_ties', None) == (which, len(rows)) and rows is rep['rows']
                cs = self.comm.sum_array(eng.reassign_rows('choose', thresh, which, None if on_dev else rows, picks, n=len(rows)))
                return rep['exclude'] + np.rint(cs).astype(np.int64)
            rep = self._report(which, thresh if method == 'conf' else self._cached_thresh(which, thresh))
            return rep[method].copy()
        cs, _ = eng.reassign(method, thresh, which, self._dense_picks(sparse_picks) if method == 'choose' else None)
        cs = self.comm.sum_array(cs)
        if _MASK_DTYPE[method] != np.float64:
            cs = np.rint(cs).astype(np.int64)
        return cs

    def _cached_thresh(self, which, thresh):
        for k in self.__dict__.get('_report_cache', {}):
            if k[0] == which:
                return k[1]                                       # exclude / average do not depend on it: reuse the pass
        # No pass over this z yet, and the caller needs no `conf` column: for the INITIAL z say so (thresh < 0) — its best hits are
        # then found on the score codes alone (k_report_init_codes).  `output_report` asks exclude / choose / average of the initial z
        # and never its conf (model.py:441-446); a later reassign('conf', initial=True) runs the full pass.
        return -1.0 if which == Z_INITIAL else thresh

    def reassign_group_sums(self, method, group_rows, thresh=0.9, initial=False, group_token=None):
        """Per-group column sums of the assignment matrix: row g of the result is
        `reassign(method, thresh, initial)[group_rows[g], :].sum(0).A1` — the per-barcode count matrix
        of `scTelescope.output_report` (model.py:611-625) for `group_rows = barcode_read_indices
        .values()` — computed in one device pass, without materialising the N x K assignment.
        A row listed in several groups (or twice in one) counts once per listing, like fancy
        indexing does.  Row-sharded runs pass the indices local to this rank's rows; the sums are
        all-reduced.  The grouping's layers are cached under a hash of its CONTENT (a caller may refill the
        same list between two reports); hashing tens of millions of listings held in Python lists or sets
        costs more than the device pass, and a report asks up to six methods of one grouping — a caller that
        passes the same hashable `group_token` with the same grouping skips the hash."""
        if method not in REASSIGN_METHODS:
            raise ValueError('Argument "method" should be one of (exclude, choose, average, conf, unique, all)')
        which = self._which(initial)
        picks = self._picks(which) if method == 'choose' else None
        layers, n_groups, digest = self._group_layers(group_rows, group_token)
        out = np.zeros((n_groups, self.K))
        for li, grp in enumerate(layers):
            # the map of a layer goes to the device once and serves every method asked of it (a report asks for up to six); the engine
            # keeps the tag of the map it holds (dropped by anyone else's set_groups): the CONTENT of the grouping, not its identity
            key = (digest, n_groups, li)
            if getattr(self._eng, 'groups_token', None) != key:
                self._eng.set_groups(grp, n_groups)
                self._eng.groups_token = key
            if li == 0:
                self._eng.reassign_groups(method, thresh, which, None, n_groups, self._dense_picks(picks), out=out)
            else:
                out += self._eng.reassign_groups(method, thresh, which, None, n_groups, self._dense_picks(picks))
        out = self.comm.sum_array(out.ravel()).reshape(out.shape)
        if _MASK_DTYPE[method] != np.float64:
            out = np.rint(out).astype(np.int64)
        return out

    def _group_layers(self, group_rows, token=None):
        """Row -> group maps for `reassign_group_sums`, one per LAYER: within a layer every row belongs to at most one group; the
        k-th listing of a row (over all groups, duplicates inside a group included) goes to layer k.  Built once per grouping
        (vectorised: one stable sort of the listings)."""
        import hashlib
        cached = getattr(self, '_group_layer_cache', None)
        if token is not None and cached is not None and cached[0] == ('token', token):
            return cached[1], cached[2], cached[0]              # the caller vouches for the grouping behind the token
        groups = [np.asarray(list(g) if not isinstance(g, np.ndarray) else g, dtype=np.int64) for g in group_rows]
        n_groups = len(groups)
        lens = np.array([len(g) for g in groups], dtype=np.int64)
        rows = np.concatenate(groups) if n_groups and lens.sum() else np.zeros(0, np.int64)
        # keyed on the grouping's CONTENT (a caller may refill the same list object between two reports)
        digest = ('token', token) if token is not None else \
            hashlib.blake2b(lens.tobytes() + np.ascontiguousarray(rows).tobytes(), digest_size=16).hexdigest()
        if cached is not None and cached[0] == digest:
            return cached[1], cached[2], digest
        gid = np.repeat(np.arange(n_groups, dtype=np.int32), lens)
        keep = (rows >= 0) & (rows < self.N)
        rows, gid = rows[keep], gid[keep]
        order = np.argsort(rows, kind='stable')
        rs, gs = rows[order], gid[order]
        first = np.ones(len(rs), bool)
        first[1:] = rs[1:] != rs[:-1]
        start = np.maximum.accumulate(np.where(first, np.arange(len(rs)), 0)) if len(rs) else np.zeros(0, np.int64)
        rank = np.arange(len(rs)) - start                       # 0 for a row's first listing, 1 for its second, ...
        layers = []
        for l in range(int(rank.max()) + 1 if len(rank) else 1):
            grp = np.full(self.N, -1, np.int32)
            sel = rank == l
            grp[rs[sel]] = gs[sel]
            layers.append(grp)
        self._group_layer_cache = (digest, layers, n_groups)
        return layers, n_groups, digest

    def lookup(self, ridx, fidx, method='exclude', thresh=0.9, initial=False, assignment=None):
        """`(tl.z[ridx, fidx], tl.reassign(method, thresh)[ridx, fidx])` for arrays of (fragment, locus) pairs — what
        `Telescope.update_sam` reads per alignment (model.py:483,508-511: `prob = tl.z[ridx, fidx]`, `mat[ridx, fidx] > 0`) — from
        two device passes over the DISTINCT rows asked for, without materialising either N x K matrix (`tl.z` alone is 16 GB of
        values at 2e9 stored entries).  `prob` is the z of the last E-step (0 where the pair is not in z's pattern), like `tl.z`;
        `initial` only selects the z the ASSIGNMENT is made from.  Pass `assignment=tl.reassign('choose', ...)` to look up the
        picks that object drew (a fresh `choose` draws its own, consuming the caller's RNG stream like `reassign`)."""
        if method not in REASSIGN_METHODS:
            raise ValueError('Argument "method" should be one of (exclude, choose, average, conf, unique, all)')
        ridx = np.asarray(ridx, dtype=np.int64).ravel()
        fidx = np.asarray(fidx, dtype=np.int64).ravel()
        if ridx.shape != fidx.shape:
            raise ValueError('ridx and fidx must have the same length')
        if ridx.size and (ridx.min() < 0 or ridx.max() >= self.N or fidx.min() < 0 or fidx.max() >= self.K):
            raise IndexError('index out of range')
        which_a = self._which(initial)
        which_z = self._which(False)
        picks = None
        if method == 'choose':
            picks = assignment._args[3] if assignment is not None else self._picks(which_a)
        r = self._need_raw()
        rows, inv = np.unique(ridx, return_inverse=True)
        lens = (r.indptr[rows + 1] - r.indptr[rows]).astype(np.int64)
        off = np.zeros(len(rows) + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        pk = None
        if picks is not None and len(picks[0]):                  # sparse (tied rows, picks) -> one pick per listed row
            pk = np.zeros(len(rows), np.int32)
            pos = np.searchsorted(picks[0], rows)
            hit = (pos < len(picks[0])) & (np.asarray(picks[0])[np.minimum(pos, len(picks[0]) - 1)] == rows)
            pk[hit] = np.asarray(picks[1])[pos[hit]]
        eng = self._eng
        if which_a == which_z:
            z, m = eng.rows_lookup(method, thresh, which_a, rows, off, pk)
        else:
            z, _ = eng.rows_lookup(method, thresh, which_z, rows, off, None, want_mask=False)
            _, m = eng.rows_lookup(method, thresh, which_a, rows, off, pk, want_z=False)
        # the pair's position inside its row (CSR rows are sorted): a vectorised search over the rows' slices
        key_q = inv.astype(np.int64) * self.K + fidx
        cols = np.concatenate([r.indices[r.indptr[i]:r.indptr[i + 1]] for i in rows]) if len(rows) else np.zeros(0, np.int64)
        key_e = np.repeat(np.arange(len(rows), dtype=np.int64), lens) * self.K + cols
        pos = np.searchsorted(key_e, key_q)
        found = (pos < len(key_e)) & (key_e[np.minimum(pos, max(len(key_e) - 1, 0))] == key_q) if len(key_e) else np.zeros(len(key_q), bool)
        prob = np.zeros(len(ridx))
        val = np.zeros(len(ridx))
        zf = z[pos[found]]
        prob[found] = np.where(zf < 0, 0.0, zf)                   # -1: dropped from z's pattern (an implicit zero of the CSR)
        val[found] = m[pos[found]]
        return prob, val.astype(_MASK_DTYPE[method])

    def reassign(self, method, thresh=0.9, initial=False):
        """model.py:808-865 — the assignment matrix.  Returned as an `Assignment`: `.sum(0)` (all the
        reference's `output_report` asks of it, model.py:435-457) is answered by one device pass over
        the rows; anything else (`[i, j]`, `.data`, `scipy.sparse.csr_matrix(a)`, ...) builds the
        scipy CSR on first use.  `choose` draws its random picks NOW, so the caller's numpy RNG
        stream is consumed exactly where the reference consumes it."""
        if method not in REASSIGN_METHODS:
            raise ValueError('Argument "method" should be one of (exclude, choose, average, conf, unique, all)')
        which = self._which(initial)
        picks = self._picks(which) if method == 'choose' else None
        return Assignment(self, method, thresh, which, picks)

    def _assignment_matrix(self, method, thresh, which, picks):
        _, mask = self._eng.reassign(method, thresh, which, self._dense_picks(picks), want_mask=True)
        r = self._need_raw()
        keep = mask != 0
        rows = np.repeat(np.arange(self.N), np.diff(r.indptr))[keep]
        indptr = np.zeros(self.N + 1, dtype=r.indptr.dtype)
        np.cumsum(np.bincount(rows, minlength=self.N), out=indptr[1:])
        return sp.csr_matrix((mask[keep].astype(_MASK_DTYPE[method]), r.indices[keep], indptr),
                             shape=r.shape)
