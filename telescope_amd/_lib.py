"""ctypes binding of libtelescope_em.so (C ABI: include/telescope_em.h).

There is NO CPU fallback.  If the shared library is missing, or no HIP device
is usable, every entry point raises — the product path never computes on the
host.
"""
import ctypes as C
import os
import sys
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# TSEM_LIB: kernel experiments (A/B builds of the same source); the product always uses the in-tree build
LIB_PATH = os.environ.get('TSEM_LIB') or os.path.join(HERE, 'libtelescope_em.so')

OK, ERR_ARG, ERR_HIP, ERR_NOMEM, ERR_TIMEOUT = 0, -1, -2, -3, -4
RA_CODE = {'exclude': 0, 'choose': 1, 'average': 2, 'conf': 3, 'unique': 4, 'all': 5}
Z_PREV, Z_CUR, Z_INITIAL, Z_FIRST, Z_USER = 0, 1, 2, 3, 4
EMK_AUTO, EMK_TWOPASS, EMK_FUSED = 0, 1, 2


class EngineError(RuntimeError):
    code = 0


# host / ABI, set-up, EM support, report / row passes, collectives, CSR primitives (telescope_amd/csrc/tsem_internal.h)
LIB_UNITS = ('tsem_host', 'tsem_setup', 'tsem_em', 'tsem_report', 'tsem_comm', 'tsem_csr')
FZ_UNITS = tuple('tsem_fz_p%d' % p for p in range(1, 9))


def build_library(force=False, verbose=False, extra_flags=(), out=None):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU).  Fourteen translation units — the six of
    LIB_UNITS and the eight that instantiate the fused kernel for one team size each (most of the build time) — are compiled
    in parallel and linked into one shared object."""
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(HERE, 'csrc')
    units = [os.path.join(csrc, u + '.hip') for u in LIB_UNITS + FZ_UNITS]
    deps = units + sorted(os.path.join(csrc, h) for h in os.listdir(csrc) if h.endswith('.h')) + [os.path.join(ROOT, 'include', 'telescope_em.h')]
    target = out or LIB_PATH
    objdir = os.path.join(csrc, '_obj' + ('' if out is None else '_' + os.path.basename(out)))
    common = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-ffp-contract=off',
              '-I' + os.path.join(ROOT, 'include'), '-I' + csrc] + list(extra_flags)
    link_stamp = os.path.join(objdir, '_link.flags')        # (a change of the compile flags alone must rebuild too)

    def unit_flags(src):
        # -ffp-contract=off everywhere but in the units of the persistent EM / lnl kernel.  The report, mask and z kernels must round
        # every product Q * (pi theta) before they add (the reference's sequence: the INTEGER outputs hang on exact ties between z
        # values; an FMA-contracted row sum was one ulp off and broke a tie in the sharded soak, tests/fuzz_reports.py seed 10).  The
        # fused kernel's sums feed pi / theta / lnl only (floating-point tolerance, their order of additions differs from scipy's
        # anyway), and its log-table lnl pass with fp64 entries spills 28 registers without contraction (4.6 -> 6.0 ms).
        return [f for f in common if f != '-ffp-contract=off'] if os.path.basename(src).startswith('tsem_fz_p') else common

    def stamp_ok(src):
        st = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + '.o.flags')
        return os.path.exists(st) and open(st).read() == ' '.join(unit_flags(src))
    same_flags = os.path.exists(link_stamp) and open(link_stamp).read() == ' '.join(common) and all(stamp_ok(u) for u in units)
    if (not force and same_flags and os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps)):
        if verbose:
            print('build_library: %s is newer than all %d sources and headers: nothing to compile' % (os.path.relpath(target, ROOT), len(deps)),
                  flush=True)
        return target
    os.makedirs(objdir, exist_ok=True)

    import re

    def includes(path, seen):
        """Quoted #include closure of one source file (the library's own headers)."""
        for name in re.findall(r'^\s*#include\s+"([^"]+)"', open(path).read(), re.M):
            for d in (os.path.dirname(path), csrc, os.path.join(ROOT, 'include')):
                q = os.path.join(d, name)
                if os.path.exists(q) and q not in seen:
                    seen.add(q)
                    includes(q, seen)
        return seen

    def compile_unit(src):
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + '.o')
        stamp = obj + '.flags'
        cu = unit_flags(src)
        flags = ' '.join(cu)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == flags and \
                all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in [src] + sorted(includes(src, set()))):
            if verbose:
                print('up to date: %s' % os.path.relpath(obj, ROOT), flush=True)
            return obj                                   # this unit's object is newer than everything it is made from
        cmd = cu + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        open(stamp, 'w').write(flags)
        compiled.append(os.path.basename(src))
        return obj
    compiled = []
    with ThreadPoolExecutor(max_workers=len(units)) as ex:
        objs = list(ex.map(compile_unit, units))
    link = ['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', target] + objs + ['-ldl', '-lpthread']   # (RCCL is resolved at run time)
    if verbose:
        print(' '.join(link), flush=True)
    subprocess.run(link, check=True)
    with open(link_stamp, 'w') as fh:
        fh.write(' '.join(common))
    if verbose:      # what this call actually did (whether build() compiled anything must be visible)
        print('build_library: compiled %d of %d units (%s), linked %s' % (len(compiled), len(units), ', '.join(compiled) or 'none',
                                                                          os.path.relpath(target, ROOT)), flush=True)
    return target


def sources_fingerprint():
    """sha256 (16 hex digits) over the library's sources — what a measurement kept under profiles/ is stamped with, so that a later
    run can tell whether the kernels it describes are still the ones in the tree (bench.py `roofline.traffic_unit`)."""
    import hashlib
    csrc = os.path.join(HERE, 'csrc')
    names = sorted(f for f in os.listdir(csrc) if f.endswith(('.hip', '.h')))
    hsh = hashlib.sha256()
    for f in names + [os.path.join(ROOT, 'include', 'telescope_em.h')]:
        with open(f if os.path.isabs(f) else os.path.join(csrc, f), 'rb') as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()[:16]


_lib = None


def _p(dtype):
    return np.ctypeslib.ndpointer(dtype=dtype, flags='C_CONTIGUOUS')


def exported_symbols():
    """Names the header declares; tests check the .so exports every one."""
    import re
    hdr = open(os.path.join(ROOT, 'include', 'telescope_em.h')).read()
    return sorted(set(re.findall(r'\b(tsem_[a-z0-9_]+)\s*\(', hdr)))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            'libtelescope_em.so is not built (%s). Run `python -c "import __graft_entry__ as g; '
            'g.build()"`. There is no CPU fallback.' % LIB_PATH)
    # torch ships its own libamdhip64; if it is going to be used in this process (multi-GPU
    # plumbing), it must be loaded BEFORE ours so both share one HIP runtime.
    # A single-process command-line run never touches torch: TSEM_NO_TORCH=1 (set by telescope_amd/cli.py when it is not a rank
    # of a torch.distributed launch) saves the 1.5-2 s of `import torch` there.
    if 'torch' in sys.modules or os.environ.get('TSEM_NO_TORCH', '0') != '1':
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
    L.tsem_create.argtypes = [C.POINTER(vp), C.c_int]
    L.tsem_destroy.argtypes = [vp]
    L.tsem_destroy.restype = None
    L.tsem_last_error.argtypes = [vp]
    L.tsem_last_error.restype = C.c_char_p
    L.tsem_set_stream.argtypes = [vp, vp]
    L.tsem_set_option.argtypes = [vp, C.c_char_p, i64]
    L.tsem_synchronize.argtypes = [vp]
    L.tsem_load_scores.argtypes = [vp, i64, i32, vp, vp, vp, vp, i32]
    L.tsem_generate.argtypes = [vp, i64, i64, i32, vp, i32, u64, i32, dbl]
    L.tsem_max_score.argtypes = [vp, C.POINTER(i32)]
    L.tsem_set_lut.argtypes = [vp, vp, i32]
    L.tsem_score_lut.argtypes = [i32, C.c_double, vp]
    L.tsem_dims.argtypes = [vp, C.POINTER(i64), C.POINTER(i32), C.POINTER(i64)]
    L.tsem_export_csr.argtypes = [vp, vp, vp, vp]
    L.tsem_rowstats.argtypes = [vp, vp, vp, vp, vp]
    L.tsem_export_rowinfo.argtypes = [vp, vp, vp]
    L.tsem_set_model.argtypes = [vp, vp, vp, vp, vp, dbl, dbl]
    L.tsem_set_params.argtypes = [vp, vp, vp]
    L.tsem_get_params.argtypes = [vp, C.c_int, vp, vp]
    L.tsem_reduce_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.tsem_bind_reduce_buffer.argtypes = [vp, vp, i64]
    L.tsem_em_pass.argtypes = [vp]
    L.tsem_em_update.argtypes = [vp, C.POINTER(dbl)]
    L.tsem_lnl_pass.argtypes = [vp]
    L.tsem_read_reduce.argtypes = [vp, vp, i64, i64]
    L.tsem_em_steps.argtypes = [vp, i32, vp]
    L.tsem_em_chunk.argtypes = [vp, i32, dbl, i32, i32, C.POINTER(i32), C.POINTER(i32), vp, vp, C.POINTER(dbl)]
    L.tsem_prepare_likelihood.argtypes = [vp]
    L.tsem_fallback_twopass.argtypes = [vp]
    L.tsem_recover_timeout.argtypes = [vp, C.POINTER(i32)]
    L.tsem_final_lnl.argtypes = [vp, C.POINTER(dbl)]
    L.tsem_comm_unique_id.argtypes = [vp]
    L.tsem_comm_create.argtypes = [C.POINTER(vp), C.c_int, vp, C.c_int, C.c_int]
    L.tsem_comm_destroy.argtypes = [vp]
    L.tsem_comm_destroy.restype = None
    L.tsem_comm_last_error.argtypes = []
    L.tsem_comm_last_error.restype = C.c_char_p
    L.tsem_comm_attach.argtypes = [vp, vp]
    L.tsem_comm_allreduce.argtypes = [vp, i64, i64]
    L.tsem_comm_allreduce_host.argtypes = [vp, vp, i64, C.c_int]
    L.tsem_comm_library_info.argtypes = [C.c_char_p, i32]
    L.tsem_comm_local_group.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    L.tsem_comm_local_group_destroy.argtypes = [vp]
    L.tsem_comm_local_group_destroy.restype = None
    L.tsem_comm_create_local.argtypes = [C.POINTER(vp), vp, C.c_int]
    L.tsem_set_prev_lnl.argtypes = [vp, dbl]
    L.tsem_legacy_randint.argtypes = [vp, C.POINTER(i32), vp, i64, vp]
    L.tsem_em_run.argtypes = [vp, dbl, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(dbl),
                              vp, vp, vp, vp]
    L.tsem_export_z.argtypes = [vp, C.c_int, vp]
    L.tsem_estep.argtypes = [vp, vp, vp, vp]
    L.tsem_set_user_z.argtypes = [vp, vp]
    L.tsem_mstep.argtypes = [vp, vp, vp, vp]
    L.tsem_calc_lnl.argtypes = [vp, vp, vp, vp, C.POINTER(dbl)]
    L.tsem_best_counts.argtypes = [vp, C.c_int, vp]
    L.tsem_best_ties.argtypes = [vp, C.c_int, i64, vp, vp, C.POINTER(i64)]
    L.tsem_report_colsums.argtypes = [vp, C.c_int, C.c_double, vp, C.POINTER(i64)]
    L.tsem_report_ties.argtypes = [vp, i64, vp, vp]
    L.tsem_reassign_rows.argtypes = [vp, C.c_int, C.c_double, C.c_int, vp, vp, i64, vp]
    L.tsem_reassign.argtypes = [vp, C.c_int, dbl, C.c_int, vp, vp, vp]
    L.tsem_rows_lookup.argtypes = [vp, C.c_int, C.c_int, dbl, i64, vp, vp, vp, vp, vp]
    L.tsem_reassign_groups.argtypes = [vp, C.c_int, dbl, C.c_int, vp, vp, C.c_int32, vp]
    L.tsem_set_groups.argtypes = [vp, vp, C.c_int32]
    L.tsem_csr_norm_rows.argtypes = [C.c_int, i64, vp, vp, vp]
    L.tsem_csr_binmax_rows.argtypes = [C.c_int, i64, i32, vp, vp, vp]
    L.tsem_csr_scale.argtypes = [C.c_int, C.c_int, i64, i32, vp, vp, vp]
    L.tsem_kernel_stats.argtypes = [vp, C.c_int, C.POINTER(dbl), C.POINTER(i64), C.POINTER(i64)]
    L.tsem_report_stats.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64), C.POINTER(i64), C.POINTER(C.c_int32)]
    L.tsem_phase_times.argtypes = [vp, C.c_int, vp, C.POINTER(i64)]
    L.tsem_device_memory.argtypes = [vp, C.c_int, C.POINTER(i64), C.POINTER(i64), vp]
    L.tsem_layout_info.argtypes = [vp, vp]
    L.tsem_debug_fused_prof.argtypes = [vp, vp]
    L.tsem_debug_fused_startup.argtypes = [vp, vp]
    L.tsem_debug_log1p.argtypes = [C.c_int, C.c_int32, vp, vp]
    L.tsem_debug_log1p_tab.argtypes = [C.c_int, C.c_int32, vp, vp]
    L.tsem_debug_log1p_of_log.argtypes = [C.c_int, C.c_int32, vp, vp, vp, vp, vp]
    L.tsem_debug_stream_read.argtypes = [C.c_int, i64, i32, C.POINTER(dbl)]
    L.tsem_debug_subblock.argtypes = [vp, C.c_int64, C.c_int32, vp, C.c_int64]
    for name in exported_symbols():
        fn = getattr(L, name)
        if name not in ('tsem_destroy', 'tsem_last_error', 'tsem_comm_destroy', 'tsem_comm_last_error',
                        'tsem_debug_subblock', 'tsem_comm_local_group_destroy'):
            fn.restype = C.c_int
    L.tsem_debug_subblock.restype = C.c_int64
    _lib = L
    return L


def ptr(a):
    """Raw pointer of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(C.c_void_p)


class Engine(object):
    """One handle == one GPU.  Thin, typed wrapper over the C ABI."""

    def __init__(self, device=0):
        L = lib()
        h = C.c_void_p()
        rc = L.tsem_create(C.byref(h), int(device))
        if rc != OK:
            raise EngineError('tsem_create failed (%d): %s — the EM engine needs an MI355X HIP '
                              'device; there is no CPU fallback.'
                              % (rc, L.tsem_last_error(None).decode()))
        self._L, self._h, self.device = L, h, device

    def close(self):
        if getattr(self, '_h', None):
            self._L.tsem_destroy(self._h)
            self._h = None

    __del__ = close

    def _ck(self, rc):
        if rc != OK:
            e = EngineError('libtelescope_em error %d: %s' % (rc, self._L.tsem_last_error(self._h).decode()))
            e.code = rc
            raise e

    # -- plumbing --
    def set_stream(self, stream_handle):
        self._ck(self._L.tsem_set_stream(self._h, C.c_void_p(stream_handle)))

    def set_option(self, key, value):
        self._ck(self._L.tsem_set_option(self._h, key.encode(), int(value)))
        self.__dict__.setdefault('options', {})[key] = int(value)      # (what was set: em() restores "kernel_timing")

    def synchronize(self):
        self._ck(self._L.tsem_synchronize(self._h))

    # -- matrix --
    def load_scores(self, indptr, indices, raw, n_cols, lut):
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        raw = np.ascontiguousarray(raw, dtype=np.uint16)
        if lut is None:                                   # the table follows: max_score() -> set_lut()
            self._ck(self._L.tsem_load_scores(self._h, len(indptr) - 1, int(n_cols), ptr(indptr), ptr(indices),
                                              ptr(raw), None, 0))
            return
        lut = np.ascontiguousarray(lut, dtype=np.float64)
        self._ck(self._L.tsem_load_scores(self._h, len(indptr) - 1, int(n_cols), ptr(indptr), ptr(indices),
                                          ptr(raw), ptr(lut), len(lut)))

    def generate(self, row_begin, row_end, n_cols, len_cdf, seed, dist, uniq_frac):
        len_cdf = np.ascontiguousarray(len_cdf, dtype=np.uint32)
        self._ck(self._L.tsem_generate(self._h, int(row_begin), int(row_end), int(n_cols), ptr(len_cdf),
                                       len(len_cdf), int(seed), int(dist), float(uniq_frac)))

    def max_score(self):
        m = C.c_int32()
        self._ck(self._L.tsem_max_score(self._h, C.byref(m)))
        return m.value

    def set_lut(self, lut):
        lut = np.ascontiguousarray(lut, dtype=np.float64)
        self._ck(self._L.tsem_set_lut(self._h, ptr(lut), len(lut)))

    def dims(self):
        n, k, z = C.c_int64(), C.c_int32(), C.c_int64()
        self._ck(self._L.tsem_dims(self._h, C.byref(n), C.byref(k), C.byref(z)))
        return n.value, k.value, z.value

    def export_csr(self):
        n, k, nnz = self.dims()
        indptr = np.empty(n + 1, np.int64)
        indices = np.empty(nnz, np.int32)
        raw = np.empty(nnz, np.uint16)
        self._ck(self._L.tsem_export_csr(self._h, ptr(indptr), ptr(indices), ptr(raw)))
        return indptr, indices, raw

    # -- model --
    def rowstats(self):
        _, k, _ = self.dims()
        stats, pisum0 = np.zeros(3), np.zeros(k)
        cnt, hsh = np.zeros(k, np.uint64), np.zeros(k, np.uint64)
        self._ck(self._L.tsem_rowstats(self._h, ptr(stats), ptr(pisum0), ptr(cnt), ptr(hsh)))
        return stats, pisum0, cnt, hsh

    def row_info(self):
        """(Y uint8[N], weights f64[N]) of the local rows as the device holds them (model.py:679,690)."""
        n, _, _ = self.dims()
        y, w = np.zeros(n, np.uint8), np.zeros(n)
        self._ck(self._L.tsem_export_rowinfo(self._h, ptr(y), ptr(w)))
        return y, w

    def set_model(self, stats, pisum0, col_count, col_hash, pi_prior, theta_prior):
        stats = np.ascontiguousarray(stats, dtype=np.float64)
        pisum0 = np.ascontiguousarray(pisum0, dtype=np.float64)
        col_count = np.ascontiguousarray(col_count, dtype=np.uint64)
        col_hash = np.ascontiguousarray(col_hash, dtype=np.uint64)
        self._ck(self._L.tsem_set_model(self._h, ptr(stats), ptr(pisum0), ptr(col_count), ptr(col_hash),
                                        float(pi_prior), float(theta_prior)))

    def set_params(self, pi, theta):
        pi = np.ascontiguousarray(pi, dtype=np.float64)
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        self._ck(self._L.tsem_set_params(self._h, ptr(pi), ptr(theta)))

    def get_params(self, which=Z_CUR):
        _, k, _ = self.dims()
        pi, theta = np.empty(k), np.empty(k)
        self._ck(self._L.tsem_get_params(self._h, which, ptr(pi), ptr(theta)))
        return pi, theta

    # -- EM --
    def reduce_buffer(self):
        p, n = C.c_void_p(), C.c_int64()
        self._ck(self._L.tsem_reduce_buffer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bind_reduce_buffer(self, dptr, count):
        self._ck(self._L.tsem_bind_reduce_buffer(self._h, C.c_void_p(dptr), int(count)))

    def em_pass(self):
        self._ck(self._L.tsem_em_pass(self._h))

    def em_update(self, want_diff=True):
        d = C.c_double()
        self._ck(self._L.tsem_em_update(self._h, C.byref(d) if want_diff else None))
        return d.value

    def lnl_pass(self):
        self._ck(self._L.tsem_lnl_pass(self._h))

    def read_reduce(self, offset, count):
        out = np.empty(count)
        self._ck(self._L.tsem_read_reduce(self._h, ptr(out), int(offset), int(count)))
        return out

    def em_steps(self, n, want_diffs=True):
        out = np.empty(n) if want_diffs else None
        self._ck(self._L.tsem_em_steps(self._h, int(n), ptr(out)))
        return out

    def em_chunk(self, n_max, epsilon=0.0, use_likelihood=False, first=False, last=False):
        """Up to `n_max` iterations of the em() loop body enqueued back to back; convergence is decided on
        the device (include/telescope_em.h).  Returns (diffs, lnls, stopped) of the committed iterations.
        With `use_likelihood` on a layout prepared for it (`prepare_likelihood`) the lnl of the chunk's last
        iteration is NaN unless `last` (it comes with the next chunk: `self.lnl_carry`, NaN when nothing was owed)."""
        n_max = int(n_max)
        diffs, lnls = np.empty(max(1, n_max)), np.empty(max(1, n_max))
        done, stopped, carry = C.c_int32(), C.c_int32(), C.c_double(float('nan'))
        self._ck(self._L.tsem_em_chunk(self._h, n_max, float(epsilon), int(bool(use_likelihood)), int(bool(first)) | (2 if last else 0),
                                       C.byref(done), C.byref(stopped), ptr(diffs), ptr(lnls), C.byref(carry)))
        n = done.value
        self.lnl_carry = carry.value
        return diffs[:n], (lnls[:n] if use_likelihood else None), bool(stopped.value)

    def prepare_likelihood(self):
        """em(use_likelihood=True): lay the matrix out so that the EM pass carries the previous iteration's log-likelihood
        (a no-op when it is, or cannot be; include/telescope_em.h)."""
        self._ck(self._L.tsem_prepare_likelihood(self._h))

    def set_prev_lnl(self, lnl):
        """The lnl the next run's first iteration is compared with under use_likelihood (model.py:786)."""
        self._ck(self._L.tsem_set_prev_lnl(self._h, float(lnl)))

    def final_lnl(self):
        out = C.c_double()
        self._ck(self._L.tsem_final_lnl(self._h, C.byref(out)))
        return out.value

    def fallback_twopass(self):
        self._ck(self._L.tsem_fallback_twopass(self._h))

    def recover_timeout(self):
        sw = C.c_int32()
        self._ck(self._L.tsem_recover_timeout(self._h, C.byref(sw)))
        return bool(sw.value)

    def comm_attach(self, comm_handle):
        self._ck(self._L.tsem_comm_attach(self._h, comm_handle))

    def comm_allreduce(self, offset, count):
        self._ck(self._L.tsem_comm_allreduce(self._h, int(offset), int(count)))

    def em_run(self, epsilon, max_iter, use_likelihood):
        _, k, _ = self.dims()
        n_iter, conv, lnl = C.c_int32(), C.c_int32(), C.c_double()
        diffs, lnls = np.full(max_iter, np.nan), np.full(max_iter, np.nan)
        pi0, th0 = np.empty(k), np.empty(k)
        self._ck(self._L.tsem_em_run(self._h, float(epsilon), int(max_iter), int(bool(use_likelihood)),
                                     C.byref(n_iter), C.byref(conv), C.byref(lnl), ptr(diffs), ptr(lnls),
                                     ptr(pi0), ptr(th0)))
        n = n_iter.value
        return dict(n_iter=n, converged=bool(conv.value), lnl=lnl.value, diffs=diffs[:n],
                    lnls=lnls[:n], pi_init=pi0, theta_init=th0)

    # -- results --
    def export_z(self, which=Z_PREV):
        _, _, nnz = self.dims()
        z = np.empty(nnz)
        self._ck(self._L.tsem_export_z(self._h, which, ptr(z)))
        return z

    def set_user_z(self, z_aligned):
        if z_aligned is not None:
            z_aligned = np.ascontiguousarray(z_aligned, dtype=np.float64)
        self._ck(self._L.tsem_set_user_z(self._h, ptr(z_aligned)))

    def estep(self, pi, theta):
        _, _, nnz = self.dims()
        pi = np.ascontiguousarray(pi, dtype=np.float64)
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        z = np.empty(nnz)
        self._ck(self._L.tsem_estep(self._h, ptr(pi), ptr(theta), ptr(z)))
        return z

    def mstep(self, z_aligned):
        _, k, _ = self.dims()
        z_aligned = np.ascontiguousarray(z_aligned, dtype=np.float64)
        pi, theta = np.empty(k), np.empty(k)
        self._ck(self._L.tsem_mstep(self._h, ptr(z_aligned), ptr(pi), ptr(theta)))
        return pi, theta

    def calc_lnl(self, z_aligned, pi, theta):
        z_aligned = np.ascontiguousarray(z_aligned, dtype=np.float64)
        pi = np.ascontiguousarray(pi, dtype=np.float64)
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        out = C.c_double()
        self._ck(self._L.tsem_calc_lnl(self._h, ptr(z_aligned), ptr(pi), ptr(theta), C.byref(out)))
        return out.value

    def best_counts(self, which):
        n, _, _ = self.dims()
        nb = np.empty(n, np.int32)
        self._ck(self._L.tsem_best_counts(self._h, which, ptr(nb)))
        return nb

    def best_ties(self, which):
        """(rows, counts) of the rows with several best hits, in row order (compacted on the device)."""
        n = C.c_int64()
        cap = 1 << 16
        while True:
            rows, counts = np.empty(cap, np.int32), np.empty(cap, np.int32)
            rc = self._L.tsem_best_ties(self._h, which, cap, ptr(rows), ptr(counts), C.byref(n))
            if rc == OK:
                return rows[:n.value].copy(), counts[:n.value].copy()
            if rc == ERR_ARG and n.value > cap:
                cap = int(n.value)
                continue
            self._ck(rc)

    def report_colsums(self, which, thresh):
        """One pass: {'conf', 'exclude', 'average'} column sums of z(`which`) and the tied rows (rows, counts) in row order."""
        _, k, _ = self.dims()
        out = np.empty(3 * k)
        n = C.c_int64()
        self._ck(self._L.tsem_report_colsums(self._h, which, float(thresh), ptr(out), C.byref(n)))
        rows, counts = np.empty(n.value, np.int32), np.empty(n.value, np.int32)
        self._ck(self._L.tsem_report_ties(self._h, n.value, ptr(rows), ptr(counts)))
        return {'conf': out[:k], 'exclude': out[k:2 * k], 'average': out[2 * k:]}, rows, counts

    def reassign_rows(self, method, thresh, which, rows, picks, n=None):
        """Column sums of reassign(method) restricted to a list of rows (rows=None: the tie rows of the last report)."""
        _, k, _ = self.dims()
        cs = np.empty(k)
        if rows is not None:
            rows = np.ascontiguousarray(rows, dtype=np.int32); n = len(rows)
        pk = np.ascontiguousarray(picks, dtype=np.int32) if picks is not None else None
        self._ck(self._L.tsem_reassign_rows(self._h, RA_CODE[method], float(thresh), which,
                                            ptr(rows) if rows is not None else None,
                                            ptr(pk) if pk is not None else None, int(n), ptr(cs)))
        return cs

    def reassign(self, method, thresh, which, picks=None, want_mask=False):
        n, k, nnz = self.dims()
        cs = np.empty(k)
        mask = np.empty(nnz) if want_mask else None
        if picks is not None:
            picks = np.ascontiguousarray(picks, dtype=np.int32)
        self._ck(self._L.tsem_reassign(self._h, RA_CODE[method], float(thresh), which, ptr(picks), ptr(cs),
                                       ptr(mask)))
        return cs, mask

    def rows_lookup(self, method, thresh, which, rows, out_off, picks=None, want_z=True, want_mask=True):
        """(z, mask) of the entries of the listed rows, compact (tsem_rows_lookup): row rows[i] -> [out_off[i], out_off[i + 1])."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        out_off = np.ascontiguousarray(out_off, dtype=np.int64)
        total = int(out_off[-1]) if len(out_off) else 0
        z = np.zeros(total) if want_z else None
        m = np.zeros(total) if want_mask else None
        pk = None if picks is None else np.ascontiguousarray(picks, dtype=np.int32)
        self._ck(self._L.tsem_rows_lookup(self._h, which, RA_CODE[method], float(thresh), len(rows), ptr(rows), ptr(pk), ptr(out_off),
                                          ptr(z), ptr(m)))
        return z, m

    def set_groups(self, group_of_row, n_groups):
        """Row -> group map of the per-group sums, copied to the device once (None drops it)."""
        self.groups_token = None          # whoever sets a map may tag it afterwards (TelescopeLikelihood: a digest of the grouping)
        if group_of_row is None:
            self._ck(self._L.tsem_set_groups(self._h, None, 0))
            return
        n, _, _ = self.dims()
        grp = np.ascontiguousarray(group_of_row, dtype=np.int32)
        if grp.shape != (n,):
            raise ValueError('group_of_row must have one entry per row')
        self._ck(self._L.tsem_set_groups(self._h, ptr(grp), int(n_groups)))

    def reassign_groups(self, method, thresh, which, group_of_row, n_groups, picks=None, out=None):
        """[n_groups][K] sums of the assignment over the rows of each group; `group_of_row` None = the map of `set_groups`."""
        n, k, _ = self.dims()
        grp = None
        if group_of_row is not None:
            self.groups_token = None      # the call replaces the resident map
            grp = np.ascontiguousarray(group_of_row, dtype=np.int32)
            if grp.shape != (n,):
                raise ValueError('group_of_row must have one entry per row')
        if out is None:
            out = np.zeros((int(n_groups), k))
        if picks is not None:
            picks = np.ascontiguousarray(picks, dtype=np.int32)
        self._ck(self._L.tsem_reassign_groups(self._h, RA_CODE[method], float(thresh), which, ptr(picks), ptr(grp),
                                              int(n_groups), ptr(out)))
        return out

    # -- instrumentation --
    def kernel_stats(self, reset=False):
        ms, n, b = C.c_double(), C.c_int64(), C.c_int64()
        self._ck(self._L.tsem_kernel_stats(self._h, int(reset), C.byref(ms), C.byref(n), C.byref(b)))
        return dict(em_ms=ms.value, em_launches=n.value, algo_bytes_per_pass=b.value)

    def report_stats(self):
        """The last report_colsums: dict(kernel_ms (option kernel_timing != 0), algo_bytes, deferred_rows, kernel)."""
        ms, b, d, k = C.c_double(), C.c_int64(), C.c_int64(), C.c_int32()
        self._ck(self._L.tsem_report_stats(self._h, C.byref(ms), C.byref(b), C.byref(d), C.byref(k)))
        return dict(kernel_ms=ms.value, algo_bytes=b.value, deferred_rows=d.value,
                    kernel={0: None, 1: 'k_rowpass', 2: 'k_report_rows', 3: 'k_report_init_codes', 4: 'k_report_pack32'}[k.value])

    def device_memory(self):
        """dict(free, total, resident=dict(csr, csr_indices, ids, layout, rows)) in bytes."""
        f, t = C.c_int64(), C.c_int64()
        r = np.zeros(5, np.int64)
        self._ck(self._L.tsem_device_memory(self._h, 0, C.byref(f), C.byref(t), ptr(r)))
        return dict(free=f.value, total=t.value, resident=dict(zip(('csr', 'csr_indices', 'ids', 'layout', 'rows'), r.tolist())))

    def phase_times(self, reset=False):
        """Option 'phase_timing': mean microseconds per chunked iteration of pass / column reduce / all-reduce / update, the gap
        to the next iteration, and first-to-last mark; None when nothing was timed."""
        ms, n = np.zeros(6), C.c_int64()
        self._ck(self._L.tsem_phase_times(self._h, int(reset), ptr(ms), C.byref(n)))
        if n.value == 0:
            return None
        us = ms * 1e3 / n.value
        return dict(iterations=int(n.value), **{k: float(v) for k, v in zip(('pass', 'colreduce', 'allreduce', 'update', 'gaps', 'iteration'), us)})

    def fused_prof(self):
        out = np.zeros((64, 16), np.uint64)
        self._ck(self._L.tsem_debug_fused_prof(self._h, ptr(out)))
        return out

    def fused_startup(self):
        """Start-up timeline of the last profiled fused launch: [workgroup][entry, counted, zeroed, tables, loop start, loop end, exit, id]
        in 10 ns ticks (include/telescope_em.h)."""
        out = np.zeros((512, 8), np.uint64)
        self._ck(self._L.tsem_debug_fused_startup(self._h, ptr(out)))
        return out

    def debug_subblock(self, block, part, cap=8192):
        out = np.zeros(cap, np.uint32)
        n = self._L.tsem_debug_subblock(self._h, int(block), int(part), ptr(out), cap)
        if n < 0:
            raise EngineError('tsem_debug_subblock failed (%d)' % n)
        return out[:n]

    def layout_info(self):
        info = np.zeros(32, np.int64)
        self._ck(self._L.tsem_layout_info(self._h, ptr(info)))
        return dict(zip(('P', 'Kp', 'R', 'nb', 'N_amb', 'N_uni', 'nnz_amb', 'nnz_pad', 'twin_cols',
                         'G1', 'G2', 'fused', 'slow_path', 'max_subblock', 'value_bytes', 'hot_cols',
                         'lds_bytes', 'row_order', 'geometry', 'fallbacks', 'bin_repeats', 'reproducible', 'exact_single', 'lnl_fused', 'split', 'single_part_rows', 'row_pass_em', 'lnl_tables', 'lnl_linear', 'lnl_mid_entries', 'lnl_mid_limit', 'near_tie_rows'), info.tolist()))


def legacy_randint(counts):
    """`np.random.randint(0, counts)` on numpy's GLOBAL legacy RandomState — the same picks, the same state afterwards
    — by the library's C loop (tsem_legacy_randint): three times faster than numpy's per-element path, which matters
    for the millions of tied rows `choose` draws for."""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    out = np.empty(counts.size, np.int32)
    if counts.size == 0:
        return out
    st = np.random.get_state()
    if st[0] != 'MT19937':
        raise RuntimeError('numpy legacy RandomState is not MT19937')
    key = np.array(st[1], dtype=np.uint32, copy=True)
    pos = C.c_int32(int(st[2]))
    rc = lib().tsem_legacy_randint(ptr(key), C.byref(pos), ptr(counts), counts.size, ptr(out))
    if rc != OK:
        raise ValueError('low >= high')                     # what numpy raises for a count < 1
    # (the cached Gaussian belongs to other distributions' draws: untouched by integer draws, kept as it is)
    np.random.set_state((st[0], key, pos.value) + tuple(st[3:]))
    return out


def comm_library_info():
    """Which RCCL the library resolved at run time: 'rccl <version> (<path>)', or why none is available."""
    buf = C.create_string_buffer(1024)
    lib().tsem_comm_library_info(buf, 1024)
    return buf.value.decode()


class LocalGroup(object):
    """In-process transport (include/telescope_em.h): `world` engines on ONE device, one host thread each."""

    def __init__(self, device, world):
        L = lib()
        g = C.c_void_p()
        rc = L.tsem_comm_local_group(C.byref(g), int(device), int(world))
        if rc != OK:
            raise EngineError('tsem_comm_local_group failed (%d): %s' % (rc, L.tsem_comm_last_error().decode()))
        self._L, self.handle, self.device, self.world = L, g, device, world

    def comm(self, rank):
        return LibComm(self.device, None, rank, self.world, group=self)

    def close(self):
        if getattr(self, 'handle', None):
            self._L.tsem_comm_local_group_destroy(self.handle)
            self.handle = None


class LibComm(object):
    """The library's own communicator.  RCCL (one per process / GPU): `unique_id()` on rank 0, ship the 128
    bytes to the other ranks, then `LibComm(device, id, rank, world)` everywhere (collective).  In-process
    transport: `LocalGroup(device, world).comm(rank)`."""
    ID_BYTES = 128
    _DT = {('f64', 'sum'): 0, ('u64', 'sum'): 1, ('f64', 'max'): 2, ('i64', 'max'): 3}

    @staticmethod
    def unique_id():
        buf = np.zeros(LibComm.ID_BYTES, np.uint8)
        rc = lib().tsem_comm_unique_id(ptr(buf))
        if rc != OK:
            raise EngineError('tsem_comm_unique_id failed (%d): %s' % (rc, lib().tsem_comm_last_error().decode()))
        return buf.tobytes()

    def __init__(self, device, unique_id, rank, world, group=None):
        L = lib()
        h = C.c_void_p()
        if group is not None:
            rc = L.tsem_comm_create_local(C.byref(h), group.handle, int(rank))
            self._group = group                               # keeps the group alive
        else:
            idb = np.frombuffer(unique_id, np.uint8).copy()
            rc = L.tsem_comm_create(C.byref(h), int(device), ptr(idb), int(rank), int(world))
        if rc != OK:
            raise EngineError('tsem_comm_create failed (%d): %s' % (rc, L.tsem_comm_last_error().decode()))
        self._L, self.handle, self.rank, self.world, self.device = L, h, rank, world, device

    def allreduce(self, a, kind='f64', op='sum'):
        dt = {'f64': np.float64, 'u64': np.uint64, 'i64': np.int64}[kind]
        a = np.array(a, dtype=dt, copy=True).ravel()
        rc = self._L.tsem_comm_allreduce_host(self.handle, ptr(a), a.size, self._DT[(kind, op)])
        if rc != OK:
            raise EngineError('tsem_comm_allreduce_host failed (%d): %s' % (rc, self._L.tsem_comm_last_error().decode()))
        return a

    def close(self):
        if getattr(self, 'handle', None):
            self._L.tsem_comm_destroy(self.handle)
            self.handle = None

    __del__ = close


def debug_log1p(x, device=0, table=False):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    rc = (lib().tsem_debug_log1p_tab if table else lib().tsem_debug_log1p)(device, len(x), ptr(x), ptr(y))
    if rc != OK:
        raise EngineError('tsem_debug_log1p failed (%d)' % rc)
    return y


def debug_log1p_of_log(q, c, device=0):
    """log1p(q * c) as the log-table lnl passes form it from log q and log c (tsem_fused.h, fz_log1p_of_log)."""
    q = np.ascontiguousarray(q, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    with np.errstate(divide='ignore'):
        lq, lc = np.log(q), np.log(c)
    y = np.zeros_like(q)
    rc = lib().tsem_debug_log1p_of_log(device, len(q), ptr(lq), ptr(lc), ptr(q), ptr(c), ptr(y))
    if rc != OK:
        raise EngineError('tsem_debug_log1p_of_log failed (%d)' % rc)
    return y


def device_memory(device=0):
    """(free, total) bytes of a device's memory, without an engine."""
    f, t = C.c_int64(), C.c_int64()
    if lib().tsem_device_memory(None, int(device), C.byref(f), C.byref(t), None) != OK:
        raise EngineError('tsem_device_memory failed (no usable HIP device?)')
    return f.value, t.value


def stream_read_gbs(device=0, nbytes=8 << 30, reps=3):
    """GB/s a pure streaming read reaches on this GPU (tsem_debug_stream_read)."""
    g = C.c_double()
    rc = lib().tsem_debug_stream_read(device, int(nbytes), int(reps), C.byref(g))
    if rc:
        raise EngineError('tsem_debug_stream_read failed (%d)' % rc)
    return g.value


def csr_norm_rows(indptr, data, device=0):
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    data = np.ascontiguousarray(data, dtype=np.float64)
    out = np.empty_like(data)
    rc = lib().tsem_csr_norm_rows(device, len(indptr) - 1, ptr(indptr), ptr(data), ptr(out))
    if rc != OK:
        raise EngineError('tsem_csr_norm_rows failed (%d): %s' % (rc, lib().tsem_last_error(None).decode()))
    return out


def csr_binmax_rows(indptr, data, n_cols, device=0):
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    data = np.ascontiguousarray(data, dtype=np.float64)
    out = np.empty(len(data), np.int8)
    rc = lib().tsem_csr_binmax_rows(device, len(indptr) - 1, int(n_cols), ptr(indptr), ptr(data), ptr(out))
    if rc != OK:
        raise EngineError('tsem_csr_binmax_rows failed (%d): %s' % (rc, lib().tsem_last_error(None).decode()))
    return out


def csr_scale(mode, indptr, data, n_cols, device=0):
    """mode 0: norm(); 1: scale(); 2: scale(1)  (sparse_plus.py:46-52, 93-97)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    data = np.ascontiguousarray(data, dtype=np.float64)
    out = np.empty_like(data)
    rc = lib().tsem_csr_scale(device, int(mode), len(indptr) - 1, int(n_cols), ptr(indptr), ptr(data), ptr(out))
    if rc != OK:
        raise EngineError('tsem_csr_scale failed (%d): %s' % (rc, lib().tsem_last_error(None).decode()))
    return out
