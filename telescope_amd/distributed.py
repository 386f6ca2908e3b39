"""Row-sharded EM across GPUs: one process per GPU.

Fragments (rows) are independent given (pi, theta), so each rank owns a
contiguous row range of the score matrix.  Exchange steps:

  setup      max  : largest raw score (Q depends on the GLOBAL max, model.py:640,653)
             sum  : total/ambiguous weight, pisum0[K]           (model.py:691-699)
             max  : largest fragment weight                       (model.py:696-697)
             sum  : column signatures (u64 wrap-around)           (twin detection)
  every iter sum  : per-locus column sums thetasum[K] + an error flag
                    -> ONE all-reduce of K+2 doubles (240 KB at K=30k) on RCCL/xGMI
  lnl        sum  : one scalar (+ flag)
  reassign   sum  : K-vector per mode; `choose` gathers best-hit counts to
                    rank 0, which alone consumes the legacy RNG stream.

ONE collective transport on GPUs, and a CPU stand-in for the host logic:

* backend "nccl" (= RCCL on ROCm): the LIBRARY owns its RCCL communicator
  (`tsem_comm_*`, include/telescope_em.h).  torch.distributed only ships the 128-byte
  id from rank 0 to the others and carries the two object collectives of `choose`; the
  per-iteration all-reduce is issued by libtelescope_em.so itself on the engine's
  stream, between the EM pass and the parameter update, with no host round trip
  (`Engine.em_chunk`), and the setup / reassign sums go through the same communicator.
  If that communicator cannot be created the run ends with the library's message; torch.distributed collectives on the
  engine's reduce buffer (one host round trip per iteration) are an explicit opt-in, TSEM_ALLOW_TORCH_COLLECTIVES=1
  (`decide_transport`).
* backend "gloo": the same host logic on CPU tensors, with a tests-only engine
  standing in for the device (tests/test_distributed_gloo.py).  The reduce buffer is
  then a CPU tensor that torch all-reduces in place between `em_pass` and `em_update`.
"""
import os

import numpy as np


def shard_bounds(n_rows, world, rank=None, indptr=None):
    """Contiguous row ranges, balanced by nnz when `indptr` is given."""
    if indptr is None:
        cuts = [(n_rows * r) // world for r in range(world + 1)]
    else:
        nnz = int(indptr[-1])
        targets = [(nnz * r) // world for r in range(1, world)]
        cuts = [0] + [int(np.searchsorted(indptr, t, side='left')) for t in targets] + [n_rows]
        cuts = [min(max(c, 0), n_rows) for c in cuts]
        for i in range(1, len(cuts)):
            cuts[i] = max(cuts[i], cuts[i - 1])
    if rank is None:
        return cuts
    return cuts[rank], cuts[rank + 1]


def _detach(engine, attached):
    """Take a communicator that is about to be destroyed off the engine it was attached to (tsem_comm_attach(h, NULL))."""
    if engine is not None and attached:
        try:
            engine.comm_attach(None)
        except Exception:                                     # noqa: BLE001 — (the engine may be closed already)
            pass


def decide_transport(lib_failed, lib_error, environ=None):
    """What a GPU rank does once every rank knows whether the library's own RCCL communicator came up everywhere: 'library', or —
    ONLY when the caller opted in with TSEM_ALLOW_TORCH_COLLECTIVES=1 — 'torch' (torch.distributed all-reduces on the engine's reduce
    buffer, one host round trip per EM iteration: the same numbers, 3-10 x slower at 8 ranks).  Without the opt-in a failure raises
    EngineError with the library's message on every rank: the engine has ONE collective transport, and a run that silently took another
    one would print a throughput that is not this library's."""
    environ = os.environ if environ is None else environ
    if not lib_failed:
        return 'library'
    if environ.get('TSEM_ALLOW_TORCH_COLLECTIVES', '0') == '1':
        return 'torch'
    from ._lib import EngineError
    raise EngineError('the in-library RCCL communicator could not be created on every rank (%s).  Set TSEM_ALLOW_TORCH_COLLECTIVES=1 to '
                      'run the collectives through torch.distributed instead (one host round trip per EM iteration).' % lib_error)


class Comm(object):
    """An initialised torch.distributed process group + (on GPUs) the library's RCCL communicator."""

    def __init__(self, device=None, group=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self._torch, self._dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.lib = None                       # _lib.LibComm: the library's own communicator
        self.in_library = False               # the attached engine all-reduces by itself (Engine.em_chunk)
        if self.backend == 'nccl':
            self.device = int(os.environ.get('LOCAL_RANK', 0)) if device is None else device
            self.tdev = torch.device('cuda', self.device)
            torch.cuda.set_device(self.device)
            from ._lib import LibComm
            # The library's own communicator.  If it cannot be created on this system (every rank decides together:
            # the flags are all-reduced over torch's group) the run ENDS with the library's message on every rank — unless
            # TSEM_ALLOW_TORCH_COLLECTIVES=1 opted into torch.distributed collectives on the engine's reduce buffer
            # (decide_transport).  TSEM_TORCH_COLLECTIVES=1 makes the creation "fail" on purpose (tests of both outcomes).
            failed = os.environ.get('TSEM_TORCH_COLLECTIVES', '0') == '1'
            if failed:
                self.lib_error = 'TSEM_TORCH_COLLECTIVES=1'
            ids = [None]
            if not failed and self.rank == 0:
                try:
                    ids = [LibComm.unique_id()]
                except Exception as exc:                      # noqa: BLE001
                    failed, self.lib_error = True, str(exc)
            dist.broadcast_object_list(ids, src=0, group=group, device=self.tdev)
            if ids[0] is not None and not failed:
                # ncclCommInitRank is a collective that can HANG (a rank that never arrives, a fabric that does not
                # come up): it runs in a helper thread with a bounded wait.  A rank whose wait expires votes for the
                # torch transport below (the vote is a MAX all-reduce over torch's group, so every rank takes the
                # same decision); the stuck helper thread is abandoned (daemon).
                import threading
                box = {}

                def _create():
                    try:
                        box['lib'] = LibComm(self.device, ids[0], self.rank, self.world)
                    except Exception as exc:                  # noqa: BLE001
                        box['err'] = str(exc)
                th = threading.Thread(target=_create, name='tsem-comm-init', daemon=True)
                th.start()
                th.join(float(os.environ.get('TSEM_COMM_INIT_TIMEOUT', '120')))
                if th.is_alive():
                    failed, self.lib_error = True, 'ncclCommInitRank did not return within the time limit'
                elif 'lib' in box:
                    self.lib = box['lib']
                else:
                    failed, self.lib_error = True, box.get('err', 'communicator creation failed')
            else:
                failed = True
            flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=self.tdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            if int(flag.item()):
                if self.lib is not None:
                    self.lib.close()
                self.lib = None
                decide_transport(True, getattr(self, 'lib_error', 'another rank reported the failure'))   # raises without the opt-in
                if self.rank == 0:
                    import warnings
                    warnings.warn('telescope_amd: the library RCCL communicator is not in use (%s); collectives go '
                                  'through torch.distributed, one host round trip per EM iteration'
                                  % getattr(self, 'lib_error', 'TSEM_TORCH_COLLECTIVES=1'))
        else:
            self.device = 0 if device is None else device
            self.tdev = torch.device('cpu')
        self._red = None
        self._eng = None

    def describe(self):
        """Transport, rank count and the RCCL copy in use — for logs and the bench line."""
        if self.lib is not None:
            from ._lib import comm_library_info
            return 'in-library %s, %d ranks' % (comm_library_info(), self.world)
        t = self._torch
        ver = getattr(t.cuda, 'nccl', None)
        try:
            ver = '.'.join(str(v) for v in t.cuda.nccl.version()) if self.backend == 'nccl' else ''
        except Exception:                                     # noqa: BLE001
            ver = ''
        return 'torch.distributed %s%s, %d ranks' % (self.backend, (' (rccl ' + ver + ')') if ver else '', self.world)

    def close(self):
        if self.lib is not None:
            _detach(getattr(self, '_eng', None), self.in_library)   # (the engine must not keep a pointer to a destroyed communicator)
            self.in_library = False
            self.lib.close()
            self.lib = None

    # ---- small host-side collectives (setup / reporting) --------------------
    def _allreduce_np(self, a, op, dtype):
        t = self._torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.tdev)
        self._dist.all_reduce(t, op=op, group=self.group)
        return t.cpu().numpy()

    def max_scalar(self, v):
        if self.lib is not None:
            return int(self.lib.allreduce([v], 'i64', 'max')[0])
        return int(self._allreduce_np(np.array([v], np.int64), self._dist.ReduceOp.MAX,
                                      self._torch.int64)[0])

    def sum_array(self, a):
        if self.lib is not None:
            return self.lib.allreduce(a, 'f64', 'sum').reshape(np.shape(a))
        return self._allreduce_np(np.asarray(a, np.float64), self._dist.ReduceOp.SUM,
                                  self._torch.float64)

    def max_array(self, a):
        if self.lib is not None:
            return self.lib.allreduce(a, 'f64', 'max').reshape(np.shape(a))
        return self._allreduce_np(np.asarray(a, np.float64), self._dist.ReduceOp.MAX,
                                  self._torch.float64)

    def sum_array_u64(self, a):
        """Wrap-around integer sums (two's complement int64 == uint64 mod 2^64)."""
        if self.lib is not None:
            return self.lib.allreduce(a, 'u64', 'sum').reshape(np.shape(a))
        a = np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)
        out = self._allreduce_np(a, self._dist.ReduceOp.SUM, self._torch.int64)
        return out.view(np.uint64)

    def gather_rows(self, a):
        """int32 vectors of every rank on rank 0 (list in rank order; the other ranks get placeholders).  Tensor
        collectives on the group's device — the tied rows of `choose` are millions of counts per rank, and the
        object collectives used here before pickled them through torch's store."""
        t, dist = self._torch, self._dist
        a = np.ascontiguousarray(a, dtype=np.int32).ravel()
        n = t.tensor([a.size], dtype=t.int64, device=self.tdev)
        sizes = [t.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n, group=self.group)
        sizes = [int(x.item()) for x in sizes]
        m = max(1, max(sizes))
        buf = t.zeros(m, dtype=t.int32, device=self.tdev)
        if a.size:
            buf[:a.size] = t.from_numpy(a).to(self.tdev)
        outs = [t.empty(m, dtype=t.int32, device=self.tdev) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(buf, outs, dst=0, group=self.group)
        self._row_sizes = sizes
        if self.rank != 0:
            return [None] * self.world
        return [o[:k].cpu().numpy() for o, k in zip(outs, sizes)]

    def scatter_rows(self, parts):
        """The inverse of the last gather_rows: rank r receives parts[r] (an int32 vector of the length it sent)."""
        t, dist = self._torch, self._dist
        sizes = self._row_sizes
        m = max(1, max(sizes))
        out = t.empty(m, dtype=t.int32, device=self.tdev)
        src = None
        if self.rank == 0:
            src = []
            for p_, k in zip(parts, sizes):
                b = t.zeros(m, dtype=t.int32, device=self.tdev)
                if k:
                    b[:k] = t.from_numpy(np.ascontiguousarray(p_, dtype=np.int32)).to(self.tdev)
                src.append(b)
        dist.scatter(out, src, src=0, group=self.group)
        return out[:sizes[self.rank]].cpu().numpy()

    def barrier(self):
        if self.lib is not None:
            self.lib.allreduce([0.0], 'f64', 'sum')
        else:
            self._dist.barrier(group=self.group)

    # ---- the per-iteration exchange ---------------------------------------------
    def attach(self, engine, n_cols):
        """Real engine on RCCL: hand it the library communicator — it then all-reduces its own reduce
        buffer inside `em_chunk` and nothing below is used.  Otherwise (gloo + the tests-only engine):
        give the engine a reduce buffer that torch can all-reduce in place."""
        self._eng = engine
        if self.lib is not None:
            engine.comm_attach(self.lib.handle)
            self.in_library = True
            return
        from ._lib import Engine
        t = self._torch
        self.in_library = False
        self._staged = False
        if isinstance(engine, Engine) and self.backend != 'nccl':
            # A HIP engine's reduce buffer lives in HBM and this backend reduces host tensors.  Refused — except for the
            # DRY RUN of a multi-process job on a box with fewer GPUs than ranks (TSEM_GLOO_HOST_STAGED=1, bench.py
            # --one-device): several ranks share a device, which RCCL does not allow, so the buffer takes a round trip
            # through the host and gloo per iteration.  Same numbers, same host logic; not a product transport.
            if os.environ.get('TSEM_GLOO_HOST_STAGED', '0') != '1':
                raise RuntimeError('a HIP engine needs the nccl (RCCL) backend: its reduce buffer lives in HBM and the %s '
                                   'backend reduces host tensors' % self.backend)
            self._staged = True
            self._red = t.zeros(n_cols + 2, dtype=t.float64, device=t.device('cuda', self.device))
            engine.bind_reduce_buffer(self._red.data_ptr(), n_cols + 2)
            return
        self._red = t.zeros(n_cols + 2, dtype=t.float64, device=self.tdev)
        engine.bind_reduce_buffer(self._red.data_ptr(), n_cols + 2)

    def allreduce_device(self, engine, offset=0, count=None):
        if self.in_library:
            engine.comm_allreduce(offset, count)
            return
        red = self._red if count is None else self._red[offset:offset + count]
        if getattr(self, '_staged', False):                   # dry run: HBM -> host -> gloo -> HBM
            engine.synchronize()
            h = red.cpu()
            self._dist.all_reduce(h, op=self._dist.ReduceOp.SUM, group=self.group)
            red.copy_(h)
            self._torch.cuda.synchronize()
            return
        # the engine's kernels run on ITS stream, torch's collective on torch's current stream: order them through the
        # host (this transport makes a host round trip per iteration anyway)
        if self.backend == 'nccl':
            engine.synchronize()
        self._dist.all_reduce(red, op=self._dist.ReduceOp.SUM, group=self.group)
        if self.backend == 'nccl':
            self._torch.cuda.current_stream().synchronize()


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from torchrun's environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR/PORT) and return a Comm; None when single-process
    (unless `force`, which builds a 1-rank group — used to exercise the N > 1 code path)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and not force:
        return None
    if world <= 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('LOCAL_RANK', '0')
    if os.environ.get('MASTER_ADDR', '') in ('127.0.0.1', 'localhost', '::1'):
        # a single-node job: RCCL's bootstrap sockets (torch's communicator and the library's) need not guess an interface in a
        # container whose other interfaces may be unusable; the data path is xGMI / P2P either way
        os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = os.environ.get('TSEM_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    local = 0 if os.environ.get('TSEM_ONE_DEVICE', '0') == '1' else int(os.environ.get('LOCAL_RANK', 0))
    if backend == 'nccl':
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend=backend, **kw)
    return Comm(device=local)


class ThreadGroup(object):
    """Shared state of `world` ranks that live in ONE process as host threads, each with its own engine on the
    SAME device: the library's in-process transport (`tsem_comm_create_local`, include/telescope_em.h) plus the
    little host-side glue `choose` needs.  What a one-GPU box can run of a row-sharded job: the same
    `tsem_em_chunk`, reduce buffer, error slot and device-side stop flag as under RCCL."""

    def __init__(self, device, world):
        import threading
        from ._lib import LocalGroup
        self.device, self.world = device, world
        self.lib_group = LocalGroup(device, world)
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.box = None

    def comm(self, rank):
        return ThreadComm(self, rank)

    def close(self):
        self.lib_group.close()


class ThreadComm(object):
    """One rank of a ThreadGroup — the interface of `Comm`."""
    backend = 'in-process'

    def __init__(self, group, rank):
        self.g, self.rank, self.world, self.device = group, rank, group.world, group.device
        self.lib = group.lib_group.comm(rank)
        self.in_library = False

    def describe(self):
        return 'in-process transport (threads sharing GPU %d), %d ranks' % (self.device, self.world)

    def close(self):
        if self.lib is not None:
            _detach(getattr(self, '_eng', None), self.in_library)
            self.in_library = False
            self.lib.close()
            self.lib = None

    def max_scalar(self, v):
        return int(self.lib.allreduce([v], 'i64', 'max')[0])

    def sum_array(self, a):
        return self.lib.allreduce(a, 'f64', 'sum').reshape(np.shape(a))

    def max_array(self, a):
        return self.lib.allreduce(a, 'f64', 'max').reshape(np.shape(a))

    def sum_array_u64(self, a):
        return self.lib.allreduce(a, 'u64', 'sum').reshape(np.shape(a))

    def gather_rows(self, a):
        g = self.g
        g.slots[self.rank] = np.asarray(a)
        g.barrier.wait()
        out = list(g.slots) if self.rank == 0 else [None] * self.world
        g.barrier.wait()
        return out

    def scatter_rows(self, parts):
        g = self.g
        if self.rank == 0:
            g.box = list(parts)
        g.barrier.wait()
        mine = g.box[self.rank]
        g.barrier.wait()
        return mine

    def barrier(self):
        self.lib.allreduce([0.0], 'f64', 'sum')

    def attach(self, engine, n_cols):
        self._eng = engine
        engine.comm_attach(self.lib.handle)
        self.in_library = True

    def allreduce_device(self, engine, offset=0, count=None):
        engine.comm_allreduce(offset, count)
