#!/usr/bin/env python3
"""Benchmark of the EM reassignment hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE EM iteration (fused E-step + M-step pass over every stored
entry, the all-reduce of the per-locus column sums when N > 1, and the
parameter update) of `TelescopeLikelihood.em()` on a synthetic fragment x locus
score matrix generated on the device (telescope_amd/synthetic.py spec).  The
default workload is BASELINE.json's north-star case: 50M fragments x 30k loci,
~40 stored entries per row, fp64 — strong scaling (total work fixed) for N > 1.

Rank 0 prints ONE JSON line.  `roofline` is the dominant (EM pass) kernel's
algorithmic bytes per launch / its HIP-event-timed duration against the 8 TB/s
HBM peak; `cpu_baseline` times the oracle (scipy operator sequence of the
reference) on a bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# The parameters after W + K iterations of the DEFAULT workload (50M x 30k x ~40, zipf, seed 42, fp64 entries, 1 GPU), folded to the
# three numbers of the `check` block — measured on MI355X (BENCH_r04.json: 25 iterations; profiles/r05_bench.json: 23).  The rows are
# generated from their GLOBAL index, so N ranks hold the same matrix: a run at any N must reproduce them to summation order
# (CHECK_RTOL).  `check.matches_n1` compares with the N = 1 run made in the same process when N > 1, with this table at N = 1.
CHECK_RTOL = 1e-11
EMBEDDED_CHECK = {
    25: dict(pi_sum=0.9999999999999775, pi_weighted=0.05583806323813835, theta_weighted=0.49818481912061724),
    23: dict(pi_sum=0.9999999999999775, pi_weighted=0.055838211628101916, theta_weighted=0.49818481973259593),
}
PHASE_ITERS = 8         # iterations timed phase by phase (HIP events between the kernels) after the timed region


class Opts(object):
    def __init__(self, max_iter, em_epsilon=0.0):
        self.em_epsilon, self.max_iter = em_epsilon, max_iter
        self.pi_prior, self.theta_prior = 0, 200000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=0, choices=(0, 2, 3, 4, 5),
                    help='a BASELINE.json configuration by its number (sets --rows / --cols / --nnz-row / --scaling / --value-format): '
                         '2 = 1M x 30k x ~20 on one GPU; 3 = 10M x 30k x ~40; 4 = 50M x 30k x ~40 row-sharded over --gpus N (strong scaling: '
                         'the default workload); 5 = 200M x 50k x ~100 over 8 GPUs as 25M rows per rank (weak scaling, score codes: the '
                         'library default) with the size-independent property checks — at --gpus 1 the half that fits one GPU (100M rows)')
    ap.add_argument('--config-scale', type=float, default=1.0,
                    help='with --config: the same configuration at this fraction of its rows (dry runs of the plumbing; the line says so)')
    ap.add_argument('--properties', action='store_true',
                    help='after the timed region run the size-independent checks (every rank takes part): pi / theta are distributions, '
                         '`all` counts every stored entry, exclude + tied rows = all fragments, average sums to the fragments, no fall-back, '
                         'and the same iterations on the two-pass kernels agree to 1e-10 (implied by --config 5)')
    ap.add_argument('--rows', type=int, default=50_000_000)
    ap.add_argument('--cols', type=int, default=30_000)
    ap.add_argument('--nnz-row', type=float, default=40.0)
    ap.add_argument('--dist', choices=('zipf', 'uniform', 'family'), default='zipf')
    ap.add_argument('--uniq-frac', type=float, default=0.0)
    ap.add_argument('--seed', type=int, default=42)
    ap.add_argument('--scaling', choices=('strong', 'weak'), default='strong')
    ap.add_argument('--em-kernel', choices=('auto', 'twopass', 'fused'), default='auto')
    ap.add_argument('--value-format', choices=('auto', 'f64', 'code16'), default='f64',
                    help='entry format of the blocked layout: fp64 Q values (12 B/nnz: the reference\'s own storage and '
                         'the HBM-bound workload BASELINE.json\'s roofline target is stated on — the headline) or uint16 '
                         'score codes + LDS score table (6 B/nnz, the same fp64 arithmetic bit for bit: the library\'s '
                         '`auto` choice, timed beside the headline as `code16_layout`)')
    ap.add_argument('--block-rows', type=int, default=0)
    ap.add_argument('--chunk-blocks', type=int, default=0)
    ap.add_argument('--fused-dbg', type=int, default=0)
    ap.add_argument('--sorted-fill', type=int, default=-1, help='sub-block order: 1 row order, 0 strand-transposed (-1 auto)')
    ap.add_argument('--geometry', type=int, default=-1, help='fused kernel geometry of teams of 1-4: 0 or 2 (-1 auto)')
    ap.add_argument('--issue-early', type=int, default=-1, help='fused kernel: partner loads before (1) / after (0) the combine (-1 auto)')
    ap.add_argument('--kernel-timing', type=int, default=4,
                    help='HIP events around every n-th EM pass of the timed region (roofline.kernel_ms); 0 = none')
    ap.add_argument('--deconflict', type=int, default=-1, help='conflict-aware entry order inside rows of the code16 layout: -1 library default (on), 0 off')
    ap.add_argument('--parts', type=int, default=0, help='column parts per team (0 = fewest that fit LDS)')
    ap.add_argument('--hot-split', type=int, default=1, help='0: one accumulator slot per column (experiments)')
    ap.add_argument('--cpu-sample-rows', type=int, default=400_000)
    ap.add_argument('--cpu-iters', type=int, default=3, help='EM iterations timed on the CPU samples (SURVEY 8(d): T = 3)')
    ap.add_argument('--cpu-large-rows', type=int, default=5_000_000,
                    help='second CPU timing sample: this many rows at the workload\'s own entries per row (SURVEY 8(d): 5M x 40 when the '
                         'host has the RAM, ~25 GB peak); 0 skips it.  The first sample is BASELINE config 2 (1M rows x ~20 per row)')
    ap.add_argument('--cpu-fused-rows', type=int, default=4_000_000,
                    help='sample rows for the all-cores C baseline (oracle/em_fused.c); 0 skips it')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt-layout', action='store_true',
                    help='skip the second measurement with 2-byte score codes (N=1, default --value-format only)')
    ap.add_argument('--no-reproducible-leg', action='store_true', help='skip the `reproducible_mode` block (option reproducible on a 10M-row sample)')
    ap.add_argument('--no-precision-sweep', action='store_true',
                    help='skip the fp32-vs-fp64 tolerance sweep of BASELINE config 3 (N=1 only; ~10 s)')
    ap.add_argument('--one-device', action='store_true',
                    help='DRY RUN of --gpus N on a box with fewer GPUs: every rank uses device 0, torch.distributed gloo, the reduce '
                         'buffer staged through the host (RCCL refuses two ranks on one device).  Exercises the launcher, the row '
                         'shards, the set-up and per-iteration collectives and the result line; its throughput means nothing')
    ap.add_argument('--fail-rank', type=int, default=-1,
                    help='TEST HOOK: the persistent kernel of this rank reports a hand-off time-out in its first EM pass (fused_dbg bit 5); '
                         'every rank must refuse to commit, this rank falls back to the two-pass kernels, all redo the iteration')
    ap.add_argument('--no-n1-reference', action='store_true',
                    help='N > 1: skip the N = 1 run of the whole problem on rank 0\'s GPU (speedup_vs_n1, check.matches_n1)')
    ap.add_argument('--force-comm', action='store_true',
                    help='use the multi-rank code path (RCCL group, per-iteration all-reduce) even at world size 1')
    args = ap.parse_args()
    args.config_note = None
    if args.config:
        n = max(1, args.gpus)
        sc = args.config_scale
        if args.config == 2:
            args.rows, args.cols, args.nnz_row, args.scaling = int(1_000_000 * sc), 30_000, 20.0, 'strong'
            args.config_note = 'BASELINE config 2 (synthetic 1M fragments x 30k loci, ~20 nnz/row, fp64, 1 GPU)'
        elif args.config == 3:
            args.rows, args.cols, args.nnz_row, args.scaling = int(10_000_000 * sc), 30_000, 40.0, 'strong'
            args.config_note = 'BASELINE config 3 (synthetic 10M fragments x 30k loci, ~40 nnz/row, 1 GPU; the fp32-vs-fp64 sweep is the `precision_sweep` block)'
        elif args.config == 4:
            args.rows, args.cols, args.nnz_row, args.scaling = int(50_000_000 * sc), 30_000, 40.0, 'strong'
            args.config_note = 'BASELINE config 4 (synthetic 50M fragments x 30k loci row-sharded across %d GPU(s), one all-reduce of the column sums per iteration)' % n
        else:
            # 200M x 50k x ~100 = 2e10 stored entries over 8 GPUs = 25M rows per rank; fewer ranks keep 25M rows each (weak scaling),
            # one GPU alone takes the 1e10-entry half that fits it (tests/test_gpu_round5.py test_half_of_config5_on_one_gpu)
            per_rank = 100_000_000 if n == 1 else 25_000_000
            args.rows, args.cols, args.nnz_row, args.scaling = int(per_rank * sc), 50_000, 100.0, 'weak'
            args.value_format = 'auto'
            args.properties = True
            args.config_note = ('BASELINE config 5 (pooled single-cell style: 200M fragments x 50k loci, ~100 nnz/row, 8 GPUs) as %d rank(s) x %dM rows%s'
                                % (n, per_rank // 1_000_000, '' if n == 8 else ' — %s of the configuration' % ('the half that fits one GPU' if n == 1 else '%d / 8' % n)))
        if sc != 1.0:
            args.config_note += ' AT %g OF ITS ROWS (a dry run of the plumbing)' % sc
    return args


def cpu_baseline(args, dist_code, cdf):
    """Oracle (port of the reference's scipy op sequence) on a bounded sample,
    and the GPU result on the SAME sample -> lnl / pi deltas."""
    from telescope_amd._lib import Engine
    from telescope_amd.likelihood import TelescopeLikelihood
    from oracle.telescope_oracle import OracleModel
    import scipy.sparse as sp
    n = min(args.cpu_sample_rows, args.rows)
    T = args.cpu_iters

    def time_oracle(rows, nnz_row):
        """T EM iterations (estep + mstep, model.py:773-774) of the oracle on `rows` rows x ~nnz_row entries per row of the same
        generator (same columns, distribution, seed): seconds per iteration, stored entries, set-up seconds."""
        e = Engine(0)
        e.generate(0, rows, args.cols, synthetic.poisson_cdf_u32(nnz_row), args.seed, dist_code, args.uniq_frac)
        ip_, ix_, rw_ = e.export_csr()
        e.close()
        ts = time.perf_counter()
        m = OracleModel(sp.csr_matrix((rw_, ix_, ip_), shape=(rows, args.cols)), 0, 200000)
        ts = time.perf_counter() - ts
        t0_ = time.perf_counter()
        for _ in range(T):
            z_ = m.estep(m.pi, m.theta)
            m.pi, m.theta = m.mstep(z_)
        dt_ = time.perf_counter() - t0_
        return dict(rows=rows, nnz_row=nnz_row, nnz=int(ip_[-1]), iters=T, sec_per_iter=dt_ / T, nnz_per_sec=int(ip_[-1]) * T / dt_,
                    setup_s=ts)

    from telescope_amd import synthetic
    timings = [time_oracle(min(1_000_000, args.rows), 20.0)]            # BASELINE config 2: 1M x 30k x ~20
    if args.cpu_large_rows > 0:
        try:
            import psutil
            free_gb = psutil.virtual_memory().available / 2 ** 30
        except Exception:   # noqa: BLE001
            free_gb = 0.0
        need_gb = 140e-9 * args.cpu_large_rows * args.nnz_row + 4.0       # ~124 B per stored entry at the peak + slack
        if free_gb >= need_gb and args.rows >= args.cpu_large_rows:
            timings.append(time_oracle(args.cpu_large_rows, args.nnz_row))
        else:
            timings.append(dict(rows=args.cpu_large_rows, nnz_row=args.nnz_row, skipped='%.0f GB of host memory free, %.0f GB needed'
                                % (free_gb, need_gb)))
    best = [t for t in timings if 'nnz_per_sec' in t][-1]                # extrapolate from the LARGEST sample that ran
    eng = Engine(0)
    eng.generate(0, n, args.cols, cdf, args.seed, dist_code, args.uniq_frac)
    ip, ix, rw = eng.export_csr()
    tl = TelescopeLikelihood.from_engine(eng, Opts(T))
    tl.em()
    raw = sp.csr_matrix((rw, ix, ip), shape=(n, args.cols))
    om = OracleModel(raw, 0, 200000)
    for _ in range(T):                      # the EM loop proper: estep + mstep (model.py:773-774)
        z = om.estep(om.pi, om.theta)
        pi, theta = om.mstep(z)
        om.z, om.pi, om.theta = z, pi, theta
    # z is from the E-step before the last M-step, like model.py:795-801
    lnl_ref = om.calculate_lnl(om.z, om.pi, om.theta)
    nnz = int(ip[-1])
    rate, dt = best['nnz_per_sec'], best['sec_per_iter'] * T
    # per-locus final counts (output_report, model.py:435-457) on the sample: the integer mode must agree
    # exactly, the confidence-weighted one to rounding
    np.random.seed(args.seed)
    g_excl, g_conf = tl.reassign_colsums('exclude', 0.9), tl.reassign_colsums('conf', 0.9)
    o_excl = np.asarray(om.reassign('exclude', 0.9).sum(0)).ravel()
    o_conf = np.asarray(om.reassign('conf', 0.9).sum(0)).ravel()
    # one run to CONVERGENCE on a smaller sample: the iteration the device-side test (diff_est < epsilon, model.py:792)
    # stops in must be the oracle's.  epsilon / max_iter are chosen so that the run ends well below the cap (round 2
    # reported 100 = 100, i.e. the cap; tests/test_gpu_round3.py asserts the same on 1M- and 2M-row matrices against
    # the C oracle)
    nc = min(40_000, n)
    eps_c, cap_c = 1e-5, 1000
    eng_c = Engine(0)
    eng_c.generate(0, nc, args.cols, cdf, args.seed, dist_code, args.uniq_frac)
    ipc, ixc, rwc = eng_c.export_csr()
    tlc = TelescopeLikelihood.from_engine(eng_c, Opts(cap_c, eps_c))
    tlc.em()
    omc = OracleModel(sp.csr_matrix((rwc, ixc, ipc), shape=(nc, args.cols)), 0, 200000)
    omc.em(eps_c, cap_c)
    fused_c = None
    if args.cpu_fused_rows > 0:
        try:   # a strong CPU baseline: the same EM as one fused OpenMP pass per iteration, all host cores
            from oracle.em_fused import em_fused
            nf = min(args.cpu_fused_rows, args.rows)
            eng_f = Engine(0)
            eng_f.generate(0, nf, args.cols, cdf, args.seed, dist_code, args.uniq_frac)
            ipf, ixf, rwf = eng_f.export_csr()
            eng_f.close()
            rawf = sp.csr_matrix((rwf, ixf, ipf), shape=(nf, args.cols))
            em_fused(rawf, 0, 200000, 0.0, 1)                                  # warm the threads / page cache
            t1 = time.perf_counter(); em_fused(rawf, 0, 200000, 0.0, 1); t1 = time.perf_counter() - t1
            t5 = time.perf_counter(); r5 = em_fused(rawf, 0, 200000, 0.0, 5); t5 = time.perf_counter() - t5
            per_iter = max(1e-9, (t5 - t1) / 4.0)                               # setup and the final lnl pass cancel out
            fused_c = dict(nnz_per_sec=int(ipf[-1]) / per_iter, sec_per_iter=per_iter, sample_rows=nf,
                           sample_nnz=int(ipf[-1]), cores=os.cpu_count(), lnl=float(r5['lnl']))
        except Exception as e:   # noqa: BLE001 — the extra baseline must never break the bench line
            fused_c = dict(error=repr(e))
    return dict(nnz_per_sec=rate, sec_per_iter=dt / T, sample_nnz=best['nnz'], sample_rows=n, iters=T, fused_c=fused_c, timings=timings, best=best,
                lnl_ref=float(lnl_ref), lnl_gpu=float(tl.lnl),
                lnl_rel_delta=abs(tl.lnl - lnl_ref) / abs(lnl_ref),
                pi_max_rel_delta=float(np.max(np.abs(tl.pi - om.pi) / np.maximum(om.pi, 1e-300))),
                final_count_mismatches=int(np.count_nonzero(g_excl != o_excl)),
                # rows of the sample whose best hits hung on the last bits of 1 / rowsum: redone with the reference's order of additions
                # (tsem_npsum.h).  The end-to-end rate on adversarial matrices (the oracle's OWN parameters, two-score matrices) is measured
                # by tests/fuzz_reports.py `own`: profiles/r06_fuzz_reports.txt
                near_tie_rows=int(tl._eng.layout_info().get('near_tie_rows', 0)),
                final_conf_max_rel_delta=float(np.max(np.abs(g_conf - o_conf) / np.maximum(np.abs(o_conf), 1e-300))),
                converged_run=dict(rows=nc, em_epsilon=eps_c, max_iter=cap_c, iterations_gpu=int(tlc.n_iter), iterations_ref=int(omc.n_iter),
                                   converged_gpu=bool(tlc.converged), converged_ref=bool(omc.converged),
                                   lnl_gpu=float(tlc.lnl), lnl_ref=float(omc.lnl),
                                   lnl_rel_delta=abs(tlc.lnl - omc.lnl) / abs(omc.lnl)))


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the same
    environment torch.distributed.run would give them), relay rank 0's JSON line as OUR last stdout line, and return
    the worst exit code.  Rank r's stderr is passed through; its stdout too, except for rank 0's, which is kept so
    that the result line can be printed last."""
    import socket
    import subprocess
    import threading
    n = args.gpus
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs, out0 = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), TSEM_BENCH_SELF_LAUNCHED='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else None, text=(r == 0)))
    th = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    th.start()
    rcs = [None] * n
    deadline = None
    hard = time.time() + float(os.environ.get('TSEM_BENCH_TIMEOUT', '1500'))   # nothing here may hang the caller for good
    while any(rc is None for rc in rcs):
        if time.time() > hard and deadline is None:
            print('bench.py: ranks still running after TSEM_BENCH_TIMEOUT s: stopping them', file=sys.stderr, flush=True)
            deadline = 0.0
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(rc not in (None, 0) for rc in rcs) and deadline is None:
            deadline = time.time() + 30.0                       # a rank failed: the others may sit in a collective
        if deadline is not None and time.time() > deadline:
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.kill()                                    # exactly the processes started above
        time.sleep(0.05)
    th.join(10)
    lines = [ln.rstrip('\n') for ln in out0 if ln.strip()]
    result = None
    for ln in reversed(lines):
        if ln.startswith('{') and '"metric"' in ln:
            result = ln
            break
    for ln in lines:
        if ln is not result:
            print(ln)
    worst = max((abs(rc) for rc in rcs), default=0)
    if result is None and worst == 0:
        worst = 1
    if result is not None:
        print(result, flush=True)
    return worst


def main():
    args = parse()
    if 'RANK' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    args.gpus = world
    from telescope_amd import synthetic
    from telescope_amd._lib import Engine, EMK_AUTO, EMK_FUSED, EMK_TWOPASS
    from telescope_amd.distributed import init_from_env, shard_bounds
    from telescope_amd.likelihood import TelescopeLikelihood, EM_CHUNK

    # the engine first: without a usable GPU every rank fails HERE, loudly ("no CPU fallback"), not in the launcher
    if args.one_device:
        os.environ['TSEM_ONE_DEVICE'] = '1'
        os.environ['TSEM_GLOO_HOST_STAGED'] = '1'
    eng = Engine(0 if args.one_device else int(os.environ.get('LOCAL_RANK', '0')))
    comm = init_from_env('gloo' if args.one_device else 'nccl', force=args.force_comm) if (world > 1 or args.force_comm) else None
    rank = comm.rank if comm else 0
    local = comm.device if comm else 0
    total_rows = args.rows * (world if args.scaling == 'weak' else 1)
    r0, r1 = shard_bounds(total_rows, world, rank)
    dist_code = synthetic.DIST_CODE[args.dist]
    cdf = synthetic.poisson_cdf_u32(args.nnz_row)

    eng.set_option('row_offset', r0)
    # (--one-device: the persistent fused kernel needs all its workgroups resident at once, which two processes sharing a GPU cannot
    #  promise each other — the hand-off watchdog would catch it and fall back; the dry run is about the plumbing, so it starts there)
    eng.set_option('em_kernel', EMK_TWOPASS if args.one_device and args.em_kernel == 'auto' else
                   {'auto': EMK_AUTO, 'twopass': EMK_TWOPASS, 'fused': EMK_FUSED}[args.em_kernel])
    eng.set_option('value_format', {'auto': 0, 'f64': 1, 'code16': 2}[args.value_format])
    if args.block_rows:
        eng.set_option('block_rows', args.block_rows)
    if args.parts:
        eng.set_option('parts', args.parts)
    if args.geometry >= 0:
        eng.set_option('geometry', args.geometry)
    if args.sorted_fill >= 0:
        eng.set_option('sorted_fill', args.sorted_fill)
    if args.chunk_blocks:
        eng.set_option('chunk_blocks', args.chunk_blocks)
    if args.fused_dbg or args.fail_rank == rank:
        eng.set_option('fused_dbg', args.fused_dbg | (32 if args.fail_rank == rank else 0))
    eng.set_option('hot_split', args.hot_split)
    eng.set_option('deconflict', args.deconflict)
    eng.set_option('kernel_timing', args.kernel_timing)
    if args.issue_early >= 0:
        eng.set_option('issue_early', args.issue_early)
    t_setup = time.perf_counter()
    eng.generate(r0, r1, args.cols, cdf, args.seed, dist_code, args.uniq_frac)
    tl = TelescopeLikelihood.from_engine(eng, Opts(args.steps), comm)
    tl.keep_kernel_timing = True        # em() leaves the per-pass HIP events on: roofline.kernel_ms comes from the timed region
    eng.synchronize()
    t_setup = time.perf_counter() - t_setup
    _, _, nnz_local = eng.dims()

    import logging

    def run(n):
        # `TelescopeLikelihood.em()` itself (likelihood.py), n iterations at em_epsilon = 0: chunks of EM_CHUNK (32) iterations per host
        # synchronisation — pass, in-library RCCL all-reduce (N > 1 / --force-comm), update, the device-side convergence test (never
        # true at epsilon 0) — or, on the fall-back transport, one torch.distributed all-reduce and one host round trip per
        # iteration with the time-out recovery.  Only the log-likelihood pass AFTER the loop (model.py:800-801) is left out: a step
        # is one EM iteration; `whole_em_call` below times the call with it.
        tl.max_iter, tl.epsilon = n, 0.0
        tl.em(loglev=logging.DEBUG, final_lnl=False)
        assert tl.n_iter == n, (tl.n_iter, n)

    def fence():
        eng.synchronize()
        if comm is not None:
            import torch
            torch.cuda.synchronize()
            comm.barrier()
            torch.cuda.synchronize()

    if args.warmup:
        run(args.warmup)
    eng.kernel_stats(reset=True)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    ks = eng.kernel_stats()
    nnz_total = nnz_local
    if comm is not None:
        elapsed = float(comm.max_array(np.array([elapsed]))[0])
        nnz_total = int(comm.sum_array(np.array([float(nnz_local)]))[0])

    # the parameters every rank now holds, folded to three numbers: the same for every N at strong scaling (the rows are generated
    # from their GLOBAL index, so N ranks hold the same matrix) — what a run at N > 1 is checked against
    wts = np.arange(1, args.cols + 1, dtype=np.float64) / args.cols

    def fold(e, iterations):
        pi_, theta_ = e.get_params()
        return dict(iterations=iterations, pi_sum=float(pi_.sum()), pi_weighted=float(np.dot(pi_, wts)),
                    theta_weighted=float(np.dot(theta_, wts)))

    def same(a, b):
        return all(abs(a[k] - b[k]) <= CHECK_RTOL * max(abs(b[k]), 1e-300) for k in ('pi_sum', 'pi_weighted', 'theta_weighted'))

    check = fold(eng, args.warmup + args.steps) if rank == 0 else None

    # ---- where an iteration's time goes: PHASE_ITERS more iterations on EVERY rank (the all-reduce is collective) with a HIP event at
    # every phase boundary — pass | column reduce | all-reduce | update | gap to the next pass.  Outside the timed region; the events
    # cost the stream a few microseconds each, so the phases add up to slightly more than `ms_per_step`.
    eng.set_option('kernel_timing', 0)
    eng.set_option('phase_timing', 1)
    eng.phase_times(reset=True)
    run(PHASE_ITERS)
    fence()
    phases = eng.phase_times(reset=True)
    eng.set_option('phase_timing', 0)
    eng.set_option('kernel_timing', args.kernel_timing)
    props = None
    if args.properties:
        props = properties_leg(tl, eng, comm, total_rows, nnz_total, args)     # collective: every rank
    if rank != 0:
        _shutdown(comm)
        return

    n1 = None
    if world > 1 and args.scaling == 'strong' and not args.no_n1_reference:
        # the N = 1 reference in THIS process: the whole problem on rank 0's GPU (it fits: 24 GB of entries), same options, same
        # warm-up + steps, no communicator — its time is what `speedup_vs_n1` divides by, its parameters what `check.matches_n1`
        # compares with.  (The other ranks have left; nothing here is collective.)
        try:
            e1 = Engine(local)
            for k_, v_ in getattr(eng, 'options', {}).items():
                if k_ not in ('row_offset', 'fused_dbg', 'phase_timing', 'kernel_timing', 'em_kernel'):
                    e1.set_option(k_, v_)
            e1.set_option('em_kernel', EMK_TWOPASS if args.one_device and args.em_kernel == 'auto' else
                          {'auto': EMK_AUTO, 'twopass': EMK_TWOPASS, 'fused': EMK_FUSED}[args.em_kernel])
            e1.set_option('kernel_timing', args.kernel_timing)
            e1.generate(0, total_rows, args.cols, cdf, args.seed, dist_code, args.uniq_frac)
            tl1 = TelescopeLikelihood.from_engine(e1, Opts(args.steps), None)
            tl1.keep_kernel_timing = True
            if args.warmup:
                tl1.max_iter, tl1.epsilon = args.warmup, 0.0
                tl1.em(loglev=logging.DEBUG, final_lnl=False)
            e1.kernel_stats(reset=True)
            e1.synchronize()
            tl1.max_iter, tl1.epsilon = args.steps, 0.0
            t1 = time.perf_counter()
            tl1.em(loglev=logging.DEBUG, final_lnl=False)
            e1.synchronize()
            el1 = time.perf_counter() - t1
            ks1 = e1.kernel_stats()
            n1 = dict(ms_per_step=el1 / args.steps * 1e3, kernel_ms=ks1['em_ms'] / max(1, ks1['em_launches']),
                      check=fold(e1, args.warmup + args.steps),
                      how='the whole problem (%d rows) on rank 0\'s GPU in this process, same options, no communicator' % total_rows)
            e1.close()
            del tl1
        except Exception as e:   # noqa: BLE001 — the reference leg must never cost the result line
            n1 = dict(error=repr(e))
    emb = EMBEDDED_CHECK.get(check['iterations']) if _is_default_workload(args, total_rows) else None
    if n1 is not None and 'check' in n1:
        check['matches_n1'] = same(check, n1['check'])
        check['n1'] = n1['check']
        check['matches_n1_how'] = 'against the N = 1 run of the same workload made in this process on rank 0\'s GPU, rtol %g' % CHECK_RTOL
    elif emb is not None:
        check['matches_n1'] = same(check, emb)
        check['matches_n1_how'] = 'against the N = 1 values embedded in bench.py (EMBEDDED_CHECK), rtol %g' % CHECK_RTOL
    else:
        check['matches_n1'] = None
        check['matches_n1_how'] = 'no N = 1 reference for this workload / iteration count'
    if emb is not None:
        check['matches_embedded'] = same(check, emb)
    whole = None
    if world == 1:
        # the same call as a user makes it: `steps` iterations and the final log-likelihood pass (model.py:800-801)
        eng.set_option('kernel_timing', 0)
        tl.keep_kernel_timing = False
        eng.final_lnl()                     # (the first launch of the lnl kernel loads its code object: ~80 ms, once per process)
        tl.max_iter, tl.epsilon = args.steps, 0.0
        w_ms = []
        for _ in range(2):                  # (twice: the first call after the timed region pays ~40 ms of one-off costs; both are reported)
            fence()
            t1 = time.perf_counter()
            tl.em(loglev=logging.DEBUG)
            fence()
            w_ms.append((time.perf_counter() - t1) * 1e3)
        whole = dict(iterations=int(tl.n_iter), ms=w_ms[1], ms_first_call=w_ms[0], lnl=float(tl.lnl),
                     note='tl.em() with max_iter = steps, em_epsilon = 0, incl. the log-likelihood pass after the loop; '
                          'no per-pass HIP events')
    report_pass = None
    if world == 1:
        # the report pass over the final z (conf | exclude | average of output_report, model.py:432-457) on the bench's matrix, after the
        # timed region: HIP events around its dominant kernel (tsem_report_stats), wall clock of the whole call.  Not part of `value`.
        try:
            from telescope_amd._lib import Z_PREV
            eng.set_option('kernel_timing', 1)
            r_ms, r_k = [], []
            for _ in range(4):
                fence()
                t1 = time.perf_counter()
                eng.report_colsums(Z_PREV, 0.9)
                r_ms.append((time.perf_counter() - t1) * 1e3)
                r_k.append(eng.report_stats())
            eng.set_option('kernel_timing', 0)
            st = r_k[-1]
            kms = min(x['kernel_ms'] for x in r_k[1:])
            report_pass = {'kernel': st['kernel'], 'kernel_ms': kms, 'algo_bytes': st['algo_bytes'],
                           'achieved_GBps': st['algo_bytes'] / (kms * 1e-3) / 1e9 if kms > 0 else None,
                           'frac_of_8TBps': st['algo_bytes'] / (kms * 1e-3) / 8e12 if kms > 0 else None,
                           'call_ms': min(r_ms[1:]), 'first_call_ms': r_ms[0], 'rows_left_to_the_exact_kernel': st['deferred_rows'],
                           'algo_bytes_how': '4 B per stored entry (2 B popularity id + 2 B score code) + 8 B row pointer + 4 B output per row',
                           'round5': 'k_report_rows<4,16,false>: 4.6-5.9 ms (profiles/r05_report_kernel_stats.txt)'}
        except Exception as e:   # noqa: BLE001 — an extra measurement must never break the bench line
            report_pass = {'error': repr(e)}
    info = eng.layout_info()
    traffic = _pmc_traffic(total_rows, args, world, info.get('value_bytes', 8)) if info.get('fused') else None
    k_ms = ks['em_ms'] / max(1, ks['em_launches'])
    achieved = ks['algo_bytes_per_pass'] / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    # the nominal AND the measured-stream peak.  A pure streaming read of an 8 GB scratch buffer on this GPU, in this
    # run (tsem_debug_stream_read: 16-byte non-temporal loads, eight in flight per thread, best of three launches)
    try:
        from telescope_amd._lib import stream_read_gbs
        peak_measured = stream_read_gbs(local, 8 << 30, 3)
    except Exception as e:   # noqa: BLE001 — e.g. no room for the scratch buffer beside a very large workload
        peak_measured = None
        print('bench.py: stream probe failed: %r' % (e,), file=sys.stderr)
    out = {
        'metric': 'EM iterations/sec on fragment x locus CSR (nnz/sec alongside)',
        'value': args.steps / elapsed, 'unit': 'iter/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
        'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'nnz_per_sec': nnz_total * args.steps / elapsed,
        'check': check,
        'phase_us': dict(phases, how='HIP events at the phase boundaries of %d iterations AFTER the timed region (rank 0): EM pass kernel(s) | '
                                     'k_colreduce | all-reduce of K+2 doubles | k_update | gap to the next pass' % PHASE_ITERS) if phases else None,
        'n1_reference': n1,
        'speedup_vs_n1': (n1['ms_per_step'] / (elapsed / args.steps * 1e3)) if (n1 and 'ms_per_step' in n1) else None,
        'timed_call': 'TelescopeLikelihood.em(final_lnl=False): chunks of %d iterations per host synchronisation' % EM_CHUNK,
        'whole_em_call': whole,
        'report_pass': report_pass,
        'properties': props,
        'config': {
            'workload': ((args.config_note + ': ') if args.config_note else '') +
                        'synthetic %s fragments x %dk loci, ~%g nnz/row, %s columns, fp64 arithmetic, '
                        'pi_prior=0 theta_prior=200000, em_epsilon=0 (fixed iterations)'
                        % ('%dM' % (total_rows // 1_000_000) if total_rows >= 1_000_000 else '%dk' % (total_rows // 1000),
                           args.cols // 1000, args.nnz_row, args.dist),
            'baseline_config': args.config or (4 if _is_default_workload(args, total_rows) else None),
            'rows': total_rows, 'cols': args.cols, 'nnz': nnz_total, 'dist': args.dist, 'seed': args.seed,
            'parallelism': (('row-sharded x%d, 1 in-library RCCL all-reduce(K+2 f64)/iter' % world) if comm.in_library else
                            ('row-sharded x%d, FALL-BACK transport: torch.distributed all-reduce(K+1 f64) + host round trip per iter' % world))
            if comm is not None else 'single GPU',
            'transport': (comm.describe() + (' — ONE-DEVICE DRY RUN, host-staged' if args.one_device else '')) if comm is not None else None,
            'launcher': 'self (bench.py started the ranks)' if os.environ.get('TSEM_BENCH_SELF_LAUNCHED') else
                        ('torchrun / external' if world > 1 else 'single process'),
            'em_kernel': args.em_kernel, 'layout': info,
            'value_format': 'code16+lut (6 B/nnz stored)' if info.get('value_bytes') == 2 else 'f64 (12 B/nnz stored)', 'setup_s': round(t_setup, 3),
        },
        'roofline': {
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_unit': _pmc_stamp(),
            'peak_nominal': HBM_PEAK_GBS, 'peak_measured': peak_measured,
            'frac_of_measured': (achieved / peak_measured) if peak_measured else None,
            'peak_measured_how': 'pure streaming read of 8 GB in this run (tsem_debug_stream_read), GB/s',
            'kernel': 'EM pass k_em_fused (rank 0 shard)', 'kernel_ms': k_ms, 'kernel_launches_timed': ks['em_launches'],
            'limiter': ('LDS atomics/gathers (2-byte score codes halve the HBM bytes)'
                        if info.get('value_bytes') == 2 else
                        'HBM stream (83-87 % of what a pure streaming read reaches on this part); the LDS work of a step is next (profiles/HISTORY.md 9.2)'),
            'algo_bytes_per_launch': ks['algo_bytes_per_pass'],
        },
    }
    if world == 1 and args.value_format == 'f64' and info.get('fused') and not args.no_alt_layout:
        # the same workload with 2-byte score codes + LDS score table (the library's `auto` layout: half the
        # HBM bytes, the same fp64 arithmetic): reported beside the headline; not part of `value`
        eng.close()
        del tl
        eng2 = Engine(local)
        eng2.set_option('value_format', 0)
        eng2.generate(r0, r1, args.cols, cdf, args.seed, dist_code, args.uniq_frac)
        tl2 = TelescopeLikelihood.from_engine(eng2, Opts(args.steps), None)
        info2 = eng2.layout_info()
        if info2.get('value_bytes') == 2:
            eng2.set_option('kernel_timing', args.kernel_timing)
            tl2.keep_kernel_timing = True
            tl2.max_iter, tl2.epsilon = max(1, args.warmup), 0.0
            tl2.em(loglev=logging.DEBUG, final_lnl=False)
            eng2.kernel_stats(reset=True)
            eng2.synchronize()
            tl2.max_iter = args.steps
            t0 = time.perf_counter()
            tl2.em(loglev=logging.DEBUG, final_lnl=False)
            eng2.synchronize()
            el2 = time.perf_counter() - t0
            ks2 = eng2.kernel_stats()
            k2 = ks2['em_ms'] / max(1, ks2['em_launches'])
            a2 = ks2['algo_bytes_per_pass'] / (k2 * 1e-3) / 1e9
            out['code16_layout'] = {
                'value_format': 'code16+lut (6 B/nnz stored); the library default (`value_format=auto`)',
                'ms_per_step': el2 / args.steps * 1e3, 'nnz_per_sec': nnz_total * args.steps / el2,
                'speedup_vs_f64_layout': (elapsed / args.steps) / (el2 / args.steps),
                'roofline': {'bound': 'hbm', 'achieved': a2, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': a2 / HBM_PEAK_GBS,
                             'kernel_ms': k2, 'algo_bytes_per_launch': ks2['algo_bytes_per_pass'],
                             'traffic': _pmc_traffic(total_rows, args, world, 2),
                             'peak_measured': peak_measured, 'frac_of_measured': (a2 / peak_measured) if peak_measured else None,
                             'limiter': 'LDS (random ds_add_f64 scatter + gathers), not HBM: half the bytes of the headline layout'},
            }
        eng2.close()
        del tl2
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(args, dist_code, cdf)
        out['cpu_baseline'] = {
            'value': cb['nnz_per_sec'] / nnz_total, 'unit': 'iter/s', 'cores': 1, 'kind': 'port',
            'sample': 'oracle/telescope_oracle.py (scipy.sparse operator sequence of the reference, single-threaded like scipy) '
                      'timed for T = %d EM iterations on %s; value = the rate of the LARGEST sample (%d rows x ~%g per row, %d nnz: '
                      '%.3f s/iter = %.3g nnz/s) / workload nnz (linear extrapolation); 1 core used of %d'
                      % (cb['iters'], ' and '.join('%d rows x ~%g per row' % (t['rows'], t['nnz_row']) for t in cb['timings']),
                         cb['best']['rows'], cb['best']['nnz_row'], cb['best']['nnz'], cb['best']['sec_per_iter'],
                         cb['nnz_per_sec'], os.cpu_count()),
            'samples': cb['timings'],
            'nnz_per_sec': cb['nnz_per_sec'],
        }
        out['speedup_vs_cpu'] = out['nnz_per_sec'] / cb['nnz_per_sec']
        fc = cb.get('fused_c')
        if fc and 'nnz_per_sec' in fc:
            out['cpu_baseline_fused_c'] = {
                'value': fc['nnz_per_sec'] / nnz_total, 'unit': 'iter/s', 'cores': fc['cores'], 'kind': 'port',
                'sample': 'oracle/em_fused.c (plain C restatement: one fused OpenMP pass over the CSR rows per iteration, '
                          'thread-private column accumulators) on the first %d rows (%d nnz), all host cores: %.4f s/iter = '
                          '%.3g nnz/s; value = that rate / workload nnz' % (fc['sample_rows'], fc['sample_nnz'],
                                                                             fc['sec_per_iter'], fc['nnz_per_sec']),
                'nnz_per_sec': fc['nnz_per_sec'],
            }
            out['speedup_vs_cpu_fused_c'] = out['nnz_per_sec'] / fc['nnz_per_sec']
        elif fc:
            out['cpu_baseline_fused_c'] = fc
        out['parity_on_sample'] = {k: cb[k] for k in ('lnl_ref', 'lnl_gpu', 'lnl_rel_delta', 'pi_max_rel_delta',
                                                      'final_count_mismatches', 'near_tie_rows', 'final_conf_max_rel_delta',
                                                      'sample_rows', 'iters', 'converged_run')}
    if world == 1 and not args.no_precision_sweep:
        # BASELINE config 3 (10M x 30k x ~40): error of reduced-precision storage / accumulation against fp64
        try:
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import logging
            import precision_sweep
            logging.disable(logging.WARNING)
            out['precision_sweep'] = precision_sweep.sweep(rows=min(10_000_000, args.rows), cols=args.cols,
                                                           nnz_row=args.nnz_row, iters=30)
            logging.disable(logging.NOTSET)
        except Exception as e:   # noqa: BLE001 — never let the extra block break the bench line
            out['precision_sweep'] = dict(error=repr(e))
    if world == 1 and not args.no_reproducible_leg:
        # option `reproducible` (exact, order-independent sums, profiles/HISTORY.md 5.1) on a 10M-row sample of the workload: what it costs
        # per EM pass and whether two independent contexts agree bit for bit (the default mode is timed beside it)
        try:
            out['reproducible_mode'] = reproducible_leg(local, min(10_000_000, args.rows), args, cdf, dist_code)
        except Exception as e:   # noqa: BLE001
            out['reproducible_mode'] = dict(error=repr(e))
    _shutdown(comm)
    try:   # RCCL prints its version banner through C stdio: flush it first so the JSON is the LAST line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:   # noqa: BLE001
        pass
    print(json.dumps(out), flush=True)


def properties_leg(tl, eng, comm, total_rows, nnz_total, args, iters=4):
    """Size-independent properties of a run no CPU oracle can follow (BASELINE config 5: 2e10 stored entries) — the checks of
    tests/test_gpu_round5.py test_half_of_config5_on_one_gpu, collective over the ranks of a row-sharded run.  Every rank calls this;
    the dictionary (all booleans + the numbers behind them) is the same on every rank."""
    import logging
    import math

    def gsum(v):
        return float(comm.sum_array(np.array([float(v)]))[0]) if comm is not None else float(v)

    def gmax(v):
        return float(comm.max_array(np.array([float(v)]))[0]) if comm is not None else float(v)
    p = {}
    cols = args.cols
    eng.set_option('kernel_timing', 0)
    eng.set_params(np.repeat(1. / cols, cols), np.repeat(1. / cols, cols))
    tl.max_iter, tl.epsilon = iters, 0.0
    tl.em(loglev=logging.DEBUG)
    info = eng.layout_info()
    pi_f, theta_f, lnl_f = tl.pi.copy(), tl.theta.copy(), float(tl.lnl)
    p['iterations'] = iters
    p['fallbacks_all_ranks'] = int(gsum(info['fallbacks']))
    p['fused_kernel_on_every_rank'] = gmax(0 if info['fused'] else 1) == 0
    p['no_fallback'] = p['fallbacks_all_ranks'] == 0 and (p['fused_kernel_on_every_rank'] or args.one_device)   # (the one-device dry run starts on the two-pass kernels)
    p['lnl'] = lnl_f
    p['lnl_finite'] = bool(math.isfinite(lnl_f))
    p['pi_sum'], p['theta_sum'] = float(pi_f.sum()), float(theta_f.sum())
    p['parameters_are_distributions'] = bool(abs(pi_f.sum() - 1.0) < 1e-11 and abs(theta_f.sum() - 1.0) < 1e-11 and (pi_f >= 0).all() and (theta_f >= 0).all())
    mem = eng.device_memory()
    p['resident_bytes_per_entry_max_rank'] = gmax(sum(mem['resident'].values()) / max(1, eng.dims()[2]))
    p['all_initial_sum'] = int(tl.reassign_colsums('all', initial=True).sum())
    p['all_counts_every_entry'] = p['all_initial_sum'] == int(nnz_total)
    excl = tl.reassign_colsums('exclude')
    rep = next(iter(tl._report_cache.values()))
    ties = int(gsum(len(rep['rows'])))
    empty = int(total_rows - gsum(info['N_amb'] + info['N_uni']))     # fragments without a stored entry (in no mask)
    p['exclude_sum'], p['tied_rows'] = int(excl.sum()), ties
    p['exclude_plus_ties_is_every_fragment'] = int(excl.sum()) + ties + empty == int(total_rows)
    avg = tl.reassign_colsums('average')
    p['average_sum'] = float(avg.sum())
    p['average_sums_to_fragments'] = bool(abs(avg.sum() - (total_rows - empty)) < 1e-6 * total_rows)
    p['near_tie_rows_all_ranks'] = int(gsum(eng.layout_info().get('near_tie_rows', 0)))
    # the same iterations on the two-pass kernels (the layout is rebuilt with fp64 entries) from the same start
    try:
        eng.fallback_twopass()
        eng.set_params(np.repeat(1. / cols, cols), np.repeat(1. / cols, cols))
        tl.em(loglev=logging.DEBUG)
        p['twopass_pi_max_rel_delta'] = float(np.max(np.abs(tl.pi - pi_f) / np.maximum(pi_f, 1e-300)))
        p['twopass_lnl_rel_delta'] = abs(float(tl.lnl) - lnl_f) / abs(lnl_f)
        p['twopass_agrees'] = bool(np.allclose(tl.pi, pi_f, rtol=1e-10, atol=1e-300) and np.allclose(tl.theta, theta_f, rtol=1e-10, atol=1e-300) and
                                   p['twopass_lnl_rel_delta'] <= 1e-10 and np.array_equal(tl.reassign_colsums('exclude'), excl))
    except Exception as e:   # noqa: BLE001
        p['twopass_agrees'] = False
        p['twopass_error'] = repr(e)
    p['all_hold'] = bool(all(p[k] for k in ('no_fallback', 'lnl_finite', 'parameters_are_distributions', 'all_counts_every_entry',
                                          'exclude_plus_ties_is_every_fragment', 'average_sums_to_fragments', 'twopass_agrees')))
    return p


def reproducible_leg(device, rows, args, cdf, dist_code, iters=10):
    import logging
    from telescope_amd._lib import Engine
    from telescope_amd.likelihood import TelescopeLikelihood
    logging.disable(logging.WARNING)
    res = {}
    try:
        for mode in (0, 1, 1):
            eng = Engine(device)
            eng.set_option('reproducible', mode)
            eng.generate(0, rows, args.cols, cdf, args.seed, dist_code, 0.05)
            tl = TelescopeLikelihood.from_engine(eng, Opts(iters), None)
            eng.em_chunk(2, 0.0, False)
            eng.synchronize()
            t0 = time.perf_counter()
            eng.em_chunk(iters, 0.0, False)
            eng.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / iters
            pi, theta = eng.get_params()
            info = eng.layout_info()
            res.setdefault(mode, []).append((ms, pi.copy(), theta.copy(), info['bin_repeats'], info.get('exact_single', 0)))
            eng.close()
            del tl
    finally:
        logging.disable(logging.NOTSET)
    d, (a, b) = res[0][0], res[1]
    return dict(sample_rows=rows, iterations=iters + 2, ms_per_step_default=d[0], ms_per_step=min(a[0], b[0]),
                cost_vs_default=min(a[0], b[0]) / d[0], repeated_passes=int(a[3]),
                form='one pass, three tables per part' if a[4] else 'two passes (three tables per part do not fit / rows too short for teams of 5-8)',
                two_runs_bit_identical=bool((a[1] == b[1]).all() and (a[2] == b[2]).all()),
                pi_max_rel_delta_vs_default=float(abs(a[1] - d[1]).max() / d[1].max()))


def _is_default_workload(args, total_rows):
    return (total_rows, args.cols, args.nnz_row, args.dist, args.uniq_frac, args.seed, args.value_format) == \
        (50_000_000, 30_000, 40.0, 'zipf', 0.0, 42, 'f64') and args.em_kernel in ('auto', 'fused') and not args.fused_dbg


def _pmc_traffic(total_rows, args, world, value_bytes):
    """HBM bytes per launch (GB) from a separate rocprofv3 --pmc run of this same workload (profiles/)."""
    try:
        for tj in json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))['runs']:
            w = tj['workload']
            if (w['rows'], w['cols'], w['nnz_row'], w['dist'], w['n_gpus'], w['value_bytes']) == \
                    (total_rows, args.cols, args.nnz_row, args.dist, world, value_bytes):
                return tj['traffic_bytes_per_launch'] / 1e9
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return None


def _pmc_stamp():
    """Where `roofline.traffic` comes from — a constant read from profiles/pmc_traffic.json, not something this run measured — with the
    stamp of that measurement and whether the library's sources are still the ones it was made on."""
    try:
        from telescope_amd._lib import sources_fingerprint
        m = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'))).get('measured', {})
        now = sources_fingerprint()
        state = 'the sources of this run are the measured ones' if m.get('sources_sha16') == now else \
            'STALE: the library sources changed since (now %s)' % now
        return 'GB per launch (PMC, profiles/pmc_traffic.json: measured %s at commit %s, sources %s — %s)' % (
            m.get('date', '?'), m.get('commit', '?'), m.get('sources_sha16', '?'), state)
    except Exception:   # noqa: BLE001
        return 'GB per launch (PMC, profiles/pmc_traffic.json)'


def _shutdown(comm):
    if comm is not None:
        import torch.distributed as dist
        try:
            comm.close()                    # the library's RCCL communicator
        except Exception:   # noqa: BLE001
            pass
        try:
            dist.destroy_process_group()
        except Exception:   # noqa: BLE001 — never let teardown hide the result line
            pass


if __name__ == '__main__':
    main()
