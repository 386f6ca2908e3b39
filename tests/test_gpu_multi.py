"""Row-sharded runs on REAL RCCL: one process per GPU, the library's own communicator (`ncclCommInitRank` / `ncclAllReduce` on the
engine's stream, telescope_amd/csrc/tsem_comm.hip) at world size > 1.

Every test here needs at least two GPUs and skips on the one-GPU boxes of the development pool; the first box with two or more
runs them by itself (`pytest -m gpu`).  Nothing below sets TSEM_ONE_DEVICE / TSEM_GLOO_HOST_STAGED / TSEM_BACKEND: these are the
commands the driver's scaling run and a user's `torchrun ... -m telescope_amd resume` execute, not the dry-run transport of
tests/test_gpu_round3.py::test_bench_two_ranks_dry_run_on_one_gpu.

Checked against the single-process run of the same workload: bench.py's `check` block (pi, theta after warm-up + steps, folded to
three numbers; the rows are generated from their GLOBAL index, so N ranks hold the same matrix) to 1e-11, and the report files of
`telescope resume` against the files the reference wrote (tests/golden/resume_*).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
pytestmark = pytest.mark.gpu

_COMMON = ['--rows', '4000000', '--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--no-alt-layout', '--no-precision-sweep',
           '--no-reproducible-leg', '--uniq-frac', '0.05']
_LINES = {}


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:   # noqa: BLE001
        return 0


@pytest.fixture(scope='module')
def two_gpus(gpu_device):
    n = _n_gpus()
    if n < 2:
        pytest.skip('needs >= 2 GPUs (this box has %d): real RCCL at world size > 1' % n)
    return n


def _clean_env():
    env = dict(os.environ)
    for k in ('TSEM_ONE_DEVICE', 'TSEM_GLOO_HOST_STAGED', 'TSEM_BACKEND', 'TSEM_TORCH_COLLECTIVES', 'RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return env


def _port():
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    return port


def _torchrun(n):
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
            '--master-port', str(_port())]


def _bench(key, prefix, extra):
    if key not in _LINES:
        r = subprocess.run(prefix + [os.path.join(ROOT, 'bench.py')] + extra + _COMMON, cwd=ROOT, capture_output=True, text=True,
                           timeout=1200, env=_clean_env())
        assert r.returncode == 0, r.stderr[-4000:]
        _LINES[key] = (json.loads(r.stdout.strip().splitlines()[-1]), r.stderr)
    return _LINES[key]


def _same_check(many, one, iterations):
    assert many['check']['iterations'] == one['check']['iterations'] == iterations
    for key in ('pi_sum', 'pi_weighted', 'theta_weighted'):
        assert abs(many['check'][key] - one['check'][key]) <= 1e-11 * abs(one['check'][key]), (key, many['check'], one['check'])


@pytest.mark.parametrize('launcher', ['self', 'torchrun'])
def test_bench_two_gpus_on_rccl(two_gpus, launcher):
    """`python bench.py --gpus 2` (bench.py starts its ranks) and the driver's form `python -m torch.distributed.run
    --nproc-per-node 2 ... bench.py --gpus 2`: the in-library communicator at two ranks, one ncclAllReduce(K + 2 doubles) per
    iteration between the pass and the update.  Same parameters as one GPU; the line names the RCCL copy that served it."""
    one, _ = _bench('one', [sys.executable], ['--gpus', '1'])
    if launcher == 'self':
        two, _ = _bench('self', [sys.executable], ['--gpus', '2'])
    else:
        two, _ = _bench('torchrun', _torchrun(2), ['--gpus', '2'])
    assert two['n_gpus'] == 2 and one['n_gpus'] == 1 and two['config']['nnz'] == one['config']['nnz']
    tr = two['config']['transport']
    assert 'in-library' in tr and 'librccl' in tr and 'DRY RUN' not in tr, tr
    assert 'in-library RCCL all-reduce' in two['config']['parallelism']
    assert ('self' in two['config']['launcher']) == (launcher == 'self')
    _same_check(two, one, 8)
    assert two['value'] > 0 and two['config']['layout']['fallbacks'] == 0
    _self_validating(two, 2)


def _self_validating(line, n):
    """What makes a scaling run diagnose itself: the line's own N = 1 reference (the whole problem on rank 0's GPU in
    the same process) agrees with the N-rank parameters, the speed-up is quoted against it, the transport is the library's RCCL (a
    silent fall-back to torch collectives turns this red, not just slow), and every phase of an iteration was timed."""
    assert 'in-library RCCL all-reduce' in line['config']['parallelism'], line['config']['parallelism']
    assert line['check']['matches_n1'] is True, line['check']
    assert line['n1_reference'] and line['n1_reference']['check']['iterations'] == line['check']['iterations']
    assert line['speedup_vs_n1'] and line['speedup_vs_n1'] > 0.5, line['speedup_vs_n1']     # (a 4M-row test matrix: start-up bound)
    ph = line['phase_us']
    assert ph and ph['iterations'] >= 4
    for key in ('pass', 'colreduce', 'allreduce', 'update', 'gaps', 'iteration'):
        assert ph[key] >= 0.0, (key, ph)
    assert ph['pass'] > 0 and ph['allreduce'] > 0 and ph['update'] > 0, ph
    assert abs(ph['pass'] + ph['colreduce'] + ph['allreduce'] + ph['update'] - ph['iteration']) <= 0.05 * ph['iteration'] + 5.0, ph
    assert line['roofline']['frac'] > 0 and 'rank 0' in line['roofline']['kernel']


def test_bench_on_every_gpu_of_the_box(two_gpus):
    """N = all GPUs (up to 8): what `SCALE_rNN.json`'s last point runs, on a small matrix."""
    n = min(8, two_gpus)
    if n == 2:
        pytest.skip('covered by test_bench_two_gpus_on_rccl')
    one, _ = _bench('one', [sys.executable], ['--gpus', '1'])
    many, _ = _bench('all', _torchrun(n), ['--gpus', str(n)])
    assert many['n_gpus'] == n and many['config']['nnz'] == one['config']['nnz']
    _same_check(many, one, 8)
    _self_validating(many, n)


def test_time_out_on_one_rank_is_survived_by_all(two_gpus):
    """Rank 1's persistent kernel reports a hand-off time-out in its first pass (`--fail-rank 1`: fused_dbg bit 5, what the
    watchdog leaves behind): slot K of the all-reduced sums tells EVERY rank, nobody commits, rank 1 alone rebuilds its layout for
    the two-pass kernels, every rank redoes the iteration — no rank is left waiting in a collective, and the parameters are the
    single-GPU run's."""
    one, _ = _bench('one', [sys.executable], ['--gpus', '1'])
    two, err = _bench('fail1', _torchrun(2), ['--gpus', '2', '--fail-rank', '1'])
    _same_check(two, one, 8)
    assert 'continuing with the two-pass kernels' in err           # rank 1's fall-back notice (stderr of the rank processes)
    assert two['config']['layout']['fallbacks'] == 0                 # rank 0 (whose layout the line reports) kept the fused kernel


@pytest.mark.parametrize('mode', ['choose', 'conf'])
def test_resume_row_sharded_on_two_gpus(two_gpus, tmp_path, mode):
    """`python -m torch.distributed.run --nproc-per-node 2 -m telescope_amd resume <checkpoint>` with one GPU per rank: the
    reference's report files (rank 0 draws the picks of `choose` and writes them), one copy of the progress lines."""
    cmd = _torchrun(2) + ['-m', 'telescope_amd', 'resume', os.path.join(GOLD, 'resume_checkpoint.npz'), '--outdir', str(tmp_path),
                          '--exp_tag', 'run', '--reassign_mode', mode]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=_clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stderr.count('EM converged after 16 iterations.') == 1
    assert 'Final log-likelihood: 95252.596293.' in r.stderr and 'Row-sharded over 2 ranks' in r.stderr
    for suffix in ('run_stats.tsv', 'TE_counts.tsv'):
        got = open(os.path.join(str(tmp_path), 'run-' + suffix)).read()
        want = open(os.path.join(GOLD, 'resume_%s-%s' % (mode, suffix))).read()
        if got != want:   # rows of equal final_prop may come in any order (the reference sorts with an unstable sort, model.py:449)
            assert sorted(got.splitlines()) == sorted(want.splitlines()), suffix


def test_use_likelihood_on_two_gpus(two_gpus, tmp_path):
    """`--use_likelihood` row-sharded: the lnl of every iteration is all-reduced too and decides convergence on the device; the
    reference stops after 25 iterations at 95252.596614."""
    cmd = _torchrun(2) + ['-m', 'telescope_amd', 'resume', os.path.join(GOLD, 'resume_checkpoint.npz'), '--outdir', str(tmp_path),
                          '--exp_tag', 'run', '--use_likelihood']
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=_clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'EM converged after 25 iterations.' in r.stderr and 'Final log-likelihood: 95252.596614.' in r.stderr


# ---- BASELINE configs 4 and 5 at their stated sizes: one command each the day an 8-GPU node runs `pytest -m gpu` --------

@pytest.fixture(scope='module')
def eight_gpus(gpu_device):
    n = _n_gpus()
    if n < 8:
        pytest.skip('needs 8 GPUs (this box has %d): BASELINE configs 4 / 5 at full size' % n)
    import torch
    free = min(torch.cuda.mem_get_info(d)[0] for d in range(8))
    if free < 200 * 2 ** 30:
        pytest.skip('needs 200 GB of free HBM on each of the 8 GPUs (least: %.0f GB)' % (free / 2 ** 30))
    return n


def _bench_config(extra, timeout=3000):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + extra + ['--no-cpu-baseline', '--no-alt-layout', '--no-precision-sweep',
                                                                                 '--no-reproducible-leg'],
                       cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=_clean_env())
    assert r.returncode == 0, r.stderr[-4000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_config5_at_full_size_on_eight_gpus(eight_gpus):
    """`python bench.py --config 5 --gpus 8`: 200M fragments x 50k loci x ~100 per row = 2e10 stored entries, 25M rows per rank, score
    codes, one in-library RCCL all-reduce per iteration.  No CPU oracle follows this size: the line's `properties` block holds the
    checks of test_half_of_config5_on_one_gpu made collectively — no fall-back on any rank, pi / theta distributions, `all` counts every
    stored entry, exclude + tied rows = every fragment, `average` sums to the fragments, and the same four iterations on the two-pass
    kernels (every rank's layout rebuilt with fp64 entries) agree to 1e-10."""
    line = _bench_config(['--config', '5', '--gpus', '8', '--steps', '6', '--warmup', '2'])
    assert line['n_gpus'] == 8 and line['scaling'] == 'weak' and line['config']['baseline_config'] == 5
    assert 'BASELINE config 5' in line['config']['workload'] and 'AT ' not in line['config']['workload']
    assert line['config']['rows'] == 200_000_000 and abs(line['config']['nnz'] - 2.0e10) < 2e8
    assert 'in-library RCCL all-reduce' in line['config']['parallelism'], line['config']['parallelism']
    p = line['properties']
    assert p['all_hold'] is True and p['fallbacks_all_ranks'] == 0 and p['fused_kernel_on_every_rank'], p
    assert p['resident_bytes_per_entry_max_rank'] < 11.5, p
    assert line['config']['layout']['value_bytes'] == 2 and line['roofline']['frac'] > 0.2


def test_config4_at_full_size_on_eight_gpus(eight_gpus):
    """`python bench.py --config 4 --gpus 8` — the driver's scaling run at its last point: 50M x 30k x ~40 row-sharded over 8 ranks,
    checked against the N = 1 run of the whole problem made in the same process on rank 0's GPU (`check.matches_n1`, rtol 1e-11), with
    the phases of an iteration timed and the speed-up quoted (north_star: >= 6 x)."""
    line = _bench_config(['--config', '4', '--gpus', '8', '--steps', '20', '--warmup', '3'])
    assert line['n_gpus'] == 8 and line['scaling'] == 'strong' and line['config']['baseline_config'] == 4
    assert line['config']['rows'] == 50_000_000 and 'BASELINE config 4' in line['config']['workload']
    _self_validating(line, 8)
    assert line['check'].get('matches_embedded', True) is True
    assert line['config']['layout']['fallbacks'] == 0
    print('config 4 on 8 GPUs: %.3f ms per iteration, %.2f x the N = 1 run' % (line['ms_per_step'], line['speedup_vs_n1']))
