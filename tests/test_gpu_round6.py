"""Round 6, `-m gpu`: near-ties of the final z decided with the reference's own row sum (np.add.reduceat order), the CSR column ids
staying dropped through a report, the C host of the boundary, the config presets of bench.py.  Everything goes through the C ABI."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLD, ROOT, Opts

pytestmark = pytest.mark.gpu


def _near_tie_matrix(seed, n=6000, k=400, max_len=40, long_rows=0, scores=(3, 4)):
    """A matrix whose rows are full of near-ties once the parameters below are set: columns come in PAIRS (2j, 2j + 1) whose
    pi * theta differ by one to three ulp, a row takes both columns of a pair with the same score, so its two largest z values are
    a few ulp apart — whether they round together hangs on the last bit of 1 / rowsum, i.e. on the ORDER the row is added in."""
    rng = np.random.RandomState(seed)
    lens = rng.randint(1, max_len // 2 + 1, n) * 2
    lens[rng.rand(n) < 0.1] = 1
    if long_rows:
        lens[rng.choice(n, long_rows, replace=False)] = rng.choice([130, 258, 300, 398], long_rows)    # beyond the streaming kernel's 256 entries too
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.empty(indptr[-1], np.int32)
    data = np.empty(indptr[-1], np.uint16)
    for i, l in enumerate(lens):
        s = indptr[i]
        if l == 1:
            indices[s] = rng.randint(k); data[s] = rng.choice(scores)
            continue
        pairs = np.sort(rng.choice(k // 2, l // 2, replace=False))
        indices[s:s + l] = np.repeat(2 * pairs, 2) + np.tile([0, 1], l // 2)
        data[s:s + l] = np.repeat(rng.choice(scores, l // 2), 2)
    raw = sp.csr_matrix((data, indices, indptr), shape=(n, k))
    pi = rng.dirichlet(np.full(k, 2.0))
    theta = rng.dirichlet(np.full(k, 2.0))
    for j in range(0, k, 2):                                # pi * theta of a pair: equal, or one to three ulp apart
        theta[j + 1] = theta[j]
        p = pi[j]
        for _ in range(int(rng.randint(0, 4))):
            p = np.nextafter(p, 1.0)
        pi[j + 1] = p
    return raw, pi, theta


@pytest.mark.parametrize('seed,kw', [(1, {}), (2, dict(max_len=8)), (3, dict(long_rows=40, n=3000)), (4, dict(scores=(1, 2), k=64, max_len=60)),
                                     (5, dict(max_len=250, n=1500, k=600))])
@pytest.mark.parametrize('report_kernel', [1, 0])
def test_near_ties_are_decided_with_the_references_row_sum(gpu_device, seed, kw, report_kernel):
    """with the SAME parameters on both sides every integer output of the final z equals the oracle's bit for bit —
    column sums of exclude / choose / unique / all, best-hit counts, the masks — on matrices built so that thousands of rows sit on
    near-ties (two z values one to three ulp apart).  The streaming report kernel defers such rows to k_report_slow, the generic row
    pass to its FIX launch; both redo the row sum in np.add.reduceat's order (tsem_npsum.h) and count the rows they redid."""
    from oracle.telescope_oracle import OracleModel, binmax_rows
    from telescope_amd import _lib
    from telescope_amd.likelihood import TelescopeLikelihood, score_lut
    raw, pi, theta = _near_tie_matrix(seed, **kw)
    n, k = raw.shape
    eng = _lib.Engine(0)
    eng.set_option('report_kernel', report_kernel)
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), k, score_lut(int(raw.data.max())))
    tl = TelescopeLikelihood.from_engine(eng, Opts(max_iter=1, em_epsilon=0.0))
    tl._raw = raw
    eng.set_params(pi, theta)
    om = OracleModel(raw, 0, 200000)
    om.z = om.estep(pi, theta)
    zo = sp.csr_matrix(om.z)
    # a threshold that IS a z value of many rows (conf compares with >=): the largest z of some row
    zmax = zo.max(1).toarray().ravel()
    thresh = float(np.sort(zmax[zmax > 0.5])[len(zmax[zmax > 0.5]) // 2]) if np.any(zmax > 0.5) else 0.9
    for th in (0.9, thresh):
        sums, tie_rows, tie_cnt = eng.report_colsums(_lib.Z_CUR, th)
        want_ex = np.asarray(om.reassign('exclude', th).sum(0)).ravel()
        assert np.array_equal(sums['exclude'].astype(np.int64), np.rint(want_ex).astype(np.int64)), ('exclude', th)
        assert np.allclose(sums['average'], np.asarray(om.reassign('average', th).sum(0)).ravel(), rtol=1e-12, atol=1e-9)
        assert np.allclose(sums['conf'], np.asarray(om.reassign('conf', th).sum(0)).ravel(), rtol=1e-12, atol=1e-9), ('conf', th)
        nb = np.diff(binmax_rows(zo).indptr)
        assert np.array_equal(tie_rows, np.flatnonzero(nb > 1)) and np.array_equal(tie_cnt, nb[nb > 1])
    assert np.array_equal(eng.best_counts(_lib.Z_CUR), np.diff(binmax_rows(zo).indptr))
    for m in ('exclude', 'average', 'conf', 'all', 'unique'):
        cs, mask = eng.reassign(m, thresh, _lib.Z_CUR, want_mask=True)
        mo = sp.csr_matrix(om.reassign(m, thresh)).astype(np.float64)
        dense_mask = sp.csr_matrix((mask, raw.indices.copy(), raw.indptr.copy()), shape=raw.shape)   # (eliminate_zeros works in place)
        dense_mask.eliminate_zeros()
        d = (dense_mask - mo)
        assert abs(d).max() <= 1e-12 if d.nnz else True, (m, abs(d).max())
        assert np.allclose(cs, np.asarray(mo.sum(0)).ravel(), rtol=1e-12, atol=1e-9), m
    redone = eng.layout_info()['near_tie_rows']
    assert redone > n // 20, redone                         # the matrix IS full of them
    eng.close()


def test_no_near_tie_rows_on_the_bundled_matrix(gpu_device):
    """The other side of the band: ordinary data has no row inside it — the bundled `telescope test` matrix through em() and all
    reports redoes nothing (and stays bit-exact against its golden, tests/test_gpu_parity.py)."""
    from telescope_amd.likelihood import TelescopeLikelihood
    f = np.load(os.path.join(GOLD, 'bundled_raw_scores.npz'))
    raw = sp.csr_matrix((f['data'], f['indices'], f['indptr']), shape=tuple(f['shape']))
    tl = TelescopeLikelihood(raw, Opts(), device=0)
    tl.em()
    for m in ('exclude', 'average', 'conf', 'all', 'unique'):
        tl.reassign_colsums(m)
        sp.csr_matrix(tl.reassign(m))
    assert tl._eng.layout_info()['near_tie_rows'] == 0


# ---- BASELINE configs 4 and 5 through bench.py's presets: the plumbing, at 1/100 of the rows, two ranks sharing this GPU (VERDICT r5 #2c) ----

def _bench_json(extra, timeout=1500):
    env = dict(os.environ)
    for k in ('TSEM_ONE_DEVICE', 'TSEM_GLOO_HOST_STAGED', 'TSEM_BACKEND', 'TSEM_TORCH_COLLECTIVES', 'RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + extra + ['--no-cpu-baseline', '--no-alt-layout', '--no-precision-sweep',
                                                                                 '--no-reproducible-leg', '--steps', '4', '--warmup', '1'],
                       cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    import json
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_config5_preset_dry_run_on_one_gpu(gpu_device):
    """`bench.py --config 5 --config-scale 0.01 --gpus 2 --one-device`: the preset's geometry (50k loci, ~100 per row, 250k rows per rank,
    weak scaling), the collective property checks and the result line — what the 8-GPU test of tests/test_gpu_multi.py runs at full size —
    on two rank processes sharing this GPU; the parameters equal the single-process run of the same 500k rows, and so do its properties."""
    two = _bench_json(['--config', '5', '--config-scale', '0.01', '--gpus', '2', '--one-device'])
    one = _bench_json(['--gpus', '1', '--rows', '500000', '--cols', '50000', '--nnz-row', '100', '--value-format', 'auto', '--properties'])
    assert two['n_gpus'] == 2 and two['scaling'] == 'weak' and two['config']['baseline_config'] == 5
    assert 'BASELINE config 5' in two['config']['workload'] and 'AT 0.01 OF ITS ROWS' in two['config']['workload']
    assert two['config']['rows'] == 500_000 == one['config']['rows'] and two['config']['nnz'] == one['config']['nnz']
    for key in ('pi_sum', 'pi_weighted', 'theta_weighted'):
        assert abs(two['check'][key] - one['check'][key]) <= 1e-11 * abs(one['check'][key]), key
    for line in (one, two):
        p = line['properties']
        assert p['all_hold'] is True and p['fallbacks_all_ranks'] == 0, p
        assert p['all_initial_sum'] == line['config']['nnz']
    assert one['properties']['fused_kernel_on_every_rank'] and one['properties']['twopass_pi_max_rel_delta'] < 1e-10
    assert two['properties']['exclude_sum'] == one['properties']['exclude_sum'] and two['properties']['tied_rows'] == one['properties']['tied_rows']
    assert abs(two['properties']['lnl'] - one['properties']['lnl']) <= 1e-10 * abs(one['properties']['lnl'])


def test_config4_preset_dry_run_on_one_gpu(gpu_device):
    """`bench.py --config 4 --config-scale 0.01 --gpus 2 --one-device`: the strong-scaling preset with its own N = 1 reference."""
    two = _bench_json(['--config', '4', '--config-scale', '0.01', '--gpus', '2', '--one-device'])
    assert two['n_gpus'] == 2 and two['scaling'] == 'strong' and two['config']['baseline_config'] == 4 and two['config']['rows'] == 500_000
    assert two['check']['matches_n1'] is True and two['n1_reference']['check']['iterations'] == 5


def test_config5_half_through_the_preset(gpu_device):
    """`bench.py --config 5 --gpus 1`: the 1e10-entry half that fits one GPU, with the property checks in the line (skips below 200 GB free)."""
    from telescope_amd import _lib
    free, total = _lib.device_memory(0)
    if free < 200 * 2 ** 30:
        pytest.skip('needs 200 GB of free HBM, %.0f GB free of %.0f' % (free / 2 ** 30, total / 2 ** 30))
    line = _bench_json(['--config', '5', '--gpus', '1'], timeout=2400)
    assert line['config']['rows'] == 100_000_000 and abs(line['config']['nnz'] - 1.0e10) < 1e8 and 'the half that fits one GPU' in line['config']['workload']
    p = line['properties']
    assert p['all_hold'] is True and p['fused_kernel_on_every_rank'] and p['resident_bytes_per_entry_max_rank'] < 11.5, p


# ---- a C host of the boundary, no Python in the process --------------------------------------------------------------------

def _build_c_host(tmp_path):
    exe = str(tmp_path / 'run_bundled')
    lib_dir = os.path.join(ROOT, 'telescope_amd')
    subprocess.run(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-O2', '-I' + os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'tests', 'c_host', 'run_bundled.c'), '-o', exe, '-L' + lib_dir, '-ltelescope_em',
                    '-Wl,-rpath,' + lib_dir, '-Wl,-rpath-link,/opt/rocm/lib'], check=True)
    return exe


def _parse_c_host(stdout):
    kv = {}
    for ln in stdout.splitlines():
        parts = ln.split()
        if parts:
            kv[parts[0]] = parts[1:]
    return kv


@pytest.mark.parametrize('table', ['numpy', 'libm'])
def test_c_host_runs_the_bundled_matrix_without_python(gpu_device, tmp_path, table):
    """tests/c_host/run_bundled.c — C99, `#include "telescope_em.h"`, linked against libtelescope_em.so — loads the bundled matrix
    from a flat file and runs create -> load_scores -> max_score -> set_lut -> rowstats -> set_model -> em_run -> report_colsums in a
    process that never loaded Python, torch or numpy: 16 iterations, lnl 95252.596293, the per-locus `exclude` counts of the golden
    case.  With the numpy score table of the fixture the counts are the reference's bit for bit and lnl agrees to 1e-12; with
    tsem_score_lut (libm's expm1: 1 ulp apart from numpy's in ~10 % of the entries, telescope_em.h) to 1e-9."""
    from conftest import load_case
    exe = _build_c_host(tmp_path)
    env = {k: v for k, v in os.environ.items() if not k.startswith('PYTHON')}
    r = subprocess.run([exe, os.path.join(GOLD, 'bundled_flat.bin')] + (['libm'] if table == 'libm' else []), capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    kv = _parse_c_host(r.stdout)
    c = load_case('bundled')
    assert int(kv['iterations'][0]) == int(c['n_iter']) == 16 and int(kv['iterations'][2]) == 1
    lnl = float(kv['lnl'][0])
    assert abs(lnl - 95252.596293) < 1e-5
    assert abs(lnl - float(c['lnl'])) <= (1e-12 if table == 'numpy' else 1e-9) * abs(float(c['lnl'])), (lnl, float(c['lnl']))
    assert np.array_equal(np.array(kv['exclude'], dtype=np.int64), np.asarray(c['ra_exclude_0_colsum']).astype(np.int64))
    assert int(kv['ties'][2]) == 0                                   # near_tie_rows: none on ordinary data
    # the process really had no Python in it: its only link-time dependencies are the engine and the C library
    ldd = subprocess.run(['ldd', exe], capture_output=True, text=True).stdout
    assert 'libtelescope_em.so' in ldd and 'python' not in ldd.lower() and 'torch' not in ldd.lower()


# ---- the packed report kernel (tsem_report_pack.h): rows of every shape against the capacity kernel, the generic pass and the oracle ----

def _shape_matrix(seed, n, k, kind):
    rng = np.random.RandomState(seed)
    if kind == 'short':            # 0 .. 9 entries, many empty and single-entry rows: chunks of 64 one-lane rows
        lens = rng.randint(0, 10, n)
    elif kind == 'mixed':          # around 40 with a tail up to 700: rows of 1 .. 64 lanes, and rows beyond the kernel's 512 entries
        lens = rng.poisson(40, n)
        big = rng.rand(n) < 0.02
        lens[big] = rng.randint(400, min(700, k) + 1, int(big.sum()))
        lens[rng.rand(n) < 0.05] = 1
        lens[rng.rand(n) < 0.01] = 0
    elif kind == 'lane_edges':     # lengths at the lane boundaries: 7, 8, 9, 15, 16, 17, .. 512, 513
        base = np.array([7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513])
        lens = base[rng.randint(0, len(base), n)]
    else:
        raise ValueError(kind)
    lens = np.minimum(lens, k)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = np.empty(indptr[-1], np.int32)
    for i, l in enumerate(lens):
        if l:
            # hot-locus skew, so that both the LDS table and the L2 tail of pi * theta are hit
            c = np.unique((k * rng.rand(3 * l + 8) ** 3).astype(np.int64))
            while len(c) < l:
                c = np.unique(np.concatenate([c, rng.randint(0, k, l)]))
            indices[indptr[i]:indptr[i + 1]] = np.sort(rng.choice(c, l, replace=False))
    data = rng.randint(1, 6, indptr[-1]).astype(np.uint16)     # five distinct scores: exact ties are common
    return sp.csr_matrix((data, indices, indptr), shape=(n, k))


@pytest.mark.parametrize('kind,n,k', [('short', 30000, 700), ('mixed', 6000, 20000), ('lane_edges', 1500, 40000), ('mixed', 4000, 900)])
def test_packed_report_kernel_on_rows_of_every_shape(gpu_device, kind, n, k):
    """k_report_pack32 (the default for the final z at conf_prob > 0.51; 8 and 16 entries per lane: report_dbg = 128 / 256) against the
    capacity kernel it replaces (report_dbg = 8), the generic
    row pass (report_kernel = 0) and the oracle: column sums of conf / exclude / average, the tied rows and their best-hit counts —
    integer outputs equal bit for bit — on empty rows, single-entry rows, rows that end exactly on a lane, rows of 64 lanes, rows longer
    than the kernel takes (-> k_report_slow), K above and below the LDS table, exact ties (five distinct scores)."""
    from oracle.telescope_oracle import OracleModel, binmax_rows
    from telescope_amd import _lib
    from telescope_amd.likelihood import TelescopeLikelihood, score_lut
    raw = _shape_matrix(11 + n, n, k, kind)
    eng = _lib.Engine(0)
    eng.load_scores(raw.indptr, raw.indices, raw.data.astype(np.uint16), k, score_lut(int(raw.data.max())))
    tl = TelescopeLikelihood.from_engine(eng, Opts(max_iter=3, em_epsilon=0.0))
    tl._raw = raw
    tl.em()
    out = {}
    for name, kern, dbg in (('packed', 1, 0), ('packed8', 1, 128), ('packed16', 1, 256), ('capacity', 1, 8), ('generic', 0, 0)):
        eng.set_option('report_kernel', kern); eng.set_option('report_dbg', dbg)
        for th in (0.9, 0.6):
            sums, r, c = eng.report_colsums(_lib.Z_PREV, th)
            out[(name, th)] = (sums['exclude'].copy(), sums['average'].copy(), sums['conf'].copy(), r.copy(), c.copy())
    eng.set_option('report_kernel', 1); eng.set_option('report_dbg', 0)
    om = OracleModel(raw, 0, 200000)
    om.em(0.0, 3)
    zo = sp.csr_matrix(om.z)
    nb = np.diff(binmax_rows(zo).indptr)
    for th in (0.9, 0.6):
        a = out[('packed', th)]
        for other in ('packed8', 'packed16', 'capacity', 'generic'):
            b = out[(other, th)]
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]), (other, th)
            assert np.allclose(a[1], b[1], rtol=1e-12, atol=1e-9) and np.allclose(a[2], b[2], rtol=1e-12, atol=1e-9), (other, th)
        # against the oracle (its z comes from its own parameters: equal to ~1e-13; a flip would need a near-tie, and those are redone)
        want = np.rint(np.asarray(om.reassign('exclude', th).sum(0)).ravel()).astype(np.int64)
        assert np.array_equal(a[0].astype(np.int64), want), th
        assert np.allclose(a[2], np.asarray(om.reassign('conf', th).sum(0)).ravel(), rtol=1e-9, atol=1e-9), th
        assert np.array_equal(a[3], np.flatnonzero(nb > 1)) and np.array_equal(a[4], nb[nb > 1]), th
    eng.close()


def test_report_stats_name_the_kernel_that_ran(gpu_device):
    """tsem_report_stats (bench.py's `report_pass`): the final z at conf_prob 0.9 runs the packed fp32 filter and leaves only a handful of
    rows to the exact row kernel; report_dbg = 8 the capacity kernel; the initial z without a conf column the score-code kernel; with
    `kernel_timing` on, the pass's HIP-event time is there; the column sums do not depend on which kernel ran."""
    from telescope_amd import _lib
    from test_gpu_parity import _synthetic_tl
    tl = _synthetic_tl(400_000, 30_000, 24, 'zipf', uniq=0.05, opts=Opts(max_iter=4, em_epsilon=0.0))
    tl.em()
    eng = tl._eng
    eng.set_option('kernel_timing', 1)
    sums, r, c = eng.report_colsums(_lib.Z_PREV, 0.9)
    st = eng.report_stats()
    assert st['kernel'] == 'k_report_pack32' and st['kernel_ms'] > 0 and st['algo_bytes'] == 4 * eng.dims()[2] + 12 * eng.dims()[0]
    assert 0 <= st['deferred_rows'] < 4000, st                      # ~1e-4 of the rows sit inside the filter's margins
    eng.set_option('report_dbg', 8)
    sums8, r8, c8 = eng.report_colsums(_lib.Z_PREV, 0.9)
    assert eng.report_stats()['kernel'] == 'k_report_rows'
    eng.set_option('report_dbg', 0)
    assert np.array_equal(sums['exclude'], sums8['exclude']) and np.array_equal(r, r8) and np.array_equal(c, c8)
    assert np.allclose(sums['conf'], sums8['conf'], rtol=1e-12, atol=1e-9) and np.allclose(sums['average'], sums8['average'], rtol=1e-12, atol=1e-9)
    eng.report_colsums(_lib.Z_INITIAL, -1.0)
    assert eng.report_stats()['kernel'] == 'k_report_init_codes'
    eng.set_option('kernel_timing', 0)
    eng.report_colsums(_lib.Z_PREV, 0.9)
    assert eng.report_stats()['kernel_ms'] == 0.0
