"""Host-side logic that needs no GPU: the C-ABI library loads and exports every
declared symbol, the product path fails loudly without a device, the synthetic
generator spec, row sharding, the Q lookup table."""
import ctypes
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import ROOT, Opts, case_matrix, load_case
from telescope_amd import _lib, synthetic
from telescope_amd.distributed import shard_bounds
from telescope_amd.likelihood import TelescopeLikelihood, score_lut


@pytest.fixture(scope='module')
def built():
    _lib.build_library()
    return _lib.lib()


def test_library_exports_every_declared_symbol(built):
    names = _lib.exported_symbols()
    assert len(names) >= 30 and 'tsem_em_pass' in names and 'tsem_reassign' in names
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), n


def test_header_cites_reference_interfaces():
    hdr = open(os.path.join(ROOT, 'include', 'telescope_em.h')).read()
    for cite in ('model.py:702-722', 'model.py:724-742', 'model.py:744-760', 'model.py:762-806',
                 'model.py:808-865', 'sparse_plus.py'):
        assert cite in hdr


def has_gpu():
    try:
        _lib.Engine(0).close()
        return True
    except _lib.EngineError:
        return False


@pytest.mark.skipif(has_gpu(), reason='a GPU is present')
def test_product_path_fails_loudly_without_gpu(built):
    c = load_case('tiny_ties')
    with pytest.raises(_lib.EngineError, match='no CPU fallback'):
        TelescopeLikelihood(case_matrix(c), Opts(c))
    with pytest.raises(_lib.EngineError):
        _lib.csr_norm_rows(np.array([0, 1]), np.array([1.0]))


@pytest.mark.skipif(has_gpu(), reason='a GPU is present')
def test_bench_self_launches_its_ranks_and_fails_loudly_without_gpu(built):
    """`python bench.py --gpus 2` with no launcher around it starts its two ranks itself (it used to exit
    with "must be launched with torch.distributed.run"); here, without a GPU, BOTH ranks must get as far as the
    engine and die of "no CPU fallback", and the parent must pass the failure on."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert 'must be launched' not in r.stderr + r.stdout
    assert r.stderr.count('no CPU fallback') == 2, r.stderr[-2000:]


def test_rccl_is_resolved_at_run_time_not_linked(built):
    """One RCCL per process, chosen deliberately (the copy torch already mapped, else the loader's): the shared
    library must not carry a DT_NEEDED librccl, and must say which copy it resolved."""
    import subprocess
    out = subprocess.run(['readelf', '-d', _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert 'NEEDED' in out and 'librccl' not in out
    info = _lib.comm_library_info()
    assert info.startswith('rccl ') and ('librccl' in info or 'unavailable' in info), info


def test_legacy_randint_is_numpys_stream(built):
    """`choose` draws in numpy's legacy global stream (sparse_plus.py:140-154).  The library's C loop must return
    np.random.randint(0, counts) element for element and leave the global state exactly where numpy leaves it — for any
    position in the 624-word block, counts of 1 (nothing consumed), powers of two and the rejection cases."""
    rs = np.random.RandomState(7)
    for trial, n in enumerate((1, 5, 623, 624, 625, 5000, 200_000)):
        counts = rs.randint(1, [2, 3, 4, 5, 9, 17, 255, 70000][trial % 8] + 1, size=n).astype(np.int32)
        counts[::7] = 1
        np.random.seed(1000 + trial)
        np.random.random_sample(trial * 37)                # start somewhere inside a block (and leave a cached state alone)
        np.random.standard_normal(trial % 2)               # odd trials leave a cached Gaussian behind
        s0 = np.random.get_state()
        want = np.random.randint(0, counts)
        s_want = np.random.get_state()
        np.random.set_state(s0)
        got = _lib.legacy_randint(counts)
        s_got = np.random.get_state()
        assert np.array_equal(got, want)
        assert s_got[0] == s_want[0] and np.array_equal(s_got[1], s_want[1]) and s_got[2:] == s_want[2:]
        assert np.random.randint(0, 1 << 30) == (np.random.set_state(s_want) or np.random.randint(0, 1 << 30))
    with pytest.raises(ValueError):
        _lib.legacy_randint(np.array([2, 0, 3]))


def test_no_product_import_of_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, 'telescope_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_score_lut_matches_reference_expression():
    c = load_case('bundled')
    raw = case_matrix(c)
    lut = score_lut(int(c['max_score']))
    assert np.array_equal(lut[raw.data], c['Q_data'])
    c = load_case('tiny_wide_range')
    assert np.array_equal(score_lut(int(c['max_score']))[case_matrix(c).data], c['Q_data'])


def test_synthetic_generator_spec():
    ip, ix, rw = synthetic.generate(5000, 300, 12, seed=7, dist='zipf', uniq_frac=0.1)
    lens = np.diff(ip)
    assert lens.min() >= 1 and abs(lens[lens > 1].mean() - 12) < 0.5
    assert 0.07 < (lens == 1).mean() < 0.13
    assert rw.min() >= 139 and rw.max() <= 300 and ix.min() >= 0 and ix.max() < 300
    starts = np.zeros(len(ix), bool); starts[ip[:-1]] = True
    assert (np.diff(ix)[~starts[1:]] > 0).all()          # sorted, no duplicates within a row
    # any row range reproduces the same rows
    ip2, ix2, rw2 = synthetic.generate(5000, 300, 12, seed=7, dist='zipf', uniq_frac=0.1,
                                       row_begin=1234, row_end=2345)
    assert np.array_equal(ix2, ix[ip[1234]:ip[2345]]) and np.array_equal(rw2, rw[ip[1234]:ip[2345]])
    # zipf is skewed, uniform is not
    _, ixu, _ = synthetic.generate(5000, 300, 12, seed=7, dist='uniform')
    assert np.bincount(ix, minlength=300)[1] > 5 * np.bincount(ixu, minlength=300)[1]


def test_poisson_cdf_table():
    cdf = synthetic.poisson_cdf_u32(40)
    assert (np.diff(cdf.astype(np.int64)) >= 0).all() and cdf[-1] > 0.999999 * 2 ** 32
    h = (synthetic.hash3(1, np.arange(200000), 0) >> np.uint64(32)).astype(np.uint32)
    s = np.searchsorted(cdf, h, side='right')
    assert abs(s.mean() - 40) < 0.1 and abs(s.var() - 40) < 1.0


def test_shard_bounds():
    assert shard_bounds(10, 4) == [0, 2, 5, 7, 10]
    assert shard_bounds(10, 4, 3) == (7, 10)
    indptr = np.concatenate([[0], np.cumsum([100] * 5 + [1] * 95)])
    cuts = shard_bounds(100, 2, indptr=indptr)
    assert cuts[0] == 0 and cuts[-1] == 100 and 2 <= cuts[1] <= 4      # balanced by nnz, not rows
    assert shard_bounds(3, 8)[-1] == 3 and len(shard_bounds(3, 8)) == 9  # more ranks than rows


def test_single_process_cli_runs_do_not_import_torch():
    """`import torch` is most of the wall clock of a small `telescope resume`; the loader skips it when the command line is not a
    rank of a torch.distributed launch (TSEM_NO_TORCH=1, telescope_amd/cli.py) and keeps it otherwise (torch's HIP runtime must be
    the one both share when torch is going to be used)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); from telescope_amd import _lib; _lib.lib(); "
            "print('torch' in sys.modules)" % root)
    for env_val, want in (('1', 'False'), ('0', 'True')):
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, TSEM_NO_TORCH=env_val))
        assert r.returncode == 0, r.stderr[-1500:]
        assert r.stdout.strip().splitlines()[-1] == want, (env_val, r.stdout)


def test_every_translation_unit_is_built():
    """telescope_amd/_lib.py compiles the library's units and the units that instantiate the fused kernel in parallel: every *.hip under csrc/
    must be on its list (a unit left out would only show up as a missing kernel at run time)."""
    import glob
    from telescope_amd import _lib
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'telescope_amd', 'csrc')
    on_disk = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(csrc, '*.hip')))
    assert on_disk == sorted(list(_lib.LIB_UNITS) + list(_lib.FZ_UNITS))


def test_family_distribution_keeps_a_row_inside_one_family():
    """dist='family' (telescope_amd/synthetic.py): every row draws its columns inside ONE family of 256 consecutive loci (plus
    column 0 in ~5 % of the rows), sorted, without duplicates; hot families exist (cubic skew)."""
    from telescope_amd import synthetic
    ip, ix, rw = synthetic.generate(4000, 3000, 20, seed=7, dist='family', uniq_frac=0.05)
    fam_rows = {}
    for i in range(4000):
        c = ix[ip[i]:ip[i + 1]]
        assert np.all(np.diff(c) > 0)
        c = c[c > 0]
        if len(c):
            f = (c - 1) // synthetic.FAMILY
            assert f.min() == f.max() < (3000 - 1) // synthetic.FAMILY
            fam_rows[int(f[0])] = fam_rows.get(int(f[0]), 0) + 1
    assert len(fam_rows) == (3000 - 1) // synthetic.FAMILY and fam_rows[0] > 3 * fam_rows[max(fam_rows)]
    assert rw.min() >= 139 and rw.max() <= 300


def test_row_sum_restatement_is_numpys_reduceat(tmp_path):
    """telescope_amd/csrc/tsem_npsum.h restates the order of additions of np.add.reduceat (what scipy's CSR `sum(axis=1)` is, hence
    what sparse_plus.py:51 normalises z by): a0 + pairwise(a1 ..).  The header has no HIP dependency: compiled here with g++ and checked
    bit for bit against numpy and against scipy on random rows of 1 .. 5000 terms (the device runs the same code for near-tie rows)."""
    import ctypes
    import subprocess
    import scipy.sparse as sp
    src = tmp_path / 'h.cpp'
    src.write_text('#include "tsem_npsum.h"\n'
                   'struct Cur { const double* p; double operator()() { return *p++; } };\n'
                   'extern "C" double nps(const double* a, long m) { Cur c{a}; return np_reduceat_sum(c, m); }\n')
    so = tmp_path / 'libnps.so'
    subprocess.run(['g++', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-I' + os.path.join(ROOT, 'telescope_amd', 'csrc'),
                    str(src), '-o', str(so)], check=True)
    L = ctypes.CDLL(str(so))
    L.nps.restype = ctypes.c_double
    L.nps.argtypes = [np.ctypeslib.ndpointer(np.float64), ctypes.c_long]
    rng = np.random.RandomState(1)
    for t in range(4000):
        m = int(rng.choice([1, 2, 3, 5, 8, 9, 10, 16, 17, 40, 100, 129, 130, 137, 200, 257, 300, 1000, 5000, rng.randint(1, 3000)]))
        a = rng.rand(m) * 10.0 ** rng.randint(-3, 3, m)
        assert L.nps(a, m) == np.add.reduceat(a, [0])[0], m
    lens = rng.randint(1, 400, 500)
    indptr = np.concatenate([[0], np.cumsum(lens)])
    data = rng.rand(indptr[-1])
    M = sp.csr_matrix((data, np.concatenate([np.arange(l) for l in lens]), indptr), shape=(500, 400))
    s = np.asarray(M.sum(1)).ravel()
    assert all(L.nps(data[indptr[i]:indptr[i + 1]].copy(), int(lens[i])) == s[i] for i in range(500))


def test_the_torch_transport_is_opt_in():
    """when the in-library RCCL communicator cannot be created the default is to END the run with the library's
    message on every rank; torch.distributed collectives are taken only with TSEM_ALLOW_TORCH_COLLECTIVES=1 (distributed.py
    decide_transport; the GPU leg is tests/test_gpu_round2.py test_torch_transport_fallback)."""
    from telescope_amd.distributed import decide_transport
    assert decide_transport(False, None, {}) == 'library'
    with pytest.raises(_lib.EngineError) as e:
        decide_transport(True, 'ncclCommInitRank did not return within the time limit', {})
    assert 'ncclCommInitRank did not return' in str(e.value) and 'TSEM_ALLOW_TORCH_COLLECTIVES=1' in str(e.value)
    assert decide_transport(True, 'x', {'TSEM_ALLOW_TORCH_COLLECTIVES': '1'}) == 'torch'
    with pytest.raises(_lib.EngineError):
        decide_transport(True, 'x', {'TSEM_ALLOW_TORCH_COLLECTIVES': '0', 'TSEM_TORCH_COLLECTIVES': '1'})


def test_bench_config_presets(monkeypatch):
    """`bench.py --config N` names BASELINE.json's configurations: rows / columns / row length / scaling per preset."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')

    def parse(*argv):
        monkeypatch.setattr(sys, 'argv', ['bench.py'] + list(argv))
        return bench.parse()
    a = parse('--config', '5', '--gpus', '8')
    assert (a.rows, a.cols, a.nnz_row, a.scaling, a.value_format, a.properties) == (25_000_000, 50_000, 100.0, 'weak', 'auto', True)
    assert 'BASELINE config 5' in a.config_note and '8 rank(s) x 25M rows' in a.config_note
    a = parse('--config', '5')
    assert a.rows == 100_000_000 and 'the half that fits one GPU' in a.config_note
    a = parse('--config', '5', '--gpus', '2', '--config-scale', '0.01')
    assert a.rows == 250_000 and 'AT 0.01 OF ITS ROWS' in a.config_note
    a = parse('--config', '4', '--gpus', '8')
    assert (a.rows, a.cols, a.nnz_row, a.scaling) == (50_000_000, 30_000, 40.0, 'strong')
    a = parse('--config', '2')
    assert (a.rows, a.cols, a.nnz_row) == (1_000_000, 30_000, 20.0)
    a = parse()
    assert a.config == 0 and a.rows == 50_000_000 and a.config_note is None


def test_c_host_compiles_as_c99_and_fails_loudly_without_a_device(built, tmp_path):
    """tests/c_host/run_bundled.c against include/telescope_em.h with `gcc -std=c99 -pedantic -Werror`, linked against the in-tree
    libtelescope_em.so; in this container (no GPU) it must stop at tsem_create with the library's message — no CPU path (the run on
    a GPU is tests/test_gpu_round6.py).  Also: tsem_score_lut is libm's table, within 1 ulp of the reference's numpy table."""
    import subprocess
    exe = str(tmp_path / 'run_bundled')
    lib_dir = os.path.join(ROOT, 'telescope_amd')
    subprocess.run(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-O2', '-I' + os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'tests', 'c_host', 'run_bundled.c'), '-o', exe, '-L' + lib_dir, '-ltelescope_em',
                    '-Wl,-rpath,' + lib_dir, '-Wl,-rpath-link,/opt/rocm/lib'], check=True)
    lut = np.zeros(213)
    assert built.tsem_score_lut(212, 100.0, lut.ctypes.data_as(ctypes.c_void_p)) == 0
    ref = score_lut(212)
    assert np.all(np.abs(lut - ref) <= np.spacing(ref)) and lut[0] == 0.0
    try:
        _lib.Engine(0).close()
        pytest.skip('a GPU is present: the loud failure is a CPU-box property')
    except _lib.EngineError:
        pass
    r = subprocess.run([exe, os.path.join(ROOT, 'tests', 'golden', 'bundled_flat.bin')], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and 'tsem_create' in r.stderr and 'HIP' in r.stderr, (r.returncode, r.stderr)
