#!/usr/bin/env python3
"""BASELINE config 3: fp32-vs-fp64 tolerance sweep on the 10M x 30k x ~40 synthetic matrix.

    python tools/precision_sweep.py [--rows N] [--iters T] [--json]

Legs, each against the fp64 run on the same matrix (same iteration count, em_epsilon = 0):
  code16      2-byte score codes + fp64 score table (the library default): the SAME fp64 numbers, half the bytes
  store_f32   Q rounded to 24 significant bits (what a 4-byte value format stores), fp64 arithmetic and sums
  store_f16m  Q rounded to 11 significant bits, fp64 arithmetic and sums
  store_bf16m Q rounded to 8 significant bits, fp64 arithmetic and sums
  accum_f32   fp32 products, row sums, posteriors AND column sums (option em_precision = 1; diagnostic kernel)
Reported: relative delta of the final log-likelihood, max relative delta of pi over the loci that hold 99.9999 % of
the mass, per-locus final counts (`exclude`): loci whose integer count differs and the largest count difference
relative to the locus count, `conf` column sums: max relative delta.  The bar of north_star is 1e-4 relative on
lnl and final counts."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Opts(object):
    def __init__(self, iters):
        self.em_epsilon, self.max_iter, self.pi_prior, self.theta_prior = 0.0, iters, 0, 200000


def run(rows, cols, nnz_row, iters, options=(), mantissa=53, seed=42):
    from telescope_amd import synthetic
    from telescope_amd._lib import Engine
    from telescope_amd.likelihood import TelescopeLikelihood
    eng = Engine(0)
    for k, v in options:
        eng.set_option(k, v)
    eng.generate(0, rows, cols, synthetic.poisson_cdf_u32(nnz_row), seed, synthetic.DIST_CODE['zipf'], 0.05)
    tl = TelescopeLikelihood.from_engine(eng, Opts(iters), lut_mantissa_bits=mantissa)
    tl.em()
    out = dict(lnl=tl.lnl, pi=tl.pi.copy(), theta=tl.theta.copy(), excl=tl.reassign_colsums('exclude'),
               conf=tl.reassign_colsums('conf', 0.9), bytes=eng.layout_info()['value_bytes'])
    eng.close()
    return out


def compare(ref, x):
    order = np.argsort(-ref['pi'])
    keep = order[:max(1, int(np.searchsorted(np.cumsum(ref['pi'][order]), 1 - 1e-6)) + 1)]
    d_ex = np.abs(x['excl'] - ref['excl'])
    big = ref['excl'] >= 100
    return dict(lnl_rel=float(abs(x['lnl'] - ref['lnl']) / abs(ref['lnl'])),
                pi_max_rel=float(np.max(np.abs(x['pi'][keep] - ref['pi'][keep]) / ref['pi'][keep])),
                theta_max_rel=float(np.max(np.abs(x['theta'][keep] - ref['theta'][keep]) / ref['theta'][keep])),
                final_count_loci_differing=int(np.count_nonzero(d_ex)),
                final_count_max_abs=int(d_ex.max()),
                final_count_max_rel_loci_ge100=float(np.max(d_ex[big] / ref['excl'][big])) if big.any() else 0.0,
                final_count_total_moved=int(d_ex.sum() // 2),
                conf_max_rel=float(np.max(np.abs(x['conf'] - ref['conf']) / np.maximum(ref['conf'], 1.0))))


LEGS = (('code16', (('value_format', 2),), 53), ('store_f32', (('value_format', 1),), 24),
        ('store_f16m', (('value_format', 1),), 11), ('store_bf16m', (('value_format', 1),), 8),
        ('accum_f32', (('value_format', 1), ('em_precision', 1)), 24))


def sweep(rows=10_000_000, cols=30_000, nnz_row=40.0, iters=30):
    ref = run(rows, cols, nnz_row, iters, (('value_format', 1),))
    res = {'workload': dict(rows=rows, cols=cols, nnz_row=nnz_row, iterations=iters, dist='zipf', uniq_frac=0.05),
           'reference': 'fp64 stored values, fp64 arithmetic (value_format = 1)', 'lnl_fp64': ref['lnl'], 'legs': {}}
    for name, options, mant in LEGS:
        res['legs'][name] = compare(ref, run(rows, cols, nnz_row, iters, options, mant))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=10_000_000)
    ap.add_argument('--cols', type=int, default=30_000)
    ap.add_argument('--nnz-row', type=float, default=40.0)
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--json', action='store_true')
    a = ap.parse_args()
    res = sweep(a.rows, a.cols, a.nnz_row, a.iters)
    if a.json:
        print(json.dumps(res))
        return
    print('# %s' % res['workload'])
    cols = ('lnl_rel', 'pi_max_rel', 'theta_max_rel', 'final_count_loci_differing', 'final_count_max_abs',
            'final_count_max_rel_loci_ge100', 'final_count_total_moved', 'conf_max_rel')
    print('%-12s' % 'leg' + ''.join('%32s' % c for c in cols))
    for name, r in res['legs'].items():
        print('%-12s' % name + ''.join(('%32d' % r[c]) if isinstance(r[c], int) else ('%32.3e' % r[c]) for c in cols))


if __name__ == '__main__':
    main()
