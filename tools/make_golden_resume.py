#!/usr/bin/env python3
"""Golden files for the `telescope resume` path, produced by the REFERENCE (dev container only).

Builds a reference `Telescope` run container around the bundled score matrix (without pysam:
`Telescope.__new__` + the attributes `save()` needs), writes the checkpoint with the reference's
`Telescope.save`, reloads it with `Telescope.load`, runs the reference EM and writes the two TSVs
with the reference's `output_report` — for every reassign mode.  Outputs (data only):
  tests/golden/resume_checkpoint.npz
  tests/golden/resume_<mode>-run_stats.tsv, resume_<mode>-TE_counts.tsv
"""
import os
import sys
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from ref_import import load_reference  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


class Opts(object):
    em_epsilon, max_iter, pi_prior, theta_prior = 1e-7, 100, 0, 200000
    conf_prob = 0.9
    reassign_mode = 'exclude'


def main():
    Telescope, TelescopeLikelihood, csr_plus = load_reference()
    f = np.load(os.path.join(GOLD, 'bundled_raw_scores.npz'))
    ts = Telescope.__new__(Telescope)
    ts.run_info = OrderedDict((k, v) for k, v in f['run_info'])
    ts.feature_length = {n: int(l) for n, l in zip(f['feat_names'], f['feat_lengths'])}
    ts.read_index = {n: i for i, n in enumerate(f['read_names'])}
    ts.feat_index = {n: i for i, n in enumerate(f['feat_names'])}
    ts.shape = (len(ts.read_index), len(ts.feat_index))
    ts.raw_scores = csr_plus((f['data'], f['indices'], f['indptr']), shape=tuple(f['shape']))
    ck = os.path.join(GOLD, 'resume_checkpoint.npz')
    ts.save(ck)
    for mode in ('exclude', 'choose', 'average', 'conf', 'unique'):
        ts2 = Telescope.load(ck)
        o = Opts(); o.reassign_mode = mode
        ts2.opts = o
        np.random.seed(ts2.get_random_seed())
        tl = TelescopeLikelihood(ts2.raw_scores, o)
        tl.em()
        ts2.output_report(tl, os.path.join(GOLD, 'resume_%s-run_stats.tsv' % mode),
                          os.path.join(GOLD, 'resume_%s-TE_counts.tsv' % mode))
        print(mode, 'seed', ts2.get_random_seed(), 'lnl', tl.lnl)


if __name__ == '__main__':
    main()
