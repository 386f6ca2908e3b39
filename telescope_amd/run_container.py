"""Run container for the `telescope resume` path: checkpoint I/O, RNG seed, reports.

Mirrors the caller contract of the reference's `Telescope` class
(/root/reference/telescope/utils/model.py:74-564) as far as the accelerated
path needs it: `load` / `save` (model.py:108-148, the `.npz` checkpoint schema),
`get_random_seed` (model.py:150-153), `print_summary` (model.py:523-555) and
`output_report` (model.py:420-477, the two TSVs incl. the header glued to the
RunInfo comment).  BAM loading and `update_sam` are out of scope.
"""
import logging as lg
from collections import Counter, OrderedDict

import numpy as np
import pandas as pd
import scipy.sparse as sp


def _str2int(s):
    """helpers.py:149-156 — run_info values come back as int, float or str."""
    for cast in (int, float):
        try:
            return cast(s)
        except ValueError:
            pass
    return s


class Telescope(object):
    def __init__(self, opts=None):
        self.opts = opts
        self.run_info = OrderedDict()
        self.feature_length = Counter()
        self.read_index, self.feat_index = {}, {}
        self.shape = None
        self.raw_scores = None

    # ---- alignment loading (model.py:155-173, via telescope_amd/loader.py) --------
    def load_alignment(self, annotation):
        from . import loader
        o = self.opts
        self.run_info['annotated_features'] = len(annotation.loci)
        r = loader.load_alignment(o.samfile, annotation, o.no_feature_key, o.overlap_mode,
                                  o.overlap_threshold, o.stranded_mode)
        self.feature_length = r['feature_length']
        self.read_index, self.feat_index = r['read_index'], r['feat_index']
        self.raw_scores = r['raw_scores']
        self.shape = self.raw_scores.shape
        for k, v in r['run_info'].items():
            self.run_info[k] = v

    # ---- checkpoint (model.py:108-148) -------------------------------------
    def save(self, filename):
        feats = sorted(self.feat_index, key=self.feat_index.get)
        raw = sp.csr_matrix(self.raw_scores)
        np.savez(filename,
                 _run_info=list(self.run_info.items()),
                 _flen_list=[self.feature_length[f] for f in feats],
                 _feat_list=feats,
                 _read_list=sorted(self.read_index, key=self.read_index.get),
                 _shape=self.shape,
                 _raw_scores_data=raw.data, _raw_scores_indices=raw.indices,
                 _raw_scores_indptr=raw.indptr, _raw_scores_shape=raw.shape)

    @classmethod
    def load(cls, filename):
        z = np.load(filename)
        obj = cls()
        for k, v in z['_run_info']:
            obj.run_info[str(k)] = _str2int(str(v))
        for f, fl in zip(z['_feat_list'], z['_flen_list']):
            obj.feature_length[str(f)] = fl
        obj.read_index = {str(n): i for i, n in enumerate(z['_read_list'])}
        obj.feat_index = {str(n): i for i, n in enumerate(z['_feat_list'])}
        obj.shape = (len(obj.read_index), len(obj.feat_index))
        if tuple(z['_shape']) != obj.shape:
            raise AssertionError('checkpoint shape %s does not match its name lists %s'
                                 % (tuple(z['_shape']), obj.shape))
        obj.raw_scores = sp.csr_matrix((z['_raw_scores_data'], z['_raw_scores_indices'],
                                        z['_raw_scores_indptr']), shape=tuple(z['_raw_scores_shape']))
        return obj

    def get_random_seed(self):
        """model.py:150-153 — note the precedence: (total % N) * K, then mod 2^32-1."""
        ret = self.run_info['total_fragments'] % self.shape[0] * self.shape[1]
        return ret % 4294967295

    # ---- log summary (model.py:523-555) -----------------------------------------
    def print_summary(self, loglev=lg.WARNING):
        d = Counter()
        for k, v in self.run_info.items():
            try:
                d[k] = int(v)
            except ValueError:
                pass
        if 'mapped_pairs' in d:
            d['pair_mapped'] = d['mapped_pairs']
        if 'mapped_single' in d:
            d['single_mapped'] = d['mapped_single']
        say = lambda m: lg.log(loglev, m)  # noqa: E731
        say("Alignment Summary:")
        say('    {} total fragments.'.format(d['total_fragments']))
        say('        {} mapped as pairs.'.format(d['pair_mapped']))
        say('        {} mapped as mixed.'.format(d['pair_mixed']))
        say('        {} mapped single.'.format(d['single_mapped']))
        say('        {} failed to map.'.format(d['unmapped']))
        say('--')
        say('    {} fragments mapped to reference; of these'.format(
            d['pair_mapped'] + d['pair_mixed'] + d['single_mapped']))
        say('        {} had one unique alignment.'.format(d['unique']))
        say('        {} had multiple alignments.'.format(d['ambig']))
        say('--')
        say('    {} fragments overlapped annotation; of these'.format(d['overlap_unique'] + d['overlap_ambig']))
        say('        {} map to one locus.'.format(d['overlap_unique']))
        say('        {} map to multiple loci.'.format(d['overlap_ambig']))
        say('\n')

    # ---- reports (model.py:420-477) -----------------------------------------------
    def output_report(self, tl, stats_filename, counts_filename, write=True):
        """Same columns, evaluation order (the RNG is consumed by init_best_random before the
        final mode), sort, rounding and file layout as the reference — including the header row
        glued onto the RunInfo comment line (model.py:470-471 writes no newline)."""
        mode, prob = self.opts.reassign_mode, self.opts.conf_prob
        names = sorted(self.feat_index, key=self.feat_index.get)
        colsum = getattr(tl, 'reassign_colsums', None) or \
            (lambda m, t=0.9, initial=False: tl.reassign(m, t, initial).sum(0).A1)
        stats = pd.DataFrame(OrderedDict([
            ('transcript', names),
            ('transcript_length', [self.feature_length[f] for f in names]),
            ('final_conf', colsum('conf', prob)),
            ('final_prop', tl.pi),
            ('init_aligned', colsum('all', initial=True)),
            ('unique_count', colsum('unique')),
            ('init_best', colsum('exclude', initial=True)),
            ('init_best_random', colsum('choose', initial=True)),
            ('init_best_avg', colsum('average', initial=True)),
            ('init_prop', tl.pi_init),
        ]))
        stats.sort_values('final_prop', ascending=False, inplace=True)
        stats = stats.round(pd.Series([2, 3, 2, 3],
                                      index=['final_conf', 'final_prop', 'init_best_avg', 'init_prop']))
        counts = pd.DataFrame(OrderedDict([('transcript', names), ('count', colsum(mode, prob))]))
        counts.sort_values('transcript', inplace=True)
        if not write:                                        # a rank other than 0 of a row-sharded run: it took part in the sums
            return
        comment = ['## RunInfo'] + ['{}:{}'.format(k, v) for k, v in self.run_info.items()]
        with open(stats_filename, 'w') as fh:
            fh.write('\t'.join(comment))
            stats.to_csv(fh, sep='\t', index=False)
        with open(counts_filename, 'w') as fh:
            counts.to_csv(fh, sep='\t', index=False)
