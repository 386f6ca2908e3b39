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


class NpzSlices(object):
    """Parts of the arrays of an `.npz` without reading the file: `np.savez` (model.py:133) stores its members UNCOMPRESSED, so
    element [a, b) of a 1-D member sits at a known offset of the archive — the local zip header, then the `.npy` header, then the
    raw values — and is read with one seek + readinto.  A compressed member (np.savez_compressed) is read whole, once, and sliced."""

    def __init__(self, filename):
        import zipfile
        self.filename = filename
        self._zf = zipfile.ZipFile(filename)
        self._whole = {}

    def close(self):
        self._zf.close()

    def names(self):
        return [n[:-4] for n in self._zf.namelist() if n.endswith('.npy')]

    def _member(self, name):
        import struct
        import zipfile
        info = self._zf.getinfo(name + '.npy')
        if info.compress_type != zipfile.ZIP_STORED:
            return None
        with open(self.filename, 'rb') as fh:
            fh.seek(info.header_offset)
            hdr = fh.read(30)                                   # local file header: signature, ..., name length (26), extra length (28)
            if hdr[:4] != b'PK\x03\x04':
                return None
            n_name, n_extra = struct.unpack('<HH', hdr[26:30])
            fh.seek(info.header_offset + 30 + n_name + n_extra)
            major, _minor = np.lib.format.read_magic(fh)
            shape, fortran, dtype = (np.lib.format.read_array_header_1_0(fh) if major == 1 else np.lib.format.read_array_header_2_0(fh))
            return fh.tell(), shape, fortran, dtype

    def shape(self, name):
        m = self._member(name)
        return m[1] if m is not None else self.whole(name).shape

    def whole(self, name):
        if name not in self._whole:
            with self._zf.open(name + '.npy') as fh:
                self._whole[name] = np.lib.format.read_array(fh, allow_pickle=False)
        return self._whole[name]

    def read(self, name, start=0, stop=None):
        """Elements [start, stop) of the 1-D member `name` (a fresh array)."""
        m = self._member(name)
        if m is None or len(m[1]) != 1 or m[3].hasobject:
            a = self.whole(name)
            return np.array(a[start:stop])
        off, shape, _fortran, dtype = m
        stop = shape[0] if stop is None else min(stop, shape[0])
        start = min(max(start, 0), stop)
        out = np.empty(stop - start, dtype=dtype)
        with open(self.filename, 'rb') as fh:
            fh.seek(off + start * dtype.itemsize)
            got = fh.readinto(memoryview(out).cast('B')) if out.size else 0
        if got != out.nbytes:
            raise IOError('%s: member %s is shorter than its header says' % (self.filename, name))
        return out


class Telescope(object):
    def __init__(self, opts=None):
        self.opts = opts
        self.run_info = OrderedDict()
        self.feature_length = Counter()
        self.read_index, self.feat_index = {}, {}
        self.shape = None
        self.raw_scores = None
        self.row_range = None            # (r0, r1): `raw_scores` holds only these fragments (load_shard); None = all of them

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

    @classmethod
    def load_shard(cls, filename, world, rank):
        """This rank's share of a checkpoint for a row-sharded run: the run information, the feature lists and the row pointers are
        read whole (K- and N-sized), the stored entries only for the rank's contiguous range of fragments — balanced by entries,
        `distributed.shard_bounds` — straight from their offsets in the archive (NpzSlices), and the fragment names (`_read_list`,
        the largest member after the entries; nothing on the EM / report path uses them) not at all.  A rank of an 8-way run of the
        50M-fragment checkpoint touches 1.6 GB of a 12.4 GB file instead of all of it.  `raw_scores` is the rank's (r1 - r0) x K
        slice, `row_range` = (r0, r1), `shape` the whole matrix's (the seed rule uses it, model.py:150-153)."""
        from .distributed import shard_bounds
        z = NpzSlices(filename)
        try:
            obj = cls()
            for k, v in z.whole('_run_info'):
                obj.run_info[str(k)] = _str2int(str(v))
            feats, flens = z.whole('_feat_list'), z.whole('_flen_list')
            for f, fl in zip(feats, flens):
                obj.feature_length[str(f)] = fl
            obj.feat_index = {str(n): i for i, n in enumerate(feats)}
            obj.read_index = None                                # not loaded (see above)
            obj.shape = tuple(int(x) for x in z.whole('_shape'))
            n_rows, n_cols = (int(x) for x in z.whole('_raw_scores_shape'))
            if obj.shape != (n_rows, n_cols) or len(obj.feat_index) != n_cols or z.shape('_read_list')[0] != n_rows:
                raise AssertionError('checkpoint shape %s does not match its matrix %s / name lists' % (obj.shape, (n_rows, n_cols)))
            indptr = z.read('_raw_scores_indptr')
            r0, r1 = shard_bounds(n_rows, world, rank, indptr=indptr)
            e0, e1 = int(indptr[r0]), int(indptr[r1])
            local_ptr = indptr[r0:r1 + 1] - indptr[r0]
            del indptr
            obj.raw_scores = sp.csr_matrix((z.read('_raw_scores_data', e0, e1), z.read('_raw_scores_indices', e0, e1), local_ptr),
                                           shape=(r1 - r0, n_cols))
            obj.row_range = (r0, r1)
            return obj
        finally:
            z.close()

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
