"""BAM + GTF -> score matrix without pysam / intervaltree.

Host-side restatement of the reference's SEQUENTIAL loader so that
`telescope assign` is self-contained on the GPU box:

  utils/alignment.py:115-161      bundles of equal query name; pairing by read/mate key;
                                  fragment classes SU SM PU PM PX
  utils/calignment.pyx:83-98      refblocks (merge_blocks(., 1)), alnlen, alnscore = AS1 (+ AS2)
  utils/helpers.py:74-104         merge_blocks
  utils/_annotation_intervaltree.py:29-102   exon intervals [start, end+1), same-locus overlaps
                                  merged, block overlap lengths, optional strand filter
  utils/model.py:877-897          `threshold` assignment: best locus iff overlap > alnlen * threshold
  utils/model.py:30-63            best alignment per locus per fragment (max alnscore + alnlen)
  utils/model.py:214-285          run counters (unique / ambig / overlap ...)
  utils/model.py:287-362          rescale (AS - minAS + 1) + alnlen as uint16, max per (fragment,
                                  locus), first-appearance ids, drop fragments hitting only
                                  column 0 (`__no_feature`)

This is loader glue, not the accelerated path: plain Python, BAM only (BGZF is a
sequence of gzip members, so the stdlib `gzip` module reads it), name-collated
input as the reference requires.  Not covered: `--updated_sam` BAM rewriting,
`--ncpu > 1` (broken at the reference's HEAD), single-cell barcodes.

Size: the BAM is STREAMED (one record at a time through a buffered gzip reader: memory does not grow with
the file), the mappings are kept as four compact integer arrays (16 B per (fragment, locus) hit, like the
reference's list of tuples but ~10x smaller), and the annotation is indexed per chromosome.  It is still a
pure-Python record loop: ~50-100 k records/s, i.e. minutes for the 10^7-record BAMs of a typical RNA-seq run
and hours beyond 10^8 — use the reference's pysam loader and `telescope resume` on its checkpoint for those.
"""
import gzip
import io
import re
import struct
from array import array
from collections import Counter, OrderedDict, defaultdict

import numpy as np
import scipy.sparse as sp

BIG_INT = 2 ** 32 - 1
_TAG_FMT = {'c': 'b', 'C': 'B', 's': 'h', 'S': 'H', 'i': 'i', 'I': 'I', 'f': 'f'}
_B_SIZE = {'c': 1, 'C': 1, 's': 2, 'S': 2, 'i': 4, 'I': 4, 'f': 4}


class Segment(object):
    __slots__ = ('qname', 'flag', 'ref_id', 'pos', 'nref', 'npos', 'tlen', 'cigar', 'AS')

    @property
    def is_paired(self): return bool(self.flag & 0x1)
    @property
    def is_proper_pair(self): return bool(self.flag & 0x2)
    @property
    def is_unmapped(self): return bool(self.flag & 0x4)
    @property
    def is_reverse(self): return bool(self.flag & 0x10)
    @property
    def is_read1(self): return bool(self.flag & 0x40)

    def blocks(self):
        """Gapless reference blocks (pysam get_blocks): M/=/X emit and advance, D/N advance."""
        out, pos = [], self.pos
        for c in self.cigar:
            ln, op = c >> 4, c & 0xF
            if op in (0, 7, 8):
                out.append((pos, pos + ln)); pos += ln
            elif op in (2, 3):
                pos += ln
        return out


def read_bam(path):
    """-> (reference names, iterator of Segment) from a BAM file, streamed."""
    fh = io.BufferedReader(gzip.open(path, 'rb'), buffer_size=1 << 20)

    def need(n):
        b = fh.read(n)
        if len(b) != n:
            raise ValueError('%s: truncated BAM' % path)
        return b

    if fh.read(4) != b'BAM\x01':
        raise ValueError('%s is not a BAM file' % path)
    (l_text,) = struct.unpack('<i', need(4))
    need(l_text)
    (n_ref,) = struct.unpack('<i', need(4))
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack('<i', need(4))
        refs.append(need(l_name)[:-1].decode())
        need(4)

    def records():
        try:
            while True:
                head = fh.read(4)
                if not head:
                    return
                if len(head) != 4:
                    raise ValueError('%s: truncated BAM record' % path)
                (bs,) = struct.unpack('<i', head)
                buf = need(bs)
                s = Segment()
                (s.ref_id, s.pos, l_rn, _mq, _bin, n_cig, s.flag, l_seq, s.nref, s.npos,
                 s.tlen) = struct.unpack_from('<iiBBHHHiiii', buf, 0)
                p = 32
                s.qname = buf[p:p + l_rn - 1].decode(); p += l_rn
                s.cigar = struct.unpack_from('<%dI' % n_cig, buf, p); p += 4 * n_cig
                p += (l_seq + 1) // 2 + l_seq
                s.AS = None
                while p < bs:
                    tag = buf[p:p + 2]; typ = chr(buf[p + 2]); p += 3
                    if typ in _TAG_FMT:
                        (val,) = struct.unpack_from('<' + _TAG_FMT[typ], buf, p)
                        p += struct.calcsize(_TAG_FMT[typ])
                        if tag == b'AS':
                            s.AS = val
                    elif typ == 'A':
                        p += 1
                    elif typ in 'ZH':
                        p = buf.index(b'\x00', p) + 1
                    elif typ == 'B':
                        sub = chr(buf[p]); (cnt,) = struct.unpack_from('<i', buf, p + 1)
                        p += 5 + cnt * _B_SIZE[sub]
                    else:
                        raise ValueError('unknown BAM tag type %r' % typ)
                yield s
        finally:
            fh.close()
    return refs, records()


def merge_blocks(ivs, dist=0):
    if len(ivs) <= 1:
        return ivs
    ivs = sorted(ivs, key=lambda x: x[0])
    ret = [ivs[0]]
    for iv in ivs[1:]:
        if iv[0] - ret[-1][1] > dist:
            ret.append(iv)
        else:
            ret[-1] = (ret[-1][0], max(iv[1], ret[-1][1]))
    return ret


class AlignedPair(object):
    __slots__ = ('r1', 'r2', 'refblocks', 'alnlen', 'alnscore')

    def __init__(self, r1, r2=None):
        self.r1, self.r2 = r1, r2
        if r1.is_unmapped:
            self.refblocks, self.alnlen, self.alnscore = [], 0, 0
            return
        b = r1.blocks() + (r2.blocks() if r2 is not None else [])
        self.refblocks = merge_blocks(b, 1)
        self.alnlen = sum(e - s for s, e in self.refblocks)
        self.alnscore = r1.AS + (r2.AS if r2 is not None else 0)

    @property
    def is_unmapped(self): return self.r1.is_unmapped
    @property
    def is_paired(self): return self.r2 is not None


class Annotation(object):
    """Exon intervals per chromosome; overlapping exons of the same locus are merged."""

    def __init__(self, gtf_file, attribute_name='locus', stranded_mode='None', feature_type='exon'):
        self.key = attribute_name
        self.loci = OrderedDict()
        by_locus = defaultdict(list)             # (chrom, locus) -> [[begin, end, locus, strand], ...]: the same-locus merge
                                                 # only ever looks at that locus's own intervals (no scan of the chromosome)
        self.run_stranded = stranded_mode != 'None'
        fh = open(gtf_file) if isinstance(gtf_file, str) else gtf_file
        for line in fh:
            if line.startswith('#'):
                continue
            f = line.rstrip('\n').split('\t')
            if len(f) < 9 or f[2] != feature_type:
                continue
            attr = dict(re.findall(r'(\w+)\s+"(.+?)";', f[8]))
            if self.key not in attr:
                continue
            loc = attr[self.key]
            self.loci.setdefault(loc, []).append(f)
            b, e = int(f[3]), int(f[4]) + 1
            ivs = by_locus[(f[0], loc)]
            hit = [iv for iv in ivs if iv[0] < e and b < iv[1]]
            if hit:
                assert len(hit) == 1, 'Error'
                ivs.remove(hit[0])
                b, e = min(b, hit[0][0]), max(e, hit[0][1])
            ivs.append([b, e, loc, f[6]])
        self.by_chrom = defaultdict(list)        # chrom -> [begin, end, locus, strand]
        for (chrom, _), ivs in by_locus.items():
            self.by_chrom[chrom].extend(ivs)
        self._index = None

    def feature_length(self):
        ret = Counter()
        for ivs in self.by_chrom.values():
            for b, e, loc, _ in ivs:
                ret[loc] += e - b
        return ret

    def _sorted(self, chrom):
        if self._index is None:
            self._index = {}
        if chrom not in self._index:
            ivs = sorted(self.by_chrom.get(chrom, ()), key=lambda iv: iv[0])
            starts = np.array([iv[0] for iv in ivs], dtype=np.int64)
            maxend = np.maximum.accumulate(np.array([iv[1] for iv in ivs], dtype=np.int64)) if ivs else starts
            self._index[chrom] = (ivs, starts, maxend)
        return self._index[chrom]

    def intersect_blocks(self, ref, blocks, frag_strand):
        ivs, starts, maxend = self._sorted(ref)
        res = Counter()
        for bs, be in blocks:
            qb, qe = bs, be + 1
            hi = int(np.searchsorted(starts, qe, side='left'))        # intervals starting before the query ends
            lo = int(np.searchsorted(maxend, qb, side='right')) if hi else 0
            for b, e, loc, strand in ivs[lo:hi]:
                if b < qe and qb < e and (not self.run_stranded or strand == frag_strand):
                    res[loc] += max(0, min(e, qe) - max(b, qb))
        return res


def _fragments(records):
    """alignment.py:115-161 -> (code, [AlignedPair...]) per bundle of equal query names."""
    def classify(alns):
        if not alns[0].is_paired:
            return ('SU' if alns[0].is_unmapped else 'SM'), [AlignedPair(a) for a in alns]
        if alns[0].is_proper_pair:
            cache, pairs = {}, []
            for a in alns:
                rk = (a.qname, a.is_read1, a.ref_id, a.pos, a.nref, a.npos, abs(a.tlen))
                mk = (a.qname, not a.is_read1, a.nref, a.npos, a.ref_id, a.pos, abs(a.tlen))
                mate = cache.pop(mk, None)
                if mate is not None:
                    pairs.append(AlignedPair(a, mate) if a.is_read1 else AlignedPair(mate, a))
                else:
                    cache[rk] = a
            return 'PM', pairs + [AlignedPair(a) for a in cache.values()]
        if len(alns) == 2 and all(a.is_unmapped for a in alns):
            return 'PU', [AlignedPair(alns[0], alns[1])]
        return 'PX', [AlignedPair(a) for a in alns]

    bundle = []
    for r in records:
        if bundle and r.qname != bundle[0].qname:
            yield classify(bundle)
            bundle = []
        bundle.append(r)
    if bundle:
        yield classify(bundle)


CODE_DESC = OrderedDict([('SU', 'single_unmapped'), ('SM', 'single_mapped'), ('PU', 'pair_unmapped'),
                         ('PM', 'pair_mapped'), ('PX', 'pair_mixed'), ('PX*', 'pair_mixed_unmapped')])


def load_alignment(samfile, annotation, no_feature_key='__no_feature', overlap_mode='threshold',
                   overlap_threshold=0.2, stranded_mode='None'):
    """-> dict(raw_scores uint16 CSR, read_index, feat_index, feature_length, run_info fields)."""
    if overlap_mode != 'threshold':
        raise NotImplementedError('only overlap_mode "threshold" is implemented (as in the reference, '
                                  'model.py:899-903)')
    refs, records = read_bam(samfile)

    def assign(pair):
        if pair.r1.is_reverse:
            strand = ('+' if stranded_mode[-1] == 'F' else '-') if pair.is_paired else \
                     ('-' if stranded_mode[0] == 'F' else '+')
        else:
            strand = ('-' if stranded_mode[-1] == 'F' else '+') if pair.is_paired else \
                     ('+' if stranded_mode[0] == 'F' else '-')
        f = annotation.intersect_blocks(refs[pair.r1.ref_id], pair.refblocks, strand)
        if not f:
            return no_feature_key
        fname, overlap = f.most_common()[0]
        return fname if overlap > pair.alnlen * overlap_threshold else no_feature_key

    info = Counter()
    ridx, fidx = OrderedDict(), OrderedDict([(no_feature_key, 0)])
    m_row, m_col, m_as, m_len = array('q'), array('i'), array('i'), array('i')   # the reference's `_mappings`, 16 B per hit
    min_as, max_as = BIG_INT, -BIG_INT
    for code, alns in _fragments(records):
        info['total_fragments'] += 1
        info[code] += 1
        if code in ('SU', 'PU'):
            continue
        mapped = [a for a in alns if not a.is_unmapped]
        ambig = len(mapped) > 1
        for a in mapped:
            min_as, max_as = min(min_as, a.alnscore), max(max_as, a.alnscore)
        feats = [assign(a) for a in mapped]
        if not any(f != no_feature_key for f in feats):
            info['nofeat_A' if ambig else 'nofeat_U'] += 1
            continue
        info['feat_A' if ambig else 'feat_U'] += 1
        byfeat = OrderedDict()
        for a, f in zip(mapped, feats):
            byfeat.setdefault(f, []).append(a)
        maps = []
        for f, fal in byfeat.items():
            fal.sort(key=lambda x: x.alnscore + x.alnlen, reverse=True)
            maps.append((alns[0].r1.qname, f, fal[0].alnscore, fal[0].alnlen))
        maps.sort(key=lambda x: x[2], reverse=True)
        for rid, fid, ascr, alen in maps:                  # first-appearance ids (model.py:309-311)
            m_row.append(ridx.setdefault(rid, len(ridx)))
            m_col.append(fidx.setdefault(fid, len(fidx)))
            m_as.append(ascr); m_len.append(alen)

    # model.py:287-362: rescale, max per (fragment, locus), drop the fragments that hit only column 0
    rr, cc = np.frombuffer(m_row, dtype=np.int64), np.frombuffer(m_col, dtype=np.int32).astype(np.int64)
    vv = (np.frombuffer(m_as, dtype=np.int32).astype(np.int64) - min_as + 1) + np.frombuffer(m_len, dtype=np.int32)
    if vv.size and vv.max() > 65535:
        raise ValueError('alignment score %d does not fit uint16 (model.py:300)' % int(vv.max()))
    n_feat = len(fidx)
    key = rr * n_feat + cc
    order = np.argsort(key, kind='stable')
    key, vv = key[order], vv[order]
    first = np.ones(key.size, dtype=bool)
    first[1:] = key[1:] != key[:-1]
    starts = np.flatnonzero(first)
    vmax = np.maximum.reduceat(vv, starts) if key.size else vv
    ukey = key[starts]
    urow, ucol = ukey // n_feat, ukey % n_feat
    has_feat = np.zeros(len(ridx), dtype=bool)
    has_feat[urow[ucol != 0]] = True
    keep = np.flatnonzero(has_feat)
    remap = np.full(len(ridx), -1, dtype=np.int64)
    remap[keep] = np.arange(keep.size)
    sel = has_feat[urow]
    counts = np.bincount(remap[urow[sel]], minlength=keep.size)
    indptr = np.zeros(keep.size + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    names = list(ridx)
    raw = sp.csr_matrix((vmax[sel].astype(np.uint16), ucol[sel].astype(np.int32), indptr),
                        shape=(keep.size, n_feat))
    uniq = int(np.sum(np.diff(raw.indptr) == 1))
    info['unmapped'] = info['SU'] + info['PU']
    info['unique'] = info['nofeat_U'] + info['feat_U']
    info['ambig'] = info['nofeat_A'] + info['feat_A']
    info['overlap_unique'] = uniq
    info['overlap_ambig'] = raw.shape[0] - uniq
    for cs, desc in CODE_DESC.items():
        info[desc] = info[cs]
    fields = ['total_fragments', 'pair_mapped', 'pair_mixed', 'single_mapped', 'unmapped', 'unique',
              'ambig', 'overlap_unique', 'overlap_ambig']
    return dict(raw_scores=raw, read_index={names[int(old)]: new for new, old in enumerate(keep)},
                feat_index=dict(fidx), feature_length=annotation.feature_length(),
                run_info=OrderedDict((f, info[f]) for f in fields), score_range=(min_as, max_as))
