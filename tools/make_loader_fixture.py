#!/usr/bin/env python3
"""Loader coverage fixture: a small synthetic BAM + GTF that exercises every fragment class of
the reference's sequential loader, and the score matrix / run counters the REFERENCE derives from it.

DEV-CONTAINER ONLY (imports /root/reference).  Writes
    tests/golden/loader_mixed.bam      name-collated BAM, 18 fragments: single-end mapped / multi-mapped / unmapped,
                                       proper pairs (multi-block CIGARs with N, D, I, S), a proper pair whose mate is
                                       missing, a pair with both mates unmapped, mixed pairs (one mate unmapped), an
                                       improper pair, alignments below the overlap threshold, two alignments to one
                                       locus, a read over two loci on opposite strands
    tests/golden/loader_mixed.gtf      5 loci on two chromosomes and both strands, overlapping exons of one locus
    tests/golden/loader_mixed_expected.npz   per stranded mode: raw score matrix (CSR), row / column names, run counters

What produces the expectations:
  * the reference's own `alignment.fetch_fragments_seq` (bundling, mate pairing, fragment classes; alignment.py:
    115-161), `model.process_overlap_frag` (best alignment per locus; model.py:30-63) and
    `Telescope._mapping_to_matrix` (rescaling, max per cell, row dropping, counters; model.py:287-362), imported
    from /root/reference and run unmodified on plain-Python record objects that carry pysam's attribute names;
  * `calignment.AlignedPair` is an unbuilt Cython class (it cimports pysam's headers, which this image lacks) and
    `_annotation_intervaltree` needs the `intervaltree` package (absent): their few lines are restated below —
    calignment.pyx:60-98, helpers.py:74-104, _annotation_intervaltree.py:36-63,92-102, and the threshold rule of
    model.py:877-897 — independently of telescope_amd/loader.py (brute-force interval scan, no index).
The BAM is written by the small encoder below (BGZF blocks via zlib); it is data, not reference material.
"""
import os
import re
import struct
import sys
import zlib
from collections import Counter, OrderedDict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')
NOFEAT = '__no_feature'

REFS = [('chrA', 100000), ('chrB', 50000)]
GTF_ROWS = [
    # chrom, start, end (1-based inclusive), strand, locus
    ('chrA', 1000, 2000, '+', 'L1'),
    ('chrA', 5000, 5600, '-', 'L2'),
    ('chrA', 9000, 9400, '+', 'L3'), ('chrA', 9300, 9900, '+', 'L3'),      # overlapping exons of one locus: merged
    ('chrA', 1500, 2600, '-', 'L4'),                                       # overlaps L1 on the other strand
    ('chrB', 100, 900, '+', 'L5'),
    ('chrB', 4000, 4100, '+', 'L5'),                                       # second, disjoint exon of L5
]

OPS = {'M': 0, 'I': 1, 'D': 2, 'N': 3, 'S': 4, 'H': 5, 'P': 6, '=': 7, 'X': 8}
PAIRED, PROPER, UNMAP, MUNMAP, REV, MREV, R1, R2, SECONDARY = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40, 0x80, 0x100


def rec(q, flag, ref=-1, pos=-1, cigar='', AS=None, nref=-1, npos=-1, tlen=0):
    return dict(qname=q, flag=flag, ref_id=ref, pos=pos, cigar=cigar, AS=AS, nref=nref, npos=npos, tlen=tlen)


def pair(q, ref, p1, c1, as1, p2, c2, as2, tlen, rev1=False, extra=0):
    """A proper pair: read 1 at p1, read 2 at p2 (0-based)."""
    f1 = PAIRED | PROPER | R1 | (REV if rev1 else MREV) | extra
    f2 = PAIRED | PROPER | R2 | (MREV if rev1 else REV) | extra
    return [rec(q, f1, ref, p1, c1, as1, ref, p2, tlen), rec(q, f2, ref, p2, c2, as2, ref, p1, -tlen)]


def build_records():
    R = []
    R += [rec('f01', 0, 0, 1100, '50M', -5)]                                            # SM, unique, L1/L4 (+ read)
    R += [rec('f02', 0, 0, 1200, '50M', -3), rec('f02', SECONDARY, 0, 5100, '50M', -9),
          rec('f02', SECONDARY | REV, 0, 30000, '50M', -12)]                            # SM, ambiguous: L1, L2, intergenic
    R += [rec('f03', UNMAP)]                                                            # SU
    R += pair('f04', 0, 5050, '30M500N20M', -2, 5700, '50M', -4, 700)                   # PM unique, spliced read 1, L2
    R += pair('f05', 0, 1300, '20M3D30M', -6, 1500, '50M', 0, 250) + \
         pair('f05', 0, 9100, '50M', -10, 9350, '5S40M2I3M', -11, 300, extra=SECONDARY)  # PM ambiguous: L1 and L3
    R += pair('f06', 1, 200, '50M', -1, 400, '50M', -1, 250) + \
         [rec('f06', PAIRED | PROPER | R1 | SECONDARY, 1, 4000, '50M', -20, 1, 4200, 250)]   # PM, one alignment without its mate
    R += [rec('f07', PAIRED | UNMAP | MUNMAP | R1), rec('f07', PAIRED | UNMAP | MUNMAP | R2)]   # PU
    R += [rec('f08', PAIRED | MUNMAP | R1, 0, 1600, '50M', -7, 0, 1600, 0),
          rec('f08', PAIRED | UNMAP | R2, 0, 1600, '', None, 0, 1600, 0)]                # PX: read 2 unmapped
    R += [rec('f09', PAIRED | R1, 0, 1700, '50M', -8, 1, 300, 0),
          rec('f09', PAIRED | R2 | REV, 1, 300, '50M', -2, 0, 1700, 0)]                  # PX: improper pair, two hits
    R += pair('f10', 0, 40000, '50M', -1, 40200, '50M', -1, 250)                         # PM, no feature at all
    R += [rec('f11', 0, 0, 1050, '50M', -15), rec('f11', SECONDARY, 0, 1900, '50M', -4),
          rec('f11', SECONDARY, 0, 5500, '50M', -6)]                                     # two hits in L1 (best kept), one in L2
    R += [rec('f12', 0, 0, 955, '50M', 0)]                                               # 5 of 50 bases in L1: below 20 %
    R += [rec('f13', REV, 0, 1800, '50M', -2)]                                           # reverse read over L1 (+) and L4 (-)
    R += [rec('f14', 0, 1, 850, '40M3000N10M2I8M', -3)]                                  # spliced across both L5 exons
    R += pair('f15', 0, 1950, '50M', -3, 2100, '50M', -5, 200, rev1=True)                # pair, read 1 reverse, L1 end / L4
    R += [rec('f16', 0, 0, 9395, '10M', -1)]                                             # inside the merged L3 exon
    R += [rec('f17', PAIRED | MUNMAP | R1 | REV, 1, 120, '50M', -30, 1, 120, 0),
          rec('f17', PAIRED | UNMAP | R2, 1, 120, '', None, 1, 120, 0)]                  # PX, lowest score of the file
    R += [rec('f18', 0, 0, 5580, '50M', 0), rec('f18', SECONDARY, 0, 9880, '50M', 0)]    # ties in score: L2 (21 bases) vs L3 (20)
    return R


# ------------------------------------------------------------------------------------------------- BAM writer
def bam_bytes(records):
    out = bytearray(b'BAM\x01')
    text = '@HD\tVN:1.6\tSO:unsorted\tGO:query\n' + ''.join('@SQ\tSN:%s\tLN:%d\n' % r for r in REFS)
    out += struct.pack('<i', len(text)) + text.encode()
    out += struct.pack('<i', len(REFS))
    for name, ln in REFS:
        out += struct.pack('<i', len(name) + 1) + name.encode() + b'\x00' + struct.pack('<i', ln)
    for r in records:
        cig = [(int(n), OPS[o]) for n, o in re.findall(r'(\d+)([MIDNSHP=X])', r['cigar'])]
        l_seq = sum(n for n, o in cig if o in (0, 1, 4, 7, 8)) or 30
        qn = r['qname'].encode() + b'\x00'
        body = struct.pack('<iiBBHHHiiii', r['ref_id'], r['pos'], len(qn), 30, 4680, len(cig), r['flag'], l_seq,
                           r['nref'], r['npos'], r['tlen'])
        body += qn + b''.join(struct.pack('<I', (n << 4) | o) for n, o in cig)
        body += bytes([0x11] * ((l_seq + 1) // 2)) + bytes([30] * l_seq)                 # sequence 'A...' , qualities
        body += b'NMC\x00'
        if r['AS'] is not None:
            body += b'ASi' + struct.pack('<i', r['AS'])                                  # (one width is enough for the parser)
        body += b'XSZ' + b'note\x00' + b'ZBBs' + struct.pack('<ihh', 2, 1, 2)           # a Z tag and a B array to skip over
        out += struct.pack('<i', len(body)) + body
    return bytes(out)


def bgzf(data, block=20000):
    out = bytearray()
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + [b'']             # the empty block is BGZF's EOF marker
    for ch in chunks:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(ch) + co.flush()
        bsize = len(comp) + 25
        out += b'\x1f\x8b\x08\x04' + struct.pack('<IBBH', 0, 0, 0xff, 6) + b'BC' + struct.pack('<HH', 2, bsize)
        out += comp + struct.pack('<II', zlib.crc32(ch) & 0xffffffff, len(ch))
    return bytes(out)


# ------------------------------------------------------------------------- plain record objects with pysam's names
class Seg(object):
    def __init__(self, r):
        self.__dict__.update(query_name=r['qname'], flag=r['flag'], reference_id=r['ref_id'],
                             reference_start=r['pos'], next_reference_id=r['nref'], next_reference_start=r['npos'],
                             template_length=r['tlen'], _cigar=r['cigar'], _AS=r['AS'], _tags={})
        self.reference_name = REFS[r['ref_id']][0] if r['ref_id'] >= 0 else None
    is_paired = property(lambda s: bool(s.flag & PAIRED))
    is_proper_pair = property(lambda s: bool(s.flag & PROPER))
    is_unmapped = property(lambda s: bool(s.flag & UNMAP))
    is_reverse = property(lambda s: bool(s.flag & REV))
    is_read1 = property(lambda s: bool(s.flag & R1))

    def get_blocks(self):
        out, pos = [], self.reference_start
        for n, o in re.findall(r'(\d+)([MIDNSHP=X])', self._cigar):
            n = int(n)
            if o in 'M=X':
                out.append((pos, pos + n)); pos += n
            elif o in 'DN':
                pos += n
        return out

    def get_tag(self, t):
        if t == 'AS':
            if self._AS is None:
                raise KeyError(t)
            return self._AS
        return self._tags[t]

    def set_tag(self, t, v, *a, **k):
        self._tags[t] = v


def merge_blocks(ivs, dist=0):                           # helpers.py:74-104
    if len(ivs) <= 1:
        return list(ivs)
    ivs = sorted(ivs, key=lambda x: x[0])
    ret = [ivs[0]]
    for iv in ivs[1:]:
        if iv[0] - ret[-1][1] > dist:
            ret.append(iv)
        else:
            ret[-1] = (ret[-1][0], max(iv[1], ret[-1][1]))
    return ret


class AlignedPair(object):                               # calignment.pyx:60-98 (unbuildable Cython: restated)
    def __init__(self, r1, r2=None):
        self.r1, self.r2 = r1, r2
    is_paired = property(lambda s: s.r2 is not None)
    is_unmapped = property(lambda s: s.r1.is_unmapped)
    r1_is_reversed = property(lambda s: s.r1.is_reverse)
    ref_name = property(lambda s: s.r1.reference_name)
    query_id = property(lambda s: s.r1.query_name)

    @property
    def refblocks(self):
        b = self.r1.get_blocks() + (self.r2.get_blocks() if self.r2 is not None else [])
        return merge_blocks(b, 1)
    alnlen = property(lambda s: sum(b[1] - b[0] for b in s.refblocks))
    alnscore = property(lambda s: s.r1.get_tag('AS') + (s.r2.get_tag('AS') if s.r2 is not None else 0))

    def set_tag(self, t, v, *a, **k):
        self.r1.set_tag(t, v)
        if self.r2 is not None:
            self.r2.set_tag(t, v)


class BruteAnnotation(object):                            # _annotation_intervaltree.py:36-63,92-102 without the tree
    def __init__(self, rows, stranded_mode):
        self.run_stranded = stranded_mode != 'None'
        self.ivs = []                                     # [chrom, begin, end, locus, strand]
        for chrom, s, e, strand, loc in rows:
            b, en = s, e + 1
            hit = [iv for iv in self.ivs if iv[0] == chrom and iv[3] == loc and iv[1] < en and b < iv[2]]
            if hit:
                assert len(hit) == 1
                self.ivs.remove(hit[0])
                b, en = min(b, hit[0][1]), max(en, hit[0][2])
            self.ivs.append([chrom, b, en, loc, strand])

    def intersect_blocks(self, ref, blocks, frag_strand):
        res = Counter()
        for bs, be in blocks:
            qb, qe = bs, be + 1
            for chrom, b, e, loc, strand in self.ivs:
                if chrom == ref and b < qe and qb < e and (not self.run_stranded or strand == frag_strand):
                    res[loc] += min(e, qe) - max(b, qb)
        return res

    def feature_length(self):
        ret = Counter()
        for chrom, b, e, loc, strand in self.ivs:
            ret[loc] += e - b
        return ret


def load_reference_loader():
    from tools.ref_import import load_reference
    Telescope, _, _ = load_reference()
    sys.modules['telescope.utils.calignment'].AlignedPair = AlignedPair
    from telescope.utils import alignment, model
    alignment.AlignedPair = AlignedPair
    return Telescope, alignment, model


class SamStub(object):
    def __init__(self, records):
        self.records = records

    def fetch(self, **kw):
        return iter([Seg(r) for r in self.records])


def expected(records, stranded_mode, Telescope, alignment, model, threshold=0.2):
    annot = BruteAnnotation(GTF_ROWS, stranded_mode)

    def assign(pair):                                     # model.py:877-897, threshold mode
        if pair.r1_is_reversed:
            strand = ('+' if stranded_mode[-1] == 'F' else '-') if pair.is_paired else ('-' if stranded_mode[0] == 'F' else '+')
        else:
            strand = ('-' if stranded_mode[-1] == 'F' else '+') if pair.is_paired else ('+' if stranded_mode[0] == 'F' else '-')
        f = annot.intersect_blocks(pair.ref_name, pair.refblocks, strand)
        if not f:
            return NOFEAT
        fname, overlap = f.most_common()[0]
        return fname if overlap > pair.alnlen * threshold else NOFEAT

    info, mappings = Counter(), []
    min_as, max_as = 2 ** 32 - 1, -(2 ** 32 - 1)
    for ci, alns in alignment.fetch_fragments_seq(SamStub(records), until_eof=True):      # the reference's own bundling / pairing
        info['total_fragments'] += 1
        code = alignment.CODES[ci][0]
        info[code] += 1
        if code in ('SU', 'PU'):
            continue
        mapped = [a for a in alns if not a.is_unmapped]
        ambig = len(mapped) > 1
        scores = [a.alnscore for a in mapped]
        min_as, max_as = min(min_as, *scores), max(max_as, *scores)
        feats = list(map(assign, mapped))
        if not any(f != NOFEAT for f in feats):
            info['nofeat_%s' % ('A' if ambig else 'U')] += 1
            continue
        info['feat_%s' % ('A' if ambig else 'U')] += 1
        for m in model.process_overlap_frag(mapped, feats):                              # the reference's own
            mappings.append((ci, m[0], m[1], m[2], m[3]))

    class O(object):
        no_feature_key = NOFEAT
    ts = Telescope.__new__(Telescope)
    ts.opts, ts.single_cell, ts.read_index, ts.feat_index, ts.run_info = O(), False, {}, {}, {}
    Telescope._mapping_to_matrix(ts, iter(mappings), (min_as, max_as), info)               # the reference's own
    raw = ts.raw_scores.tocsr()
    raw.sort_indices()
    # _mapping_to_matrix rebinds the trimmed row index to a local; rebuild the names of the kept rows the same way
    rownames = np.array(sorted(ts.read_index, key=ts.read_index.get))
    full = {}
    for code, rid, fid, ascr, alen in mappings:
        full.setdefault(rid, set()).add(fid)
    kept = [r for r in rownames if any(f != NOFEAT for f in full[r])]
    fields = ['total_fragments', 'pair_mapped', 'pair_mixed', 'single_mapped', 'unmapped', 'unique', 'ambig',
              'overlap_unique', 'overlap_ambig']
    return dict(data=raw.data.astype(np.uint16), indices=raw.indices.astype(np.int32), indptr=raw.indptr.astype(np.int64),
                shape=np.array(raw.shape), rows=np.array(kept), cols=np.array(sorted(ts.feat_index, key=ts.feat_index.get)),
                info=np.array([int(info[f]) for f in fields]), score_range=np.array([min_as, max_as]),
                feature_length=np.array([annot.feature_length()[c] for c in sorted(ts.feat_index, key=ts.feat_index.get)]))


def main():
    records = build_records()
    os.makedirs(GOLD, exist_ok=True)
    with open(os.path.join(GOLD, 'loader_mixed.bam'), 'wb') as f:
        f.write(bgzf(bam_bytes(records)))
    with open(os.path.join(GOLD, 'loader_mixed.gtf'), 'w') as f:
        f.write('# synthetic annotation for tests/test_loader_mixed.py (tools/make_loader_fixture.py)\n')
        for chrom, s, e, strand, loc in GTF_ROWS:
            f.write('%s\tsynthetic\texon\t%d\t%d\t.\t%s\t.\tgene_id "%s"; transcript_id "%s"; locus "%s";\n'
                    % (chrom, s, e, strand, loc, loc, loc))
        f.write('chrA\tsynthetic\tgene\t1\t99999\t.\t+\t.\tgene_id "skipme"; locus "skipme";\n')   # not an exon: ignored
    Telescope, alignment, model = load_reference_loader()
    out = {}
    for mode in ('None', 'F', 'R', 'FR', 'RF'):
        e = expected(records, mode, Telescope, alignment, model)
        for k, v in e.items():
            out['%s_%s' % (mode, k)] = v
        print(mode, 'matrix %s nnz %d' % (tuple(e['shape']), len(e['data'])), 'info', e['info'].tolist(),
              'score range', e['score_range'].tolist())
    np.savez_compressed(os.path.join(GOLD, 'loader_mixed_expected.npz'), **out)


if __name__ == '__main__':
    main()
