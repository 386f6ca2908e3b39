#!/usr/bin/env python3
"""Regenerate the config-1 ("telescope test") score matrix WITHOUT pysam.

DEV-CONTAINER ONLY.  Reads the reference's bundled *data* files
(/root/reference/telescope/data/alignment.bam + annotation.gtf) and restates
the reference's sequential loader rules to produce `raw_scores` (1000 x 59
uint16 CSR, nnz 18471).  The result is committed as
tests/golden/bundled_raw_scores.npz — a data fixture; no reference source is
copied.  Rules restated (file:line relative to /root/reference/telescope):

  utils/alignment.py:115-161   bundle by query name, pair mates by read/mate key
  utils/calignment.pyx:83-98   refblocks (merge_blocks(.,1)), alnlen, alnscore
  utils/helpers.py:74-104      merge_blocks
  utils/_annotation_intervaltree.py:36-63,92-102   exon merge + block overlap
  utils/model.py:877-897       threshold assignment (overlap > alnlen*0.2)
  utils/model.py:30-63         best alignment per locus (max alnscore+alnlen)
  utils/model.py:294-308,347-359  rescale, max per (frag, locus), drop rows
                               hitting only column 0, first-appearance ids
"""
import gzip
import os
import re
import struct
import sys
from collections import Counter, OrderedDict, defaultdict

import numpy as np

REF_DATA = '/root/reference/telescope/data'
NOFEAT = '__no_feature'


def read_bam(path):
    """Minimal BAM decoder (BGZF == concatenated gzip members)."""
    buf = gzip.open(path, 'rb').read()
    assert buf[:4] == b'BAM\x01'
    off = 4
    (l_text,) = struct.unpack_from('<i', buf, off); off += 4 + l_text
    (n_ref,) = struct.unpack_from('<i', buf, off); off += 4
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from('<i', buf, off); off += 4
        name = buf[off:off + l_name - 1].decode(); off += l_name
        (l_ref,) = struct.unpack_from('<i', buf, off); off += 4
        refs.append((name, l_ref))
    recs = []
    while off < len(buf):
        (bs,) = struct.unpack_from('<i', buf, off); off += 4
        end = off + bs
        (ref_id, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, nref, npos,
         tlen) = struct.unpack_from('<iiBBHHHiiii', buf, off)
        p = off + 32
        qname = buf[p:p + l_rn - 1].decode(); p += l_rn
        cigar = struct.unpack_from('<%dI' % n_cig, buf, p); p += 4 * n_cig
        p += (l_seq + 1) // 2 + l_seq
        AS = None
        while p < end:
            tag = buf[p:p + 2]; typ = chr(buf[p + 2]); p += 3
            if typ in 'cCsSiIf':
                fmt = {'c': 'b', 'C': 'B', 's': 'h', 'S': 'H', 'i': 'i',
                       'I': 'I', 'f': 'f'}[typ]
                (val,) = struct.unpack_from('<' + fmt, buf, p)
                p += struct.calcsize(fmt)
                if tag == b'AS':
                    AS = val
            elif typ == 'A':
                p += 1
            elif typ in 'ZH':
                e = buf.index(b'\x00', p); p = e + 1
            elif typ == 'B':
                sub = chr(buf[p]); (cnt,) = struct.unpack_from('<i', buf, p + 1)
                p += 5 + cnt * {'c': 1, 'C': 1, 's': 2, 'S': 2, 'i': 4,
                                'I': 4, 'f': 4}[sub]
            else:
                raise ValueError('tag type ' + typ)
        recs.append(dict(qname=qname, flag=flag, ref_id=ref_id, pos=pos,
                         nref=nref, npos=npos, tlen=tlen, cigar=cigar, AS=AS))
        off = end
    return refs, recs


def blocks_of(rec):
    """Gapless reference blocks: M/=/X emit+advance, D/N advance only."""
    out = []
    pos = rec['pos']
    for c in rec['cigar']:
        ln, op = c >> 4, c & 0xF
        if op in (0, 7, 8):
            out.append((pos, pos + ln)); pos += ln
        elif op in (2, 3):
            pos += ln
    return out


def merge_blocks(ivs, dist):
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


class Pair:
    __slots__ = ('r1', 'r2', 'refblocks', 'alnlen', 'alnscore')

    def __init__(self, r1, r2=None):
        self.r1, self.r2 = r1, r2
        b = blocks_of(r1) + (blocks_of(r2) if r2 is not None else [])
        self.refblocks = merge_blocks(b, 1)
        self.alnlen = sum(e - s for s, e in self.refblocks)
        self.alnscore = r1['AS'] + (r2['AS'] if r2 is not None else 0)


def load_annotation(gtf):
    """Per chrom list of [begin, end, locus]; same-locus overlapping exons merged."""
    itree = defaultdict(list)
    loci = OrderedDict()
    for line in open(gtf):
        if line.startswith('#'):
            continue
        f = line.rstrip('\n').split('\t')
        if f[2] != 'exon':
            continue
        attr = dict(re.findall(r'(\w+)\s+"(.+?)";', f[8]))
        if 'locus' not in attr:
            continue
        loc = attr['locus']
        loci.setdefault(loc, 0)
        b, e = int(f[3]), int(f[4]) + 1
        ivs = itree[f[0]]
        hit = [iv for iv in ivs if iv[0] < e and b < iv[1] and iv[2] == loc]
        if hit:
            assert len(hit) == 1
            ivs.remove(hit[0])
            b, e = min(b, hit[0][0]), max(e, hit[0][1])
        ivs.append([b, e, loc])
    return itree, loci


def main(out_path):
    refs, recs = read_bam(os.path.join(REF_DATA, 'alignment.bam'))
    itree, loci = load_annotation(os.path.join(REF_DATA, 'annotation.gtf'))

    def assign(pair):
        ref = refs[pair.r1['ref_id']][0]
        res = Counter()
        for bs, be in pair.refblocks:
            qb, qe = bs, be + 1
            for b, e, loc in itree.get(ref, ()):
                if b < qe and qb < e:
                    res[loc] += max(0, min(e, qe) - max(b, qb))
        if not res:
            return NOFEAT
        fname, ov = res.most_common()[0]
        return fname if ov > pair.alnlen * 0.2 else NOFEAT

    # bundle by consecutive query name
    bundles, cur = [], [recs[0]]
    for r in recs[1:]:
        if r['qname'] == cur[0]['qname']:
            cur.append(r)
        else:
            bundles.append(cur); cur = [r]
    bundles.append(cur)

    mappings = []
    minAS, maxAS = 2 ** 32 - 1, -(2 ** 32 - 1)
    info = Counter()
    for alns in bundles:
        info['total_fragments'] += 1
        f0 = alns[0]['flag']
        assert f0 & 0x1 and f0 & 0x2, 'bundled data is all proper pairs'
        info['pair_mapped'] += 1
        cache, pairs = {}, []
        for a in alns:
            is_r1 = bool(a['flag'] & 0x40)
            rk = (a['qname'], is_r1, a['ref_id'], a['pos'], a['nref'],
                  a['npos'], abs(a['tlen']))
            mk = (a['qname'], not is_r1, a['nref'], a['npos'], a['ref_id'],
                  a['pos'], abs(a['tlen']))
            mate = cache.pop(mk, None)
            if mate is not None:
                pairs.append(Pair(a, mate) if is_r1 else Pair(mate, a))
            else:
                cache[rk] = a
        pairs += [Pair(a) for a in cache.values()]
        mapped = [p for p in pairs if not (p.r1['flag'] & 0x4)]
        ambig = len(mapped) > 1
        for p in mapped:
            minAS, maxAS = min(minAS, p.alnscore), max(maxAS, p.alnscore)
        feats = [assign(p) for p in mapped]
        if not any(f != NOFEAT for f in feats):
            info['nofeat_A' if ambig else 'nofeat_U'] += 1
            continue
        info['feat_A' if ambig else 'feat_U'] += 1
        byfeat = OrderedDict()
        for p, f in zip(mapped, feats):
            byfeat.setdefault(f, []).append(p)
        maps = []
        for f, fal in byfeat.items():
            fal.sort(key=lambda x: x.alnscore + x.alnlen, reverse=True)
            maps.append((alns[0]['qname'], f, fal[0].alnscore, fal[0].alnlen))
        maps.sort(key=lambda x: x[2], reverse=True)
        mappings += maps

    ridx, fidx = OrderedDict(), OrderedDict([(NOFEAT, 0)])
    cells = {}
    for rid, fid, ascr, alen in mappings:
        i = ridx.setdefault(rid, len(ridx))
        j = fidx.setdefault(fid, len(fidx))
        v = (ascr - minAS + 1) + alen
        cells[(i, j)] = max(cells.get((i, j), 0), v)
    nrow, ncol = len(ridx), len(fidx)
    keep = sorted({i for (i, j) in cells if j != 0})
    remap = {old: new for new, old in enumerate(keep)}
    rows = [[] for _ in keep]
    for (i, j), v in cells.items():
        if i in remap:
            rows[remap[i]].append((j, v))
    indptr, indices, data = [0], [], []
    for r in rows:
        r.sort()
        indices += [j for j, _ in r]
        data += [v for _, v in r]
        indptr.append(len(indices))
    rnames = [n for n, i in ridx.items() if i in remap]
    fnames = list(fidx)
    # feature lengths: sum of merged exon interval lengths
    # (_annotation_intervaltree.py:65-76)
    flen = Counter()
    for ch in itree:
        for b, e, loc in itree[ch]:
            flen[loc] += e - b
    uniq = sum(1 for r in rows if len(r) == 1)
    run_info = OrderedDict([
        ('version', '1.0.3.1'),
        ('annotated_features', len(loci)),
        ('total_fragments', info['total_fragments']),
        ('pair_mapped', info['pair_mapped']), ('pair_mixed', 0),
        ('single_mapped', 0), ('unmapped', 0),
        ('unique', info['nofeat_U'] + info['feat_U']),
        ('ambig', info['nofeat_A'] + info['feat_A']),
        ('overlap_unique', uniq), ('overlap_ambig', len(rows) - uniq),
    ])
    data = np.asarray(data, dtype=np.uint16)
    print('shape', (len(rows), ncol), 'nnz', len(data), 'min', data.min(),
          'max', data.max(), 'minAS', minAS, 'maxAS', maxAS, dict(run_info))
    np.savez_compressed(
        out_path,
        data=data, indices=np.asarray(indices, dtype=np.int32),
        indptr=np.asarray(indptr, dtype=np.int32),
        shape=np.asarray((len(rows), ncol), dtype=np.int64),
        feat_names=np.asarray(fnames), read_names=np.asarray(rnames),
        feat_lengths=np.asarray([flen[f] for f in fnames], dtype=np.int64),
        run_info=np.asarray([(k, str(v)) for k, v in run_info.items()]),
    )


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        here, '..', 'tests', 'golden', 'bundled_raw_scores.npz')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    main(out)
