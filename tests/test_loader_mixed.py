"""Loader coverage: every fragment class of the reference's sequential loader — single-end
mapped / multi-mapped / unmapped, proper pairs incl. one whose mate is missing, pairs with both mates unmapped,
mixed and improper pairs, spliced / clipped / indel CIGARs, hits below the overlap threshold, two hits in one
locus, all five `--stranded_mode` settings — on a synthetic BAM + GTF.  The expected matrices and run counters
were produced by the REFERENCE's own `fetch_fragments_seq`, `process_overlap_frag` and `_mapping_to_matrix`
(tools/make_loader_fixture.py, dev container); this test runs telescope_amd/loader.py on the same files."""
import os

import numpy as np
import pytest

from conftest import GOLD

BAM = os.path.join(GOLD, 'loader_mixed.bam')
GTF = os.path.join(GOLD, 'loader_mixed.gtf')
FIELDS = ['total_fragments', 'pair_mapped', 'pair_mixed', 'single_mapped', 'unmapped', 'unique', 'ambig',
          'overlap_unique', 'overlap_ambig']


@pytest.mark.parametrize('mode', ['None', 'F', 'R', 'FR', 'RF'])
def test_loader_matches_reference_on_mixed_fragments(mode):
    from telescope_amd import loader
    exp = np.load(os.path.join(GOLD, 'loader_mixed_expected.npz'), allow_pickle=False)
    g = lambda k: exp['%s_%s' % (mode, k)]
    ann = loader.Annotation(GTF, 'locus', mode)
    out = loader.load_alignment(BAM, ann, stranded_mode=mode)
    raw = out['raw_scores'].tocsr()
    raw.sort_indices()
    assert tuple(raw.shape) == tuple(int(x) for x in g('shape'))
    assert np.array_equal(raw.indptr, g('indptr')) and np.array_equal(raw.indices, g('indices'))
    assert raw.data.dtype == np.uint16 and np.array_equal(raw.data, g('data'))
    assert [r for r, _ in sorted(out['read_index'].items(), key=lambda kv: kv[1])] == list(g('rows'))
    assert [c for c, _ in sorted(out['feat_index'].items(), key=lambda kv: kv[1])] == list(g('cols'))
    assert [int(out['run_info'][f]) for f in FIELDS] == g('info').tolist()
    assert list(out['score_range']) == g('score_range').tolist()
    assert [out['feature_length'][c] for c in g('cols')] == g('feature_length').tolist()


def test_bam_reader_streams_records_and_skips_foreign_tags():
    from telescope_amd import loader
    refs, records = loader.read_bam(BAM)
    assert refs == ['chrA', 'chrB']
    recs = list(records)
    assert len(recs) == 35 and recs[0].qname == 'f01' and recs[-1].qname == 'f18'
    assert recs[0].AS == -5 and recs[0].blocks() == [(1100, 1150)]
    spliced = [r for r in recs if r.qname == 'f04'][0]
    assert spliced.blocks() == [(5050, 5080), (5580, 5600)]
    clipped = [r for r in recs if r.qname == 'f05'][-1]
    assert clipped.blocks() == [(9350, 9390), (9390, 9393)]         # 5S40M2I3M: the insertion splits the blocks, no gap
    assert [r for r in recs if r.qname == 'f03'][0].is_unmapped


def test_annotation_merges_overlapping_exons_of_one_locus_only():
    from telescope_amd import loader
    ann = loader.Annotation(GTF, 'locus', 'None')
    fl = ann.feature_length()
    assert fl['L3'] == 9900 + 1 - 9000          # two overlapping exons merged into one interval [9000, 9901)
    assert fl['L5'] == (900 + 1 - 100) + (4100 + 1 - 4000)
    assert fl['L1'] == 1001 and fl['L4'] == 1101 and 'skipme' not in fl    # L1 / L4 overlap but are different loci
