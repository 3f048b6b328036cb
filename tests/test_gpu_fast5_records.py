"""Row N2: what a resquiggled read leaves behind for downstream Tombo commands -- the FAST5 group
layout (attributes + Events dataset, tombo_helper.py:2341-2460) and the index record
(tombo_helper.py:1169-1183).  The writer here is run against the same dict-backed h5py stand-in
the reference's writer was recorded on (tests/golden/gen_golden_fast5.py): same paths, same
attribute values, same Events table."""
import os
import sys
import json

import numpy as np
import pytest

from conftest import GOLDEN_DIR

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fast5_group_equals_the_reference_writer():
    import memh5
    from tombo_amd import resquiggle as rq, synth, tombo_stats as ts, tombo_helper as th
    g = np.load(os.path.join(GOLDEN_DIR, 'stats_fast5_layout.npz'))
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    for case in (0, 1):
        m = json.loads(str(g['c%d_meta' % case]))
        seq, raw, _ = synth.synth_read(model, m['n_bases'], m['seed'], **synth.DNA_SYNTH)
        mr = th.resquiggleResults(
            align_info=th.alignInfo('read_%d' % m['seed'], 'BaseCalled_template', 3, 5, 7, 11,
                                    m['n_bases'] - 20, 9),
            genome_loc=th.genomeLocation(1234, '+', 'chr_synth'), genome_seq=seq,
            mean_q_score=11.5, raw_signal=raw)
        np.random.seed(m['seed'])
        res = rq.resquiggle_read(mr, model, params, 5.0, seq_samp_type=samp)
        # Events through the batch path (statistics computed on the resident batch) ...
        _, tabs = rq.resquiggle_batch_events([mr], model, params, outlier_thresh=5.0,
                                             seq_samp_type=samp, compute_sd=m['compute_sd'])
        f = memh5.MemGroup()
        f.create_group('Analyses/RawGenomeCorrected_000')
        th.write_new_fast5_group(f, 'RawGenomeCorrected_000', res, 'median', m['compute_sd'],
                                 event_data=tabs[0])
        # ... and through the writer's own call of the c_new_mean_stds / c_new_means kernels
        f2 = memh5.MemGroup()
        f2.create_group('Analyses/RawGenomeCorrected_000')
        th.write_new_fast5_group(f2, 'RawGenomeCorrected_000', res, 'median', m['compute_sd'])
        for tree in (memh5.tree(f), memh5.tree(f2)):
            assert sorted(tree) == json.loads(str(g['c%d_keys' % case]))
            for k, v in tree.items():
                name = 'c%d|%s' % (case, k)
                if isinstance(v, np.ndarray) and v.dtype.names:
                    assert list(v.dtype.names) == ['norm_mean', 'norm_stdev', 'start', 'length', 'base']
                    assert [v.dtype[n].str for n in v.dtype.names] == ['<f8', '<f8', '<u4', '<u4', '|S1']
                    for fld in v.dtype.names:
                        want = g[name + '|' + fld]
                        if v.dtype[fld].kind == 'f':
                            np.testing.assert_array_equal(np.isnan(v[fld]), np.isnan(want))
                            np.testing.assert_array_equal(np.nan_to_num(v[fld]), np.nan_to_num(want))
                        else:
                            np.testing.assert_array_equal(v[fld], want)
                else:
                    want = g[name]
                    if want.dtype.kind in 'US':
                        assert str(v) == str(want), k
                    else:
                        assert np.asarray(v) == want, k


def test_index_record_round_trip(tmp_path):
    """index tuple layout (tombo_helper.py:1169-1183) and the pickle container"""
    import pickle
    from tombo_amd import tombo_helper as th
    rd = th.resquiggledRead(start=100, end=350, filtered=False, read_start_rel_to_raw=812,
                            strand='-', fn='/data/run1/reads/r1.fast5',
                            corr_group='RawGenomeCorrected_000/BaseCalled_template', rna=False,
                            sig_match_score=0.91, mean_q_score=10.3, read_id='abc-1')
    ent = th.index_entry(rd, basedir='/data/run1')
    assert ent == ('/reads/r1.fast5', 100, 350, 812, 'RawGenomeCorrected_000',
                   'BaseCalled_template', False, False, 0.91, 10.3, 'abc-1')
    fn = str(tmp_path / '.reads.RawGenomeCorrected_000.tombo.index')
    th.write_index({('chr1', '-'): [rd]}, fn, basedir='/data/run1')
    with open(fn, 'rb') as fp:
        raw = pickle.load(fp)
    assert raw == {('chr1', '-'): [ent]}
    back = th.read_index(fn, basedir='/data/run1')
    r0 = back[('chr1', '-')][0]
    assert r0 == th.readData(*rd[:11])
    assert th.readData._fields == ('start', 'end', 'filtered', 'read_start_rel_to_raw', 'strand',
                                   'fn', 'corr_group', 'rna', 'sig_match_score', 'mean_q_score',
                                   'read_id')


def test_read_from_results_fields():
    import numpy as np
    from tombo_amd import tombo_helper as th
    res = th.resquiggleResults(
        align_info=th.alignInfo('rid', 'BaseCalled_template', 0, 0, 0, 0, 10, 0),
        genome_loc=th.genomeLocation(50, '+', 'c'), genome_seq='ACGTACGTAC', mean_q_score=9.0,
        raw_signal=None, read_start_rel_to_raw=7, segs=np.arange(11) * 5,
        sig_match_score=0.5)
    rd = th.read_from_results(res, np.zeros(10), fn='x.fast5')
    assert (rd.start, rd.end, rd.strand, rd.read_start_rel_to_raw) == (50, 60, '+', 7)
    assert rd.corr_group == 'RawGenomeCorrected_000/BaseCalled_template' and rd.read_id == 'rid'
    assert rd.seq == 'ACGTACGTAC' and rd.means.shape == (10,)
