"""the host glue around the aligner call (tombo_amd/mapping.py) against what
the live reference made of the same FAST5 trees and scripted hits
(tests/golden/gen_golden_map.py -> map_cases.json).  CPU only: nothing here touches the engine."""
import os
import sys
import json

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'golden'))

import gen_golden_map as gm  # noqa: E402  (input builders only; the reference is not imported)
from tombo_amd import mapping, tombo_helper as th, tombo_stats as ts  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, 'golden', 'map_cases.json')))
RECORDS, CASES = gm.cases()


def _std_ref(samp_name):
    return ts.TomboModel(seq_samp_type=th.seqSampleType(samp_name, samp_name == 'RNA'))


@pytest.mark.parametrize('name', sorted(CASES))
def test_map_read_equals_reference(name):
    case, want = CASES[name], GOLD[name]['map_read']
    samp = th.seqSampleType(case['samp'], case['samp'] == 'RNA')
    f, al = gm.build(case, RECORDS)
    kw = dict(q_score_thresh=case['kw'].get('q_score_thresh', 0), seq_len_rng=case['kw'].get('seq_len_rng'))
    if 'error' in want:
        with pytest.raises(th.TomboError) as ei:
            mapping.map_read(f, al, _std_ref(case['samp']), samp, **kw)
        assert str(ei.value) == want['error']
        return
    if 'unexpected' in want:
        # aligner.seq() yields nothing: the reference means to raise 'Invalid mapping location'
        # but constructs a TomboReads instead of a TomboError (resquiggle.py:1361) and dies with
        # an unrelated exception; here the intended TomboError is raised
        with pytest.raises(th.TomboError) as ei:
            mapping.map_read(f, al, _std_ref(case['samp']), samp, **kw)
        assert str(ei.value) == 'Invalid mapping location'
        return
    mr = mapping.map_read(f, al, _std_ref(case['samp']), samp, **kw)
    assert list(mr.align_info) == want['align_info']
    assert list(mr.genome_loc) == want['genome_loc']
    assert mr.genome_seq == want['genome_seq']
    assert float(mr.mean_q_score) == want['mean_q_score']
    assert mr.start_clip_bases == want['start_clip_bases']
    assert al.n_drained == want['drained']                      # the hit iterator was exhausted
    assert mr.raw_signal is None and mr.segs is None             # signal fields stay empty


@pytest.mark.parametrize('name', sorted(CASES))
def test_fast5_read_filters_and_index_record(name):
    """read_fast5_for_mapping + filter_and_index_record = the reference's _io_and_map_read around
    the resquiggle call (which the generator answered with a scripted result)"""
    case = CASES[name]
    samp = th.seqSampleType(case['samp'], case['samp'] == 'RNA')
    for want in GOLD[name]['io']:
        f, al = gm.build(case, RECORDS)
        kw = dict(q_score_thresh=case['kw'].get('q_score_thresh', 0), seq_len_rng=case['kw'].get('seq_len_rng'),
                  sig_len_rng=case['kw'].get('sig_len_rng'))
        expect_err = want['raised'] or (want['failed'][0][0] if want['failed'] and not want['index'] else None)
        if expect_err is not None:
            with pytest.raises(th.TomboError) as ei:
                mapping.read_fast5_for_mapping(f, al, _std_ref(case['samp']), samp, **kw)
            if expect_err != 'unexpected':
                assert str(ei.value) == expect_err
            continue
        mr = mapping.read_fast5_for_mapping(f, al, _std_ref(case['samp']), samp, **kw)
        assert mr.raw_signal.dtype == np.int16 and mr.raw_signal.shape[0] == case['n_raw']
        assert mr.channel_info == th.channelInfo(10.0, 1400.5, 8192.0, b'17', 4000)
        # the scripted resquiggle result of the generator
        nb = len(mr.genome_seq) - 5
        segs = np.arange(nb + 1, dtype=np.int64) * want['step']
        res = mr._replace(read_start_rel_to_raw=123, segs=segs, sig_match_score=want['score'])
        of = None if want['obs_filter'] is None else [tuple(x) for x in want['obs_filter']]
        chrm, strand, rd = mapping.filter_and_index_record(
            res, 'dir/read_%d.fast5' % case['seed'], 'RawGenomeCorrected_000', 'BaseCalled_template',
            samp, 1.1, of)
        got = [chrm, strand] + [x.item() if hasattr(x, 'item') else x for x in rd]
        assert got == want['index'][0]


def test_read_id_fallbacks_and_missing_channel():
    case = CASES['dna_plus']
    samp = th.seqSampleType('DNA', False)
    f, al = gm.build(case, RECORDS)
    del f['/Raw/Reads'].items['Read_%d' % case['seed']].attrs['read_id']
    del f.items['UniqueGlobalKey']
    mr = mapping.read_fast5_for_mapping(f, al, _std_ref('DNA'), samp)
    assert mr.align_info.ID == str(case['seed'])    # read_num stands in for a missing read_id
    assert mr.channel_info is None                   # channel information is optional
    f2 = gm.memh5.MemGroup()
    with pytest.raises(th.TomboError) as ei:
        mapping.read_fast5_for_mapping(f2, al, _std_ref('DNA'), samp)
    assert 'Raw data is not found' in str(ei.value)


def test_sequence_helpers():
    assert th.rev_comp('AACGT') == 'ACGTT' and th.comp_seq('ACGTN') == 'TGCAN'
    assert th.invalid_seq('ACGU') and not th.invalid_seq('ACGT')
    assert th.rev_transcribe('ACGUU') == 'ACGTT'
    assert th.get_mean_q_score('5I#') == np.mean([20, 40, 2])


def test_unexpected_per_read_errors_do_not_abort_the_batch():
    """_io_and_map_read records any non-Tombo exception of a read as (traceback, 'subgrp:::fn',
    False) and goes on (resquiggle.py:1476-1479): a truncated Fastq, a missing Signal dataset and
    an aligner that raises are three failures, not an aborted batch"""
    import numpy as np
    import gen_golden_map as gm
    from scripted_aligner import ScriptedAligner
    from tombo_amd import mapping, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    trunc = gm.make_fast5('ACGTACGTACGT', '5' * 12, 500, 1)
    g = trunc['/Analyses/Basecall_1D_000/BaseCalled_template']
    del g.items['Fastq']
    g.create_dataset('Fastq', data=np.bytes_(b'@r1\nACGTACGTACGT'))      # no '+' / quality lines
    nosig = gm.make_fast5('ACGTACGTACGT', '5' * 12, 500, 2)
    del next(iter(nosig['/Raw/Reads'].values())).items['Signal']

    class Boom(ScriptedAligner):
        def map(self, *a, **k):
            raise OSError('aligner died')
    ok_but_boom = gm.make_fast5('ACGTACGTACGT', '5' * 12, 500, 3)
    index, failures = mapping.process_fast5_batch(
        [(trunc, 'a.fast5'), (nosig, 'b.fast5'), (ok_but_boom, 'c.fast5')], Boom({}, {}), model, params, samp)
    assert index == [] and len(failures) == 3
    assert [f[1] for f in failures] == ['BaseCalled_template:::%s.fast5' % c for c in 'abc']
    assert all(f[2] is False and 'Traceback' in f[0] for f in failures)
    assert 'aligner died' in failures[2][0]


def test_prep_step_status_and_overwrite():
    """th.prep_fast5 (tombo_helper.py:2259-2324) and th.write_error_status (:2326-2339) as
    process_fast5_batch applies them: no mapping or GPU work is needed for any of these reads"""
    import memh5
    import gen_golden_map as gm
    from scripted_aligner import ScriptedAligner
    from tombo_amd import mapping, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    nohit = gm.make_fast5('ACGTACGTACGT', '5' * 12, 500, 1)
    nobc = gm.make_fast5('ACGT', '5555', 500, 2, fastq=False)
    other_bc = gm.make_fast5('ACGTACGTACGT', '5' * 12, 500, 3)
    done = gm.make_fast5('ACGTACGTACGT', '5' * 12, 500, 4)
    done['/Analyses'].create_group('RawGenomeCorrected_000/BaseCalled_template').attrs['status'] = 'success'
    files = [(nohit, 'a.fast5'), (nobc, 'b.fast5'), (other_bc, 'c.fast5'), (done, 'd.fast5')]
    aligner = ScriptedAligner({}, {})
    index, failures = mapping.process_fast5_batch(files[:2] + files[3:], aligner, model, params, samp)
    assert index == []
    assert failures == [
        ('Alignment not produced', 'BaseCalled_template:::a.fast5', True),
        ('Base calls not found in FAST5 (see `tombo preprocess`)', 'b.fast5', True),
        ('Tombo data exists in [--corrected-group] and [--overwrite] is not set', 'd.fast5', True)]
    t = memh5.tree(nohit)
    assert t['/Analyses/RawGenomeCorrected_000@tombo_version'] == th.TOMBO_VERSION
    assert t['/Analyses/RawGenomeCorrected_000@basecall_group'] == 'Basecall_1D_000'
    assert t['/Analyses/RawGenomeCorrected_000/BaseCalled_template@status'] == 'Alignment not produced'
    assert memh5.tree(done)['/Analyses/RawGenomeCorrected_000/BaseCalled_template@status'] == 'success'
    # --overwrite: the old group goes, the new one carries this run's outcome
    index, failures = mapping.process_fast5_batch([(done, 'd.fast5')], aligner, model, params, samp,
                                                  overwrite=True)
    assert failures == [('Alignment not produced', 'BaseCalled_template:::d.fast5', True)]
    assert memh5.tree(done)['/Analyses/RawGenomeCorrected_000/BaseCalled_template@status'] == \
        'Alignment not produced'
    # another --basecall-group than the file holds
    index, failures = mapping.process_fast5_batch([(other_bc, 'c.fast5')], aligner, model, params, samp,
                                                  bc_grp='Basecall_1D_001')
    assert failures == [('Base calls not found in FAST5 (see `tombo preprocess`)', 'c.fast5', True)]
    # write=False (the reference's dry run): nothing is prepared or written
    fresh = gm.make_fast5('ACGTACGTACGT', '5' * 12, 500, 5)
    index, failures = mapping.process_fast5_batch([(fresh, 'e.fast5')], aligner, model, params, samp,
                                                  write=False)
    assert failures == [('Alignment not produced', 'BaseCalled_template:::e.fast5', True)]
    assert 'RawGenomeCorrected_000' not in fresh['/Analyses']
