"""Engine vs oracle off the default parameters: a hypothesis-driven sweep over the box the
reference's --segmentation-parameters (tombo/_option_parsers.py:375-385) and
--signal-align-parameters (:606-617) open up, plus outlier_thresh / max_raw_cpts /
skip_seq_scaling / const_scale.  The oracle is pinned on live-reference fixtures at 13 points of
this box (tests/golden/p_*.npz, tests/test_oracle_golden.py); here every drawn point must agree
bit for bit, stage by stage (compare_batch), including the failures."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

from conftest import FORMS, engine_dispatch, check_forms

pytestmark = pytest.mark.gpu


@st.composite
def param_points(draw):
    rna = draw(st.booleans())
    seg = (draw(st.integers(2, 40)), draw(st.integers(1, 9)), draw(st.integers(1, 3)),
           draw(st.integers(3, 20)))
    bw = draw(st.sampled_from([60, 100, 128, 200, 256, 300, 320, 321, 500, 700, 1000]))
    start_bw = draw(st.sampled_from([200, 400, 750, 1000]))
    aln = (draw(st.floats(2.0, 7.0)), draw(st.floats(1.0, 7.0)), bw, 1500,
           draw(st.sampled_from([2.0, 5.0, 10.0, 20.0, 50.0])), draw(st.integers(0, 60)),
           start_bw, draw(st.sampled_from([start_bw, 1500, 2500])), draw(st.sampled_from([60, 150, 250])))
    return dict(rna=rna, seg=seg, aln=aln,
                outlier=draw(st.sampled_from([None, 2.0, 3.0, 5.0, 8.0])),
                max_raw_cpts=draw(st.sampled_from([None, 3, 30, 200])),
                skip=draw(st.booleans()), const_scale=draw(st.sampled_from([None, None, 9.0])),
                seed=draw(st.integers(0, 10 ** 6)), n_bases=draw(st.sampled_from([120, 300, 700, 1300])))


@settings(max_examples=200, deadline=None, derandomize=True,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(pt=param_points())
@pytest.mark.parametrize('form', FORMS)
def test_engine_matches_oracle_over_the_parameter_box(form, pt):
    with engine_dispatch(form):
        _box_point(form, pt)


def _box_point(form, pt):
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    from test_gpu_parity import run_batch, compare_batch
    name = 'RNA' if pt['rna'] else 'DNA'
    if pt['rna'] and pt['outlier'] is None and pt['const_scale'] is None:
        pt = dict(pt, outlier=4.0)   # a TypeError in the reference (tombo_stats.py:228); own test
    samp = th.seqSampleType(name, False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp, pt['aln'], pt['seg'])
    kw = dict(synth.RNA_SYNTH if pt['rna'] else synth.DNA_SYNTH)
    reads = []
    for k in range(3):
        nb = pt['n_bases'] + 37 * k
        seq, raw, _ = synth.synth_read(model, nb, pt['seed'] + k, **kw)
        rs = np.random.RandomState(pt['seed'] + k)
        si = rs.choice(nb, 1000, replace=False).astype(np.int64) if nb > 1000 else None
        reads.append((raw, seq, None, si))
    eng, out, oracles = run_batch(model, params, name, reads, outlier_thresh=pt['outlier'],
                                  const_scale=pt['const_scale'], skip_seq_scaling=pt['skip'],
                                  max_raw_cpts=pt['max_raw_cpts'])
    bad = compare_batch(eng, oracles, out, repr(pt))
    assert not bad, '\n'.join(bad[:20])
    check_forms(eng, form, params)


@st.composite
def disagreeing_reads(draw):
    kind = draw(st.sampled_from(['cut', 'insert', 'truncate', 'none']))
    edit = None
    if kind == 'cut':
        edit = dict(kind='cut', n=draw(st.integers(3, 90)))
    elif kind == 'insert':
        edit = dict(kind='insert', n=draw(st.integers(3, 90)), seed=draw(st.integers(0, 99)))
    elif kind == 'truncate':
        edit = dict(kind='truncate', frac=draw(st.sampled_from([0.5, 0.8, 0.9, 0.95, 0.99])))
    return dict(rna=draw(st.booleans()), edit=edit, seed=draw(st.integers(0, 10 ** 6)),
                n_bases=draw(st.sampled_from([260, 600, 1100, 1700])),
                dwell=draw(st.sampled_from([None, 0.25, 0.4, 1.6, 3.0])),   # x the sample type's mean dwell
                noise_sd=draw(st.sampled_from([0.15, 0.25, 0.5, 0.9])),
                lead=draw(st.sampled_from([None, 20, 2500, 7000])),
                bw=draw(st.sampled_from([100, 300, 500])), bbt=draw(st.sampled_from([10, 40])))


@settings(max_examples=150, deadline=None, derandomize=True,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(pt=disagreeing_reads())
@pytest.mark.parametrize('form', FORMS)
def test_engine_matches_oracle_on_reads_that_disagree_with_their_sequence(form, pt):
    with engine_dispatch(form):
        _disagreeing_point(form, pt)


def _disagreeing_point(form, pt):
    """deletions / insertions against the mapped sequence, truncated signal, dwell and noise far
    from the model's, long leaders: the paths that end in resolve_skipped_bases_with_raw, the
    start retry and the failure strings (the oracle is pinned on 11 such reads recorded from the
    live reference, tests/golden/e_*.npz)"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    from test_gpu_parity import run_batch, compare_batch
    name = 'RNA' if pt['rna'] else 'DNA'
    samp = th.seqSampleType(name, False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=pt['bw'], band_bound_thresh=pt['bbt'])
    kw = dict(synth.RNA_SYNTH if pt['rna'] else synth.DNA_SYNTH, noise_sd=pt['noise_sd'])
    if pt['dwell'] is not None:
        kw['mean_dwell'] = max(2, int(kw['mean_dwell'] * pt['dwell']))
        kw['min_dwell'] = max(1, min(kw['min_dwell'], kw['mean_dwell'] // 2))
    if pt['lead'] is not None:
        kw['lead'] = pt['lead']
    reads = []
    for k in range(3):
        nb = pt['n_bases'] + 53 * k
        seq, raw, starts = synth.synth_read(model, nb, pt['seed'] + k, **kw)
        seq, raw = synth.edit_read(seq, raw, starts, pt['edit'])
        b = len(seq) - model.kmer_width + 1
        rs = np.random.RandomState(pt['seed'] + k)
        si = rs.choice(b, 1000, replace=False).astype(np.int64) if b > 1000 else None
        reads.append((raw, seq, None, si))
    eng, out, oracles = run_batch(model, params, name, reads)
    bad = compare_batch(eng, oracles, out, repr(pt))
    assert not bad, '\n'.join(bad[:20])
    check_forms(eng, form, params)


def test_rna_without_outlier_thresh_is_an_unexpected_error():
    """get_scale_values_from_events negates outlier_thresh (tombo_stats.py:228): with None the
    reference dies with a TypeError -- not a TomboError; status TBA_INTERNAL here"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th, resquiggle as rq
    samp = th.seqSampleType('RNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    mr = synth.synth_map_res(model, 400, 42, **synth.RNA_SYNTH)
    res = rq.resquiggle_batch([mr], model, ts.load_resquiggle_parameters(samp), None, seq_samp_type=samp)
    assert isinstance(res[0], RuntimeError) and not isinstance(res[0], th.TomboError)
