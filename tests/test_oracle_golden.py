"""Pins oracle/ (the CPU restatement) against the golden vectors generated from the live
reference (tests/golden/gen_golden.py).  Bit-exact: integer arrays equal, float arrays equal
by value and sha256."""
import numpy as np
import pytest

from conftest import golden_names
from tombo_amd import errors, tombo_stats as ts
from tombo_amd._default_parameters import SIG_MATCH_THRESH
import oracle


def run_oracle(c, scale_values=None, debug=True):
    m = c.meta
    p = oracle.make_params(c.params)
    o = oracle.make_opts(
        c.model.kmer_width, c.model.central_pos, outlier_thresh=m['outlier_thresh'],
        const_scale=None if scale_values is not None else m['const_scale'],
        scale_values=scale_values,
        skip_seq_scaling=False if scale_values is not None else m['skip_seq_scaling'],
        sig_match_thresh=SIG_MATCH_THRESH[m['samp']], max_raw_cpts=c.max_raw_cpts)
    return oracle.resquiggle_read(
        c.raw, ts.encode_seq(c.seq), c.model.level_means, c.model.level_sds, p, o,
        stall_ints=c.stall_ints, samp_ind=c.samp_ind(), debug=debug)


def m_ties(c):
    """positions in exact score ties among the picks' range, counted on the score array the
    reference itself handed to np.argsort (gen_golden.py; 0 / absent: the picks are defined)"""
    return int(c.meta.get('score_ties') or 0)


@pytest.mark.parametrize('name', golden_names(oracle_only=True))
def test_oracle_matches_reference(golden_case, name):
    c = golden_case(name)
    g = c.g
    r = run_oracle(c)
    if c.error:
        assert errors.message(r['status']) == c.error
        return
    assert r['status'] == 0, errors.message(r['status'])
    d = r['dbg']
    if m_ties(c):
        # exact ties among the change-point scores that decide the picks (outlier clipping makes
        # flat windows): the reference takes them in np.argsort's order, which is numpy's business
        # (unstable sort, differs between CPU dispatches) -- the fixture pins the signal and how
        # far the picks can differ, nothing downstream
        c.check_float('seg_norm_signal', d['seg_norm_signal'])
        assert len(d['valid_cpts']) == len(g['valid_cpts'])
        assert np.isin(d['valid_cpts'], g['valid_cpts']).mean() > 0.99
        return
    np.testing.assert_array_equal(d['valid_cpts'], g['valid_cpts'])
    c.check_float('seg_norm_signal', d['seg_norm_signal'])
    sv = g['seg_scale_values']
    np.testing.assert_array_equal(d['seg_scale_values'][:2], sv[:2])
    if not np.isnan(sv[2]):
        np.testing.assert_array_equal(d['seg_scale_values'][2:], sv[2:])
    # RNA: compute_base_means is called once more, before the event means -- on the raw events, for
    # the event-based scaling (resquiggle.py:1089-1093; not with const_scale / given scale values)
    rna_event_scaling = c.meta['samp'] == 'RNA' and c.meta['const_scale'] is None
    c.check_float('base_means_call1' if rna_event_scaling else 'base_means_call0', d['event_means'])
    if 'start_call0' in g.files:
        # the capture wrapper only records calls that returned: a lone record with the "save"
        # bandwidth is the retry after the first call raised
        retry = int(g['start_call0_bw']) == c.params.start_save_bw
        assert d['n_start_calls'] == (2 if retry else 1)
        got = d['start_calls'][2:4] if retry else d['start_calls'][0:2]
        np.testing.assert_array_equal(got, g['start_call0'])
    if 'static_read_tb' in g.files:
        assert d['used_static']
        np.testing.assert_array_equal(d['read_tb'], g['static_read_tb'])
    else:
        assert not d['used_static']
        np.testing.assert_array_equal(d['band_event_starts'], g['band_event_starts'])
        np.testing.assert_array_equal(d['fwd_last_row'], g['fwd_last_row'])
        assert d['mask_seq_len'] == int(g['adapt_start_seq_pos'])
    np.testing.assert_array_equal(d['dp_segs'], g['dp_segs'])
    assert d['dp_read_start'] == int(g['dp_read_start_rel_to_raw'])
    np.testing.assert_array_equal(r['segs'], g['segs'])
    assert r['read_start_rel_to_raw'] == int(g['read_start_rel_to_raw'])
    if 'theil_sen' in g.files:
        np.testing.assert_array_equal(d['theil_sen'], g['theil_sen'])
    c.check_float('norm_signal', r['norm_signal'])
    np.testing.assert_array_equal(r['scale_values'][:2], g['scale_values'][:2])
    assert r['sig_match_score'] == float(g['sig_match_score'])
    assert r['norm_params_changed'] == bool(g['norm_params_changed'])


@pytest.mark.parametrize('name', [n for n in golden_names()
                                  if n in ('dna_b600_w300', 'dna_b2000_w300', 'rna_b600_w500',
                                           'p_dna_seg_7_4_2_6', 'p_dna_outlier3')])
def test_oracle_second_iteration(golden_case, name):
    """run_rsqgl_iters semantics (resquiggle.py:1492-1504): re-run with the fitted scale values"""
    from tombo_amd import tombo_helper as th
    c = golden_case(name)
    g = c.g
    sv = g['scale_values']
    r2 = run_oracle(c, scale_values=th.scaleValues(sv[0], sv[1], sv[2], sv[3],
                                                   c.meta['outlier_thresh']))
    assert errors.message(r2['status']) == str(g['it2_error']) or r2['status'] == 0
    np.testing.assert_array_equal(r2['dbg']['valid_cpts'], g['it2_valid_cpts'])
    np.testing.assert_array_equal(r2['segs'], g['it2_segs'])
    assert r2['read_start_rel_to_raw'] == int(g['it2_read_start_rel_to_raw'])
    c.check_float('it2_norm_signal', r2['norm_signal'])
    np.testing.assert_array_equal(r2['scale_values'], g['it2_scale_values'])
    assert r2['sig_match_score'] == float(g['it2_sig_match_score'])
    assert r2['norm_params_changed'] == bool(g['it2_norm_params_changed'])


def test_numpy_restatements():
    """orc_np_sum / orc_median / orc_linspace against numpy itself (numpy is on both boxes)"""
    import ctypes as C
    L = oracle.lib()
    rng = np.random.default_rng(1)
    for n in (1, 2, 7, 8, 9, 15, 16, 17, 127, 128, 129, 255, 1000, 1001, 4097, 10000, 92067):
        a = rng.normal(size=n) * 10 ** rng.uniform(-3, 3, size=n)
        s = L.orc_np_sum(a.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(n))
        assert s == np.add.reduce(a), n
        assert s / n == np.mean(a)
        med = L.orc_median(a.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(n))
        assert med == np.median(a), n
        b = np.round(a)  # heavy ties
        med = L.orc_median(b.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(n))
        assert med == np.median(b), n
    L.orc_linspace.argtypes = [C.c_double, C.c_double, C.c_int64, C.POINTER(C.c_double)]
    for (a, b, n) in ((0, 37, 74), (-150.0, -150 + 251 * 1.8007968127490041, 251),
                      (3.0, 3.0, 10), (101, 893, 50), (0, 5, 1), (2, 9, 2)):
        out = np.zeros(n)
        L.orc_linspace(a, b, n, out.ctypes.data_as(C.POINTER(C.c_double)))
        np.testing.assert_array_equal(out, np.linspace(a, b, n))


def degenerate_signal(case):
    """the signal of a tests/golden/degenerate_cases.json record (as gen_golden_degenerate.make)"""
    rng = np.random.default_rng(case['seed'])
    x = rng.normal(90.0, 12.0, case['n'])
    if case['kind'] == 'flat':
        x[:] = case['value']
    elif case['kind'] == 'saturated':
        x[rng.permutation(case['n'])[:case['n'] // 2 + case['extra']]] = case['value']
    elif case['kind'] == 'half':
        x[rng.permutation(case['n'])[:case['n'] // 2]] = case['value']
    if case['dtype'] == 'int16':
        x = np.round(x).astype(np.int16)
    return x


def degenerate_cases():
    import os
    import json
    from conftest import GOLDEN_DIR
    with open(os.path.join(GOLDEN_DIR, 'degenerate_cases.json')) as fp:
        return json.load(fp)


def test_oracle_follows_the_reference_on_a_zero_mad():
    """a flat / saturated signal: the live reference raised FloatingPointError in the division by the
    scale (np.seterr(all='raise')) -> the read's unexpected error, status 100; otherwise its scale values"""
    from tombo_amd import synth, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    seq = synth.synth_read(model, 400, 1, **synth.DNA_SYNTH)[0]
    p = oracle.make_params(params)
    o = oracle.make_opts(model.kmer_width, model.central_pos, outlier_thresh=5.0,
                         sig_match_thresh=SIG_MATCH_THRESH['DNA'])
    n_raised = 0
    for case in degenerate_cases():
        x = degenerate_signal(case).astype(np.float64)
        r = oracle.resquiggle_read(x, ts.encode_seq(seq), model.level_means, model.level_sds, p, o, debug=True)
        if case['raised']:
            assert case['raised'] == 'FloatingPointError' and not case['is_tombo_error']
            assert r['status'] == 100, case
            n_raised += 1
        else:
            assert r['status'] != 100, case     # (noise against a sequence: some TomboError, not a crash)
            assert r['dbg']['seg_scale_values'][0] == case['shift'] and r['dbg']['seg_scale_values'][1] == case['scale'], case
    assert n_raised >= 4
