"""GPU: the reference's public per-stage API (tombo/resquiggle.py:63-67) re-exposed by
tombo_amd.resquiggle, against the stage-wise golden vectors recorded from the reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _map_res(c):
    from tombo_amd import tombo_helper as th
    return th.resquiggleResults(
        align_info=th.alignInfo('r', 'BaseCalled_template', 0, 0, 0, 0, c.meta['n_bases'], 0),
        genome_loc=th.genomeLocation(0, '+', 'synth'), genome_seq=c.seq, mean_q_score=10.0,
        raw_signal=c.raw, stall_ints=c.stall_ints)


@pytest.mark.parametrize('name', ['dna_b600_w300', 'dna_b2000_w300', 'rna_b600_w500'])
def test_stepwise_stages_match_reference(golden_case, name):
    from tombo_amd import resquiggle as rq, tombo_stats as ts, tombo_helper as th
    c = golden_case(name)
    g = c.g
    mr = _map_res(c)
    num_events = ts.compute_num_events(c.raw.shape[0], c.meta['n_bases'],
                                       c.params.mean_obs_per_event)
    cpts, norm, sv = rq.segment_signal(mr, num_events, c.params, 5.0)
    np.testing.assert_array_equal(cpts, g['valid_cpts'])
    c.check_float('seg_norm_signal', norm)
    assert (sv.shift, sv.scale) == tuple(g['seg_scale_values'][:2])
    assert (sv.lower_lim, sv.upper_lim) == tuple(g['seg_scale_values'][2:])
    key = 'base_means_call1' if c.meta['samp'] == 'RNA' else 'base_means_call0'
    if key not in g:
        return
    event_means = g[key]
    dp = rq.find_adaptive_base_assignment(cpts, event_means, c.params, c.model, c.seq,
                                          seq_samp_type=c.samp)
    np.testing.assert_array_equal(dp.segs, g['dp_segs'])
    assert dp.read_start_rel_to_raw == int(g['dp_read_start_rel_to_raw'])
    assert len(dp.genome_seq) == c.meta['n_bases'] and dp.ref_means.shape[0] == c.meta['n_bases']
    mu, sd = c.model.get_exp_levels_from_seq(c.seq)
    np.testing.assert_array_equal(dp.ref_means, mu)
    seg_norm = norm[dp.read_start_rel_to_raw:dp.read_start_rel_to_raw + dp.segs[-1]]
    segs = rq.resolve_skipped_bases_with_raw(dp, seg_norm, c.params, 200)
    np.testing.assert_array_equal(segs, g['segs'])
    if 'start_call0' in g and int(g['start_call0_bw']) == c.params.start_bw:
        loc, epb = rq.find_seq_start_in_events(event_means, mu, sd, c.params,
                                               c.params.start_n_bases, c.params.start_bw, c.samp)
        assert (float(loc), epb) == tuple(g['start_call0'])


def test_static_assignment_and_errors(golden_case):
    from tombo_amd import resquiggle as rq, tombo_helper as th
    c = golden_case('dna_b150_static')
    g = c.g
    mu, sd = c.model.get_exp_levels_from_seq(c.seq)
    tb = rq.find_static_base_assignment(g['base_means_call0'], mu, sd, c.params)
    np.testing.assert_array_equal(tb, g['static_read_tb'])
    with pytest.raises(th.TomboError, match='Read too short for start/end discovery'):
        rq.find_seq_start_in_events(g['base_means_call0'], mu, sd, c.params, 250, 750, c.samp)
    # a noise read fails the start score check exactly like the reference
    n = golden_case('dna_noise_body')
    mu, sd = n.model.get_exp_levels_from_seq(n.seq)
    import oracle
    from tombo_amd import tombo_stats as ts
    o = oracle.resquiggle_read(n.raw, ts.encode_seq(n.seq), n.model.level_means,
                               n.model.level_sds, oracle.make_params(n.params),
                               oracle.make_opts(6, 2, outlier_thresh=5.0, sig_match_thresh=1.1),
                               debug=True)
    with pytest.raises(th.TomboError, match='Poor raw to expected signal matching'):
        rq.find_seq_start_in_events(o['dbg']['event_means'], mu, sd, n.params, 250, 750, n.samp)
    # too many change points requested for the signal
    mr = _map_res(c)
    with pytest.raises(th.TomboError, match='Fewer changepoints found than requested'):
        rq.segment_signal(mr._replace(raw_signal=c.raw[:900]), 400, c.params, 5.0)


def test_resolve_skipped_bases_window_arguments():
    """del_fix_window / max_del_fix_window / extra_sig_factor of resolve_skipped_bases_with_raw
    (resquiggle.py:405-407) are run-time parameters of the engine: results and error messages
    recorded from the live reference (tests/golden/gen_golden_skipwin.py)"""
    import os
    import json
    from tombo_amd import resquiggle as rq, tombo_stats as ts, tombo_helper as th
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'kernels_skipwin.npz'))
    settings = json.loads(str(g['settings']))
    n_err = 0
    for name in g['names']:
        p = 'sw_%s_' % name
        params = ts.load_resquiggle_parameters(th.seqSampleType(str(g[p + 'samp']), False))
        segs = g[p + 'segs']
        dp = th.dpResults(read_start_rel_to_raw=0, segs=segs, ref_means=g[p + 'means'], ref_sds=g[p + 'sds'],
                          genome_seq='A' * (segs.shape[0] - 1))
        for k, (dfw, mdfw, esf, mrc) in enumerate(settings):
            err = str(g[p + 'err%d' % k])
            if err:
                n_err += 1
                with pytest.raises(th.TomboError) as ei:
                    rq.resolve_skipped_bases_with_raw(dp, g[p + 'norm'], params, mrc, dfw, mdfw, esf)
                assert str(ei.value) == err, (name, settings[k])
            else:
                got = rq.resolve_skipped_bases_with_raw(dp, g[p + 'norm'], params, mrc, dfw, mdfw, esf)
                np.testing.assert_array_equal(got, g[p + 'res%d' % k], err_msg='%s %r' % (name, settings[k]))
    assert n_err >= 10
