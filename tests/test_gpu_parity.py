"""GPU parity tests proper: the HIP batch engine (through the C ABI) against the CPU oracle on
the same seeded inputs, stage by stage, and against the committed golden fixtures.

Bar: bit-exact for every integer output (change points, band starts, traceback, segment
boundaries, statuses); float64 outputs compared bit-exactly where the arithmetic is restated
operation by operation, with the contractual 1e-5 tolerance on the normalised signal asserted
separately.
"""
import os
import numpy as np
import pytest

import oracle

from conftest import golden_names, check_forms

pytestmark = pytest.mark.gpu


def _engine():
    from tombo_amd import resquiggle as rq
    return rq.get_engine(0)


def _oracle_read(model, params, raw, seq, outlier_thresh, samp_name, stall_ints=None,
                 samp_ind=None, scale_values=None, const_scale=None, skip_seq_scaling=False,
                 max_raw_cpts=200):
    import oracle
    from tombo_amd import tombo_stats as ts
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    p = oracle.make_params(params)
    o = oracle.make_opts(model.kmer_width, model.central_pos, outlier_thresh=outlier_thresh,
                         const_scale=const_scale, scale_values=scale_values,
                         skip_seq_scaling=skip_seq_scaling,
                         sig_match_thresh=SIG_MATCH_THRESH[samp_name], max_raw_cpts=max_raw_cpts)
    return oracle.resquiggle_read(raw, ts.encode_seq(seq), model.level_means, model.level_sds,
                                  p, o, stall_ints=stall_ints, samp_ind=samp_ind, debug=True)


def compare_batch(eng, oracles, out, label=''):
    """Stage-wise comparison of one engine batch with per-read oracle results.
    Returns a list of mismatch descriptions (empty == parity)."""
    from tombo_amd import _native as N
    bad = []
    n = len(oracles)
    cpts = eng.get(N.GET_VALID_CPTS)
    ncp = eng.get(N.GET_N_CPTS)
    evm = eng.get(N.GET_EVENT_MEANS)
    segnorm = eng.get(N.GET_SEG_NORM)
    start = eng.get(N.GET_START)
    bst = eng.get(N.GET_BAND_STARTS)
    rtb = eng.get(N.GET_READ_TB)
    dps = eng.get(N.GET_DP_SEGS)
    dprs = eng.get(N.GET_DP_READ_START)
    tsn = eng.get(N.GET_THEIL_SEN)
    path = eng.get(N.GET_PATH)
    lastrow = eng.get(N.GET_LAST_ROW)

    def chk(i, name, a, b, exact=True):
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape:
            bad.append('%s read %d %s: shape %s vs %s' % (label, i, name, a.shape, b.shape))
            return False
        if not np.array_equal(a, b):
            d = np.flatnonzero(a != b)
            bad.append('%s read %d %s: %d/%d differ, first at %d: gpu=%r oracle=%r' % (
                label, i, name, d.size, a.size, d[0], a.flat[d[0]], b.flat[d[0]]))
            return False
        return True

    for i, o in enumerate(oracles):
        st = int(out['status'][i])
        if st != o['status']:
            bad.append('%s read %d status: gpu=%d oracle=%d' % (label, i, st, o['status']))
        d = o['dbg']
        e0 = eng.ev_off[i]
        if o['status'] in (1,):
            continue
        ok = chk(i, 'n_cpts', ncp[i], len(d['valid_cpts']))
        if len(d['valid_cpts']):
            ok = ok and chk(i, 'valid_cpts', cpts[e0:e0 + int(ncp[i])], d['valid_cpts'])
            chk(i, 'seg_norm', segnorm[eng.raw_off[i]:eng.raw_off[i + 1]], d['seg_norm_signal'])
            chk(i, 'event_means', evm[e0:e0 + max(int(ncp[i]) - 1, 0)], d['event_means'])
        if d['n_start_calls'] >= 1 and path[i, 3] >= 1:
            k = d['n_start_calls']
            chk(i, 'start', start[i, 2 * (k - 1):2 * k], d['start_calls'][2 * (k - 1):2 * k])
        if o['status'] == 0 or d['dp_segs'].any():
            chk(i, 'used_static', int(path[i, 0] == 2), int(d['used_static']))
            r0, r1 = eng.ref_off[i], eng.ref_off[i + 1]
            s0, s1 = eng.seg_off[i], eng.seg_off[i + 1]
            if not d['used_static']:
                chk(i, 'mask_seq_len', path[i, 1], d['mask_seq_len'])
            else:
                chk(i, 'static_W', path[i, 2], len(d['fwd_last_row']))
            chk(i, 'band_starts', bst[r0:r1], d['band_event_starts'])
            if len(d['fwd_last_row']) <= lastrow.shape[1]:  # not kept for the wide static DP
                chk(i, 'last_row', lastrow[i, :len(d['fwd_last_row'])], d['fwd_last_row'])
            chk(i, 'read_tb', rtb[s0:s1], d['read_tb'])
            chk(i, 'dp_segs', dps[s0:s1], d['dp_segs'])
            chk(i, 'dp_read_start', dprs[i], d['dp_read_start'])
        if o['status'] == 0 and st == 0:
            s0, s1 = eng.seg_off[i], eng.seg_off[i + 1]
            chk(i, 'segs', out['segs'][s0:s1], o['segs'])
            chk(i, 'read_start', out['read_start'][i], o['read_start_rel_to_raw'])
            chk(i, 'theil_sen', tsn[i], d['theil_sen'])
            nl = int(out['norm_len'][i])
            gn = out['norm'][eng.raw_off[i]:eng.raw_off[i] + nl]
            if chk(i, 'norm_len', nl, len(o['norm_signal'])):
                chk(i, 'norm_signal', gn, o['norm_signal'])
                if np.max(np.abs(gn - o['norm_signal'])) > 1e-5:
                    bad.append('%s read %d norm_signal beyond 1e-5' % (label, i))
            chk(i, 'scale_values', out['sv'][i][:2], o['scale_values'][:2])
            chk(i, 'score', out['score'][i], o['sig_match_score'])
            chk(i, 'changed', bool(out['changed'][i]), o['norm_params_changed'])
    return bad


def run_batch(model, params, samp_name, reads, outlier_thresh=5.0, const_scale=None,
              skip_seq_scaling=False, scale_values=None, seed0=None, max_raw_cpts=200,
              raw_dtype=np.float64):
    """reads: list of (raw, seq, stall_ints, samp_ind); raw_dtype: the sample type of the upload
    (the oracle always sees float64).  Returns (engine, out, oracles)."""
    from tombo_amd import _native as N, tombo_stats as ts
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    eng = _engine()
    eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    n = len(reads)
    p = N.make_params(params)
    o = N.make_opts(outlier_thresh=outlier_thresh, const_scale=const_scale,
                    skip_seq_scaling=skip_seq_scaling,
                    sig_match_thresh=SIG_MATCH_THRESH[samp_name], max_raw_cpts=max_raw_cpts)
    si = np.zeros((n, 1000), np.int64)
    for i, r in enumerate(reads):
        if r[3] is not None:
            si[i] = r[3]
    sv_in = sv_flags = None
    if scale_values is not None:
        sv_in = np.array([[sv.shift, sv.scale, sv.lower_lim, sv.upper_lim]
                          for sv in scale_values], dtype=np.float64)
        sv_flags = np.full(n, 3, np.int32)
    stalls = [r[2] for r in reads]
    eng.upload(p, o, [np.asarray(r[0], raw_dtype) for r in reads],
               [ts.encode_seq(r[1]) for r in reads], sv_in=sv_in, sv_flags=sv_flags,
               samp_ind=si, stall_ints=stalls if any(s is not None for s in stalls) else None)
    eng.run()
    out = eng.download()
    oracles = []
    for i, r in enumerate(reads):
        oracles.append(_oracle_read(
            model, params, r[0], r[1], outlier_thresh, samp_name, stall_ints=r[2],
            samp_ind=r[3], const_scale=const_scale, skip_seq_scaling=skip_seq_scaling,
            scale_values=None if scale_values is None else scale_values[i],
            max_raw_cpts=max_raw_cpts))
    return eng, out, oracles


@pytest.mark.parametrize('name', golden_names())
def test_golden_case_on_gpu(golden_case, name, dispatch_form):
    """every committed golden fixture through the HIP path (batch of one, in the latency AND in the
    throughput form of event detection / traceback) vs the oracle, and the final outputs vs the
    reference's recorded values"""
    from tombo_amd import errors
    c = golden_case(name)
    m = c.meta
    eng, out, oracles = run_batch(
        c.model, c.params, m['samp'], [(c.raw, c.seq, c.stall_ints, c.samp_ind())],
        outlier_thresh=m['outlier_thresh'], const_scale=m['const_scale'],
        skip_seq_scaling=m['skip_seq_scaling'], max_raw_cpts=c.max_raw_cpts)
    bad = compare_batch(eng, oracles, out, name)
    assert not bad, '\n'.join(bad)
    check_forms(eng, dispatch_form, c.params)
    g = c.g
    if c.error:
        assert errors.message(out['status'][0]) == c.error
    else:
        assert out['status'][0] == 0
        np.testing.assert_array_equal(out['segs'], g['segs'])
        assert out['read_start'][0] == int(g['read_start_rel_to_raw'])
        c.check_float('norm_signal', out['norm'][:int(out['norm_len'][0])])
        assert out['score'][0] == float(g['sig_match_score'])


def test_mixed_dna_batch_on_gpu(dispatch_form):
    """one ragged batch mixing lengths / paths (adaptive, static fallback, retry, failures)"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    reads = []
    specs = [(600, 1, {}), (150, 2, {}), (1200, 3, {}), (300, 4, {}), (1500, 5, dict(lead=5000)),
             (20, 6, dict(lead=60000)), (400, 7, dict(mean_dwell=400)), (2000, 8, {}),
             (260, 9, {}), (1001, 10, {}), (999, 11, {}), (700, 12, dict(mean_dwell=4)),
             (800, 13, dict(noise_sd=0.6)), (500, 14, dict(n_trail=3000))]
    for nb, seed, kw in specs:
        k = dict(synth.DNA_SYNTH)
        k.update(kw)
        seq, raw, _ = synth.synth_read(model, nb, 500 + seed, **k)
        si = None
        if nb > 1000:
            st = np.random.get_state()
            np.random.seed(seed)
            si = np.random.choice(nb, 1000, replace=False)
            np.random.set_state(st)
        reads.append((raw, seq, None, si))
    eng, out, oracles = run_batch(model, params, 'DNA', reads)
    bad = compare_batch(eng, oracles, out, 'mixed')
    assert not bad, '\n'.join(bad[:40])
    check_forms(eng, dispatch_form, params)
    assert sum(o['status'] == 0 for o in oracles) >= 8


def test_many_seeds_w100_on_gpu(dispatch_form):
    """BASELINE configs[0] shape (2 kb reads, bandwidth 100, band_bound_thresh 10), 48 seeds"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=100, band_bound_thresh=10)
    reads = []
    for seed in range(48):
        seq, raw, _ = synth.synth_read(model, 2000, 9000 + seed, **synth.DNA_SYNTH)
        st = np.random.get_state()
        np.random.seed(seed)
        si = np.random.choice(2000, 1000, replace=False)
        np.random.set_state(st)
        reads.append((raw, seq, None, si))
    eng, out, oracles = run_batch(model, params, 'DNA', reads)
    bad = compare_batch(eng, oracles, out, 'w100')
    assert not bad, '\n'.join(bad[:40])
    check_forms(eng, dispatch_form, params, require_all_fused=True)
    assert all(o['status'] == 0 for o in oracles)


def test_second_iteration_on_gpu(golden_case, dispatch_form):
    """run_rsqgl_iters (resquiggle.py:1492-1504): re-run with fitted scale values"""
    from tombo_amd import tombo_helper as th
    c = golden_case('dna_b2000_w300')
    g = c.g
    sv = g['scale_values']
    svs = [th.scaleValues(sv[0], sv[1], sv[2], sv[3], 5.0)]
    eng, out, oracles = run_batch(c.model, c.params, 'DNA',
                                  [(c.raw, c.seq, None, c.samp_ind())], scale_values=svs)
    bad = compare_batch(eng, oracles, out, 'iter2')
    assert not bad, '\n'.join(bad)
    check_forms(eng, dispatch_form, c.params)
    np.testing.assert_array_equal(out['segs'], g['it2_segs'])
    assert out['score'][0] == float(g['it2_sig_match_score'])


def test_resquiggle_read_dropin_on_gpu(golden_case):
    """the Python drop-in: same call as the reference, same result tuple"""
    from tombo_amd import resquiggle as rq, tombo_helper as th
    c = golden_case('dna_b2000_w300')
    g = c.g
    mr = th.resquiggleResults(
        align_info=th.alignInfo('r', 'BaseCalled_template', 0, 0, 0, 0, 2000, 0),
        genome_loc=th.genomeLocation(0, '+', 'synth'), genome_seq=c.seq, mean_q_score=10.0,
        raw_signal=c.raw)
    np.random.seed(c.meta['np_seed'])
    res = rq.resquiggle_read(mr, c.model, c.params, 5.0, seq_samp_type=c.samp)
    np.testing.assert_array_equal(res.segs, g['segs'])
    assert res.read_start_rel_to_raw == int(g['read_start_rel_to_raw'])
    assert res.sig_match_score == float(g['sig_match_score'])
    assert res.norm_params_changed == bool(g['norm_params_changed'])
    assert len(res.genome_seq) == 2000
    c.check_float('norm_signal', res.raw_signal)
    bad = golden_case('dna_dwell400_bandfail')
    mr2 = mr._replace(genome_seq=bad.seq, raw_signal=bad.raw)
    with pytest.raises(th.TomboError, match='extends beyond bandwidth'):
        rq.resquiggle_read(mr2, bad.model, bad.params, 5.0, seq_samp_type=bad.samp)


def _si(nb, seed):
    if nb <= 1000:
        return None
    st = np.random.get_state()
    np.random.seed(seed)
    si = np.random.choice(nb, 1000, replace=False)
    np.random.set_state(st)
    return si


def test_full_size_reads_vs_oracle_on_gpu(dispatch_form):
    """BASELINE configs[1] shape: 10 kb DNA reads, bandwidth 500 -- 40 reads against the oracle"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
    reads = []
    for seed in range(40):
        seq, raw, _ = synth.synth_read(model, 10000, 77000 + seed, **synth.DNA_SYNTH)
        reads.append((raw, seq, None, _si(10000, seed)))
    eng, out, oracles = run_batch(model, params, 'DNA', reads)
    bad = compare_batch(eng, oracles, out, 'cfg2')
    assert not bad, '\n'.join(bad[:40])
    check_forms(eng, dispatch_form, params, require_all_fused=True)
    assert all(o['status'] == 0 for o in oracles)
    # (performance guard, not parity: every one of these adaptive reads must have been walked by the
    # chunk-parallel traceback -- a read it leaves costs the whole batch the serial walk's 5 ms)
    from tombo_amd import _native
    done, path = eng.get(_native.GET_TB_PARALLEL), eng.get(_native.GET_PATH)[:, 0]
    assert np.all(done[path == 1] == 1), 'reads left to the lane-per-read traceback: %r' % (
        np.flatnonzero((path == 1) & (done != 1)).tolist(),)
    # ... by the kernel of the form under test (check_forms above has asserted the same of event
    # detection: k_detect + k_pick finished every read of the throughput form, none was flagged)
    tbf = eng.get(_native.GET_TB_FORM)
    assert np.all(tbf[path == 1] == (_native.TB_FORM_PAR16 if dispatch_form == 'throughput' else _native.TB_FORM_PAR64))
    assert np.all(eng.get(_native.GET_ED_FUSED) == (1 if dispatch_form == 'throughput' else 0))


def test_long_reads_vs_oracle_on_gpu(dispatch_form):
    """Reads well past one 8 192-element summation chunk of np.mean, 16-row traceback blocks by the
    thousand and several event-detection tiles: 18 kb, 25 kb and 37 kb next to a short one"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
    reads = []
    for seed, nb in enumerate((18000, 700, 25000, 37000)):
        seq, raw, _ = synth.synth_read(model, nb, 88000 + seed, **synth.DNA_SYNTH)
        reads.append((raw, seq, None, _si(nb, seed)))
    eng, out, oracles = run_batch(model, params, 'DNA', reads)
    bad = compare_batch(eng, oracles, out, 'long')
    assert not bad, '\n'.join(bad[:40])
    check_forms(eng, dispatch_form, params)
    assert all(o['status'] == 0 for o in oracles)


def test_default_bandwidth_ragged_batch_on_gpu(dispatch_form):
    """BASELINE configs[2] shape: the adaptive path at Tombo's default bandwidth 300, ragged
    lengths"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    rng = np.random.default_rng(5)
    reads = []
    for seed in range(32):
        nb = int(rng.integers(400, 6000))
        kw = dict(synth.DNA_SYNTH)
        kw['mean_dwell'] = int(rng.integers(7, 12))
        seq, raw, _ = synth.synth_read(model, nb, 31000 + seed, **kw)
        reads.append((raw, seq, None, _si(nb, seed)))
    eng, out, oracles = run_batch(model, params, 'DNA', reads)
    bad = compare_batch(eng, oracles, out, 'w300')
    assert not bad, '\n'.join(bad[:40])
    check_forms(eng, dispatch_form, params)
    assert sum(o['status'] == 0 for o in oracles) >= 28


def test_rna_batch_on_gpu(dispatch_form):
    """BASELINE configs[3] shape: direct RNA model, t-test segmentation, stall masking, event
    based scaling, raw_min_obs_per_base = 2 in the skipped-base DP"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('RNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    reads = []
    for seed, nb in enumerate((3000, 1500, 800, 2200, 3000, 400, 1100, 2600)):
        seq, raw, _ = synth.synth_read(model, nb, 88000 + seed, **synth.RNA_SYNTH)
        if seed == 4:  # a stalled stretch in the middle of the read
            raw = np.concatenate([raw[:40000], np.full(1500, raw[40000]) +
                                  np.random.default_rng(1).normal(0, 3.0, 1500), raw[40000:]])
        reads.append((raw, seq, oracle.identify_stalls(raw), _si(nb, seed)))
    assert any(len(r[2]) for r in reads), 'no stall interval exercised'
    eng, out, oracles = run_batch(model, params, 'RNA', reads)
    bad = compare_batch(eng, oracles, out, 'rna')
    assert not bad, '\n'.join(bad[:40])
    check_forms(eng, dispatch_form, params)
    assert sum(o['status'] == 0 for o in oracles) >= 6


@pytest.mark.parametrize('samp_name', ['DNA', 'RNA'])
def test_long_dwell_stretches_vs_oracle_on_gpu(samp_name, dispatch_form):
    """segment sums (wave_segment_sums, k_select.h): a group of segments that spans more samples than
    a wavefront's LDS slice is staged piece by piece, every lane carrying its running sum across the
    pieces -- flat stretches of 600-4000 samples (nothing masks them: no stall intervals are passed)
    put single events and single bases past every slice size (448 / 768 / 1 280 samples)"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType(samp_name, False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    kw = synth.RNA_SYNTH if samp_name == 'RNA' else synth.DNA_SYNTH
    rng = np.random.default_rng(77)
    reads = []
    for seed, nb in enumerate((1200, 2500, 700, 1800, 3000, 1500)):
        seq, raw, starts = synth.synth_read(model, nb, 99000 + seed, **kw)
        parts, at = [], 0
        for k in sorted(rng.choice(np.arange(20, nb - 20), 4 if samp_name == 'RNA' else 2, replace=False)):
            cut = int(starts[k])
            n_flat = int(rng.integers(600, 4000))
            parts += [raw[at:cut], raw[cut] + rng.normal(0.0, 0.02 * kw['scale'], n_flat)]
            at = cut
        raw = np.concatenate(parts + [raw[at:]])
        reads.append((raw, seq, None, _si(nb, seed)))
    eng, out, oracles = run_batch(model, params, samp_name, reads)
    bad = compare_batch(eng, oracles, out, 'dwell')
    assert not bad, '\n'.join(bad[:40])
    check_forms(eng, dispatch_form, params)
    ok = [i for i, o in enumerate(oracles) if o['status'] == 0]
    assert len(ok) >= 4, [o['status'] for o in oracles]
    # the case is what it claims: resolved bases longer than the widest slice
    longest = max(int(np.diff(oracles[i]['segs']).max()) for i in ok)
    assert longest > 1280, longest


def test_long_read_kernels_vs_oracle_on_gpu(dispatch_form):
    """k_long.h: reads past TBA_LONG_RAW samples / TBA_LONG_BASES bases get a workgroup-per-read
    cumulative sum and a wavefront-per-read traceback -- 60 kb and 29 kb reads (long by both / by
    samples only), a 26 kb read at bandwidth 300 whose band runs off the events (failure path),
    next to ordinary ones in the same batch; tile edges: lengths around multiples of the 1856-sample
    scan tile"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    for bw, specs in ((500, ((60000, {}), (900, {}), (29200, {}), (31000, dict(mean_dwell=12)))),
                      (300, ((26000, {}), (26500, dict(mean_dwell=30)), (1200, {})))):
        params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=bw)
        reads = []
        for seed, (nb, kw) in enumerate(specs):
            k = dict(synth.DNA_SYNTH)
            k.update(kw)
            seq, raw, _ = synth.synth_read(model, nb, 66000 + seed + bw, **k)
            if seed == 0:   # an exact multiple of the scan tile, then one more sample
                raw = raw[:(raw.shape[0] // 1856) * 1856 + (1 if bw == 300 else 0)]
            reads.append((raw, seq, None, _si(nb, seed)))
        eng, out, oracles = run_batch(model, params, 'DNA', reads)
        bad = compare_batch(eng, oracles, out, 'long%d' % bw)
        assert not bad, '\n'.join(bad[:40])
        check_forms(eng, dispatch_form, params)
        assert sum(o['status'] == 0 for o in oracles) >= len(specs) - 1


def test_long_rna_reads_vs_oracle_on_gpu(dispatch_form):
    """RNA reads of 6 and 9 kb (260 k / 390 k samples: dozens of event-detection tiles at radius
    5, every LDS class of the skipped-base windows, a stall in the longer one)"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('RNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    reads = []
    for seed, nb in enumerate((6000, 9000, 350)):
        seq, raw, _ = synth.synth_read(model, nb, 99000 + seed, **synth.RNA_SYNTH)
        if seed == 1:
            raw = np.concatenate([raw[:150000], np.full(2000, raw[150000]) +
                                  np.random.default_rng(2).normal(0, 3.0, 2000), raw[150000:]])
        reads.append((raw, seq, oracle.identify_stalls(raw), _si(nb, seed)))
    eng, out, oracles = run_batch(model, params, 'RNA', reads)
    bad = compare_batch(eng, oracles, out, 'rna_long')
    assert not bad, '\n'.join(bad[:40])
    check_forms(eng, dispatch_form, params)
    assert sum(o['status'] == 0 for o in oracles) >= 2


def test_batch_properties_at_scale_on_gpu(dispatch_form):
    """size-independent properties on a larger batch (no oracle): monotone boundaries, trimmed
    signal covered exactly, determinism across runs, independence from batch composition"""
    import hashlib
    from tombo_amd import _native as N, synth, tombo_stats as ts, tombo_helper as th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
    n = 384
    raws, seqs = [], []
    for i in range(n):
        seq, raw, _ = synth.synth_read(model, 4000 + 8 * (i % 50), 910000 + i, **synth.DNA_SYNTH)
        raws.append(raw)
        seqs.append(ts.encode_seq(seq))
    rng = np.random.RandomState(3)
    si = np.stack([rng.choice(4000, 1000, replace=False) for _ in range(n)])
    eng = _engine()
    eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    p = N.make_params(params)
    o = N.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA'])

    def run(idx):
        eng.upload(p, o, [raws[i] for i in idx], [seqs[i] for i in idx], samp_ind=si[idx])
        eng.run()
        out = eng.download()
        per = []
        for j in range(len(idx)):
            segs = out['segs'][eng.seg_off[j]:eng.seg_off[j + 1]]
            nl = int(out['norm_len'][j])
            sig = out['norm'][eng.raw_off[j]:eng.raw_off[j] + nl]
            h = hashlib.sha256(segs.tobytes() + sig.tobytes() + out['sv'][j].tobytes()).hexdigest()
            per.append((int(out['status'][j]), h, segs, nl, int(out['read_start'][j])))
        return per

    full = run(np.arange(n))
    assert sum(st == 0 for st, *_ in full) >= n - 2
    for j, (st, h, segs, nl, rs) in enumerate(full):
        if st != 0:
            continue
        assert segs[0] == 0 and segs[-1] == nl and np.all(np.diff(segs) >= 1)
        assert rs >= 0 and rs + nl <= raws[j].shape[0]
    again = run(np.arange(n))
    assert [h for _, h, *_ in again] == [h for _, h, *_ in full], 'not deterministic'
    sub = np.array([5, 300, 17, 128, 64, 383, 1])
    part = run(sub)
    assert [h for _, h, *_ in part] == [full[i][1] for i in sub], 'depends on batch composition'


def test_default_dispatch_above_and_below_its_thresholds_agree_on_gpu():
    """2 048 reads of 4 kb at W = 500 as ONE batch (default dispatch: the throughput kernels) and as two
    batches of 1 024 (the latency kernels): same boundaries, signal, scale values and scores read for
    read -- the two forms against each other at the benchmark's band width, beyond what the oracle
    finishes in seconds"""
    import hashlib
    from tombo_amd import _native as N, synth, tombo_stats as ts, tombo_helper as th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
    n = 2048
    sp = N.make_synth_params(**synth.DNA_SYNTH)
    gen = N.Synth(model, 0)
    eng = _engine()
    assert eng.get_dispatch() == (1024, 1024)
    eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    p = N.make_params(params)

    def run(first, count):
        raw, raw_off, seq, seq_off = gen.generate(sp, 424243, np.full(count, 4000, np.int64), raw_dtype=np.int16,
                                                  first_read=first)
        o = N.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA'], subsample_seed=5,
                        subsample_first_read=first)
        eng.upload_packed(p, o, raw, raw_off, seq, seq_off, wait=True)
        eng.run()
        out = eng.download()
        ed, tb = eng.get(N.GET_ED_FORM), eng.get(N.GET_TB_FORM)
        per = []
        for j in range(count):
            segs = out['segs'][eng.seg_off[j]:eng.seg_off[j + 1]]
            nl = int(out['norm_len'][j])
            sig = out['norm'][eng.raw_off[j]:eng.raw_off[j] + nl]
            per.append((int(out['status'][j]), hashlib.sha256(
                segs.tobytes() + sig.tobytes() + out['sv'][j].tobytes() + out['score'][j:j + 1].tobytes()).hexdigest()))
        return per, ed, tb
    whole, ed_w, tb_w = run(0, n)
    assert sum(st == 0 for st, _ in whole) >= n - 8
    ok = np.array([st == 0 for st, _ in whole])
    assert np.all(ed_w[ok] == N.ED_FORM_DETECT_PICK) and set(tb_w[ok].tolist()) <= {N.TB_FORM_PAR16, N.TB_FORM_LANE}
    halves = []
    for first in (0, 1024):
        part, ed_p, tb_p = run(first, 1024)
        okp = np.array([st == 0 for st, _ in part])
        assert np.all(ed_p[okp] == N.ED_FORM_WG_SCAN_PEAKS) and N.TB_FORM_PAR16 not in set(tb_p.tolist())
        halves += part
    assert halves == whole


def test_degenerate_inputs_on_gpu(dispatch_form):
    """too-short sequence, tiny signal, constant signal: a status, never a crash"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th, resquiggle as rq
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    good = synth.synth_map_res(model, 500, 4, **synth.DNA_SYNTH)
    mrs = [good._replace(genome_seq='ACG'),                       # shorter than the k-mer
           good._replace(raw_signal=good.raw_signal[:12]),        # almost no signal
           good._replace(raw_signal=np.full(6000, 90.0)),         # constant: MAD == 0
           good._replace(genome_seq=good.genome_seq[:200] + 'N' + good.genome_seq[201:]),
           good]
    res = rq.resquiggle_batch(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp)
    assert all(isinstance(r, Exception) for r in res[:4])
    assert isinstance(res[3], th.TomboError) and 'Invalid sequence' in str(res[3])
    assert not isinstance(res[4], Exception) and res[4].segs.shape[0] == 501


@pytest.mark.parametrize('samp_name', ['DNA', 'RNA'])
def test_side_stream_on_and_off_agree_on_gpu(samp_name):
    """tba_engine_set_side_stream: stall detection and expected levels beside normalisation / event
    detection on the engine's second stream, or everything in order on one -- the same bytes and the same
    statuses; a read with an invalid base AND too little signal for its sequence fails with the earlier
    (segmentation) error in both, as in the reference's order of calls (resquiggle.py:1160 before
    tombo_stats.py:858)"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th, _native as N
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    samp = th.seqSampleType(samp_name, False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    kw = synth.RNA_SYNTH if samp_name == 'RNA' else synth.DNA_SYNTH
    reads = []
    for seed, nb in enumerate((900, 1500, 400, 1200, 700, 1000)):
        seq, raw, _ = synth.synth_read(model, nb, 61000 + seed, **kw)
        if seed == 1:
            seq = seq[:300] + 'N' + seq[301:]                 # invalid base
        if seed == 2:
            seq, raw = seq[:100] + 'N' + seq[101:], raw[:raw.shape[0] // 30]   # ... and far too little signal
        if seed == 3 and samp_name == 'RNA':
            raw = np.concatenate([raw[:20000], np.full(1200, raw[20000]) +
                                  np.random.default_rng(3).normal(0, 3.0, 1200), raw[20000:]])
        reads.append((raw, seq))
    stall_kw = {}
    if samp_name == 'RNA':
        from tombo_amd._default_parameters import MEAN_STALL_PARAMS
        stall_kw = dict(stall_params=th.stallParams(**MEAN_STALL_PARAMS))
    eng = N.Engine(0)
    eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    p = N.make_params(params)
    o = N.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH[samp_name], subsample_seed=5, **stall_kw)
    outs = []
    for mode in (0, 1):
        eng.set_side_stream(mode)
        eng.upload(p, o, [r[0] for r in reads], [ts.encode_seq(r[1]) for r in reads])
        eng.run()
        assert eng.last_side_stream() == bool(mode)
        out = eng.download()
        outs.append({k: np.array(v, copy=True) for k, v in out.items() if isinstance(v, np.ndarray)})
    eng.set_side_stream(-1)
    a, b = outs
    assert a['status'].tolist() == b['status'].tolist()
    st = a['status'].tolist()
    assert st[1] == 22, st                                    # TBA_INVALID_SEQ
    assert st[2] not in (0, 22), st                           # the earlier error wins
    assert sum(x == 0 for x in st) >= 3, st
    for k in a:
        assert a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), k
    eng.close()


def test_big_rna_batch_is_the_same_every_run_and_the_oracles_on_gpu():
    """More wavefronts than the machine holds at once, four runs of one resident batch: the same bytes
    every time, and the oracle's on a sample of the late reads.  (Round 5: with 10 000 RNA reads a few
    wavefronts of k_main_tb_par<16> per run -- always among those dispatched after the first 1 024 --
    left speculative rows under their chunk tops, differently from run to run, while every 48-read
    parity test was green; round 6: profiles/r06_traceback_rootcause.txt, k_tb_par_verify, and the batches of
    tests/test_gpu_determinism.py at the sizes the fault showed at.)"""
    import zlib
    import bench
    from tombo_amd import _native as N, tombo_stats as ts, tombo_helper as th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH, STALL_PARAMS
    samp = th.seqSampleType('RNA', True)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
    n, nb = 4608, 1200
    seqs, raws, _ = bench.make_reads(np.full(n, nb, np.int64), 77000, min(32, os.cpu_count() or 8), 'RNA', False)
    rng = np.random.RandomState(3)
    si = np.stack([rng.choice(nb, 1000, replace=False) for _ in range(n)])
    eng = N.Engine(0)
    eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    eng.upload(N.make_params(params),
               N.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['RNA'],
                           stall_params=th.stallParams(**STALL_PARAMS)),
               raws, [ts.encode_seq(q) for q in seqs], samp_ind=si)
    runs = []
    for _ in range(4):
        eng.run()
        out = eng.download()
        tb = eng.get(N.GET_READ_TB)
        runs.append((zlib.crc32(tb.tobytes()), zlib.crc32(out['segs'].tobytes()), zlib.crc32(out['norm'].tobytes()),
                     out['status'].tobytes()))
    assert len(set(runs)) == 1, [r[:3] for r in runs]
    assert not eng.get(N.GET_TB_VERIFY_FAIL).any()
    form = eng.get(N.GET_TB_FORM)
    assert (form == N.TB_FORM_PAR16).sum() > 3500, np.bincount(form)
    segs = out['segs']
    checked = 0
    for i in list(range(4100, 4608, 23)) + [0, 1, 2047, 2048]:
        o = _oracle_read(model, params, raws[i], seqs[i], 5.0, 'RNA', stall_ints=oracle.identify_stalls(raws[i]),
                         samp_ind=si[i])
        assert o['status'] == int(out['status'][i]), (i, o['status'], int(out['status'][i]))
        if o['status'] == 0:
            np.testing.assert_array_equal(segs[eng.seg_off[i]:eng.seg_off[i + 1]], o['segs'], err_msg='read %d' % i)
            checked += 1
    assert checked >= 20
    eng.close()


def test_degenerate_scale_vs_oracle_on_gpu(dispatch_form):
    """Scale values the throughput form's loader cannot take through its reciprocal division: a MAD of
    exactly 0 (flat or saturated signal: the reference divides by it under np.seterr(all='raise') and
    dies with a FloatingPointError -- tests/golden/degenerate_cases.json; TBA_INTERNAL in both forms),
    and a signal scaled by 2^600 / 2^-600 (scale outside 2^-500 .. 2^500: k_normalize writes that read's
    normalised signal itself by true division and flags it for the kernels that keep the scores)"""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th, _native as N
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    seq, raw, _ = synth.synth_read(model, 700, 424242, **synth.DNA_SYNTH)
    sat = raw.copy()
    sat[np.argsort(raw)[:raw.shape[0] // 2 + 200]] = np.median(raw)    # more than half the samples equal
    reads = [(raw, seq, None, None), (np.full(raw.shape[0], 431.0), seq, None, None), (sat, seq, None, None),
             (raw * 2.0 ** 600, seq, None, None), (raw * 2.0 ** -600, seq, None, None), (raw, seq, None, None)]
    eng, out, oracles = run_batch(model, params, 'DNA', reads)
    bad = compare_batch(eng, oracles, out, 'degenerate')
    assert not bad, '\n'.join(bad[:20])
    assert [o['status'] for o in oracles] == [0, 100, 100, 0, 0, 0]
    ed, _ = check_forms(eng, dispatch_form, params)
    if dispatch_form == 'throughput':
        assert ed.tolist() == [N.ED_FORM_DETECT_PICK, N.ED_FORM_NONE, N.ED_FORM_NONE, N.ED_FORM_SCORES_PEAKS,
                               N.ED_FORM_SCORES_PEAKS, N.ED_FORM_DETECT_PICK]
    # exact powers of two: the scaled reads are the first read again, boundary for boundary
    s0, s1 = eng.seg_off[0], eng.seg_off[1]
    for k in (3, 4):
        np.testing.assert_array_equal(out['segs'][eng.seg_off[k]:eng.seg_off[k + 1]], out['segs'][s0:s1])
    # the same on int16 DAC values (k_normalize's integer-histogram medians; the reference's recorded
    # cases of tests/golden/degenerate_cases.json)
    from test_oracle_golden import degenerate_cases, degenerate_signal
    dac = np.round(raw).astype(np.int16)
    reads = [(dac, seq, None, None)]
    for case in degenerate_cases():
        if case['dtype'] == 'int16':
            assert case['raised'] == 'FloatingPointError'
            reads.append((degenerate_signal(case), seq, None, None))
    reads.append((dac[::-1].copy(), seq, None, None))
    eng, out, oracles = run_batch(model, params, 'DNA', reads, raw_dtype=np.int16)
    bad = compare_batch(eng, oracles, out, 'degenerate-i16')
    assert not bad, '\n'.join(bad[:20])
    assert [o['status'] for o in oracles][:3] == [0, 100, 100]


def test_big_batch_in_the_default_dispatch_vs_oracle_on_gpu():
    """One batch ABOVE both dispatch thresholds with the engine's defaults untouched -- 1 152 reads, what
    any batch of more than 1 024 reads (the 10 000-read benchmark batch) runs: k_normalize without
    its last pass, k_detect<2> + k_pick, k_main_tb_par<16> -- stage by stage against the oracle.
    400-900-base reads (adaptive path), a few static-path and failing reads, two 10 kb reads (90 k
    samples: 700 detector steps, 20-read workgroups with ragged lengths), int16-valued and flat ones."""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th, _native as N
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    rng = np.random.default_rng(2026)
    reads = []
    for i in range(1152):
        nb = int(rng.integers(400, 900))
        kw = dict(synth.DNA_SYNTH)
        if i % 97 == 0:
            nb = 10000
        elif i % 53 == 0:
            nb = int(rng.integers(20, 240))       # whole-read static band
        elif i % 41 == 0:
            kw['mean_dwell'] = 400                 # band failure
        elif i % 29 == 0:
            kw['noise_sd'] = 0.7
        seq, raw, _ = synth.synth_read(model, nb, 5150000 + i, **kw)
        if i % 31 == 0:
            raw = np.round(raw)                    # DAC-like values: exact score ties
        if i == 700:
            raw = np.full(raw.shape[0], 500.0)     # flat: MAD == 0
        reads.append((raw, seq, None, _si(nb, i)))
    eng = _engine()
    assert eng.get_dispatch() == (1024, 1024)
    eng, out, oracles = run_batch(model, params, 'DNA', reads)
    bad = compare_batch(eng, oracles, out, 'big')
    assert not bad, '%d mismatches\n' % len(bad) + '\n'.join(bad[:40])
    assert sum(o['status'] == 0 for o in oracles) >= 1000
    ed, tb = check_forms(eng, 'throughput', params)
    path = eng.get(N.GET_PATH)[:, 0]
    ok = np.array([o['status'] == 0 for o in oracles])
    assert (ed[ok] == N.ED_FORM_DETECT_PICK).mean() >= 0.95, np.bincount(ed)
    assert np.all(tb[ok & (path == 1)] == N.TB_FORM_PAR16), np.bincount(tb)
    assert (ok & (path == 1)).sum() >= 800 and (ok & (path == 2)).sum() >= 10


def test_wide_static_band_matches_oracle():
    """short reads with a lot of signal: find_static_base_assignment's band is n_events -
    mask_len cells wide (resquiggle.py:566-570), far beyond the register band classes ->
    k_dp_wide.  Reached in practice through the save-bandwidth retry (resquiggle.py:1587)."""
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    save = ts.load_resquiggle_parameters(samp, use_save_bandwidth=True)
    reads = []
    for nb, seed, kw in [(20, 9, dict(lead=60000)), (100, 31, dict(mean_dwell=300)),
                         (249, 32, dict(lead=30000)), (180, 33, {}), (1200, 34, {})]:
        seq, raw, _ = synth.synth_read(model, nb, seed, **kw)
        reads.append((raw, seq, None, _si(nb, seed)))
    eng, out, oracles = run_batch(model, save, 'DNA', reads)
    bad = compare_batch(eng, oracles, out, 'wide')
    assert not bad, '\n'.join(bad[:20])
    from tombo_amd import _native as N
    path = eng.get(N.GET_PATH)
    assert (path[:3, 0] == 2).all() and (path[:3, 2] > 3072).all(), path   # wide static bands
    assert path[3, 2] <= 3072 and path[4, 0] == 1
    assert sum(o['status'] == 0 for o in oracles) >= 4


def test_resquiggle_batch_without_signal():
    """return_signal=False: same boundaries, scale values and scores, raw_signal None, nothing
    materialised for it on the device"""
    from tombo_amd import resquiggle as rq, synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    mrs = [synth.synth_map_res(model, nb, 9100 + nb, **synth.DNA_SYNTH) for nb in (700, 1300, 20, 950)]
    a = rq.resquiggle_batch(mrs, model, params, 5.0, seq_samp_type=samp, subsample_seed=4)
    b = rq.resquiggle_batch(mrs, model, params, 5.0, seq_samp_type=samp, subsample_seed=4, return_signal=False)
    assert sum(not isinstance(x, Exception) for x in a) >= 3
    for x, y in zip(a, b):
        assert isinstance(x, Exception) == isinstance(y, Exception)
        if isinstance(x, Exception):
            assert str(x) == str(y)
            continue
        assert y.raw_signal is None and x.raw_signal is not None
        assert np.array_equal(x.segs, y.segs) and x.scale_values == y.scale_values
        assert x.sig_match_score == y.sig_match_score and x.read_start_rel_to_raw == y.read_start_rel_to_raw
