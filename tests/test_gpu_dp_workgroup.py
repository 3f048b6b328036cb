"""The workgroup-per-read form of the main forward pass (csrc/k_dp_wgm.h) against the wavefront-per-read
form (k_dp / k_dp_multi, themselves held to the oracle by test_gpu_parity.py): same band starts, same
last row, same path, same final results, bit for bit -- over the band classes it covers, reads that
fail inside the pass, short reads on the static path, and the long reads of a larger batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reads(model, n, seed0, lengths, edits=()):
    from tombo_amd import synth, tombo_stats as ts
    rng = np.random.RandomState(seed0)
    raws, seqs, si = [], [], []
    for i in range(n):
        nb = int(lengths[i % len(lengths)])
        seq, raw, starts = synth.synth_read(model, nb, seed0 + i, **synth.DNA_SYNTH)
        if i < len(edits) and edits[i]:
            seq, raw = synth.edit_read(seq, raw, starts, edits[i])
            nb = len(seq) - model.kmer_width + 1
        raws.append(raw)
        seqs.append(ts.encode_seq(seq))
        si.append(rng.choice(nb, 1000, replace=False).astype(np.int64) if nb > 1000 else np.zeros(1000, np.int64))
    return raws, seqs, np.array(si)


def _run(model, params, raws, seqs, si, wg_batch):
    from tombo_amd import _native
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    eng = _native.Engine(0)
    eng.ensure_model(model)
    eng.set_dp_workgroup_batch(wg_batch)
    eng.upload(_native.make_params(params), _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA']),
               raws, seqs, samp_ind=si)
    eng.run()
    out = eng.download(want_norm=True)
    out['band_starts'] = eng.get(_native.GET_BAND_STARTS)
    out['last_row'] = eng.get(_native.GET_LAST_ROW)
    out['read_tb'] = eng.get(_native.GET_READ_TB)
    out['by_wg'] = eng.get(_native.GET_DP_WORKGROUP)
    out['path'] = eng.get(_native.GET_PATH)
    out['ref_off'], out['seg_off'], out['raw_off'] = eng.ref_off.copy(), eng.seg_off.copy(), eng.raw_off.copy()
    eng.close()
    return out


def _same(a, b, W):
    assert np.array_equal(a['status'], b['status'])
    ok = a['status'] == 0
    assert np.array_equal(a['path'], b['path'])
    for i in np.flatnonzero(ok):
        s = slice(a['ref_off'][i], a['ref_off'][i + 1])
        assert np.array_equal(a['band_starts'][s], b['band_starts'][s]), i
        t = slice(a['seg_off'][i], a['seg_off'][i + 1])
        assert np.array_equal(a['read_tb'][t], b['read_tb'][t]), i
        assert np.array_equal(a['segs'][t], b['segs'][t]), i
        Wi = int(a['path'][i, 2])          # (the read's own band: a short read's is its whole event range)
        assert np.array_equal(a['last_row'][i, :Wi], b['last_row'][i, :Wi]), i
    for k in ('read_start', 'norm_len', 'sv', 'score', 'changed'):
        assert np.array_equal(a[k][ok], b[k][ok]), k
    for i in np.flatnonzero(ok):   # (behind a read's norm_len samples, and in a failed read's slice, the buffer is whatever it held)
        u = slice(a['raw_off'][i], a['raw_off'][i] + a['norm_len'][i])
        assert np.array_equal(a['norm'][u], b['norm'][u]), i


@pytest.mark.parametrize('W', [500, 300, 200, 129, 512, 384])
def test_workgroup_form_equals_wavefront_form(W):
    from tombo_amd import tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=W)
    edits = [None, dict(kind='truncate', frac=0.6), dict(kind='cut', n=40), dict(kind='insert', n=60, seed=3),
             dict(kind='truncate', frac=0.97)]
    raws, seqs, si = _reads(model, 19, 7100 + W, [900, 1500, 260, 2600, 700, 120, 3300], edits)
    a = _run(model, params, raws, seqs, si, 384)
    b = _run(model, params, raws, seqs, si, -1)
    assert b['by_wg'].sum() == 0
    # every read whose band (the batch bandwidth, or the whole-read band of a short read) has 129..512
    # cells went through the workgroup kernel
    ok, Wr = a['status'] == 0, a['path'][:, 2]
    assert ok.sum() >= 12
    assert np.array_equal(a['by_wg'][ok] == 1, ((Wr > 128) & (Wr <= 512))[ok]) and a['by_wg'][ok].sum() >= 10
    _same(a, b, W)


def test_long_reads_of_a_larger_batch_take_the_workgroup_form():
    """mode 1: batch above the threshold, only its is_long reads (more than TBA_LONG_BASES bases) go
    to the workgroup kernel, the rest to k_dp -- same results as all by k_dp"""
    from tombo_amd import tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    raws, seqs, si = _reads(model, 12, 9900, [400, 52000, 800, 300])
    a = _run(model, params, raws, seqs, si, 4)       # 12 reads > 4: long reads only
    b = _run(model, params, raws, seqs, si, -1)
    lens = np.array([len(s) for s in seqs])
    assert np.array_equal(a['by_wg'] == 1, (lens > 50000) & (a['status'] == 0)) and a['by_wg'].sum() == 3
    _same(a, b, 500)
