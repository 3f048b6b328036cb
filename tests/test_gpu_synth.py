"""The device-side read generator (tba_synth_*, csrc/k_synth.h) -- the input of the distinct-read
job of bench.py --preset cfg5: bit for bit the numpy restatement (synth.device_reads_reference), a
read depends on (seed, its index in the job) alone, and a batch handed to the engine as device
pointers gives the results of the same reads uploaded from host arrays (and the oracle's)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model():
    from tombo_amd import tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    return samp, ts.TomboModel(seq_samp_type=samp), ts.load_resquiggle_parameters(samp)


@pytest.mark.parametrize('dtype,reverse', [(np.int16, False), (np.float64, False), (np.int16, True)])
def test_device_batch_equals_numpy_restatement(dtype, reverse):
    from tombo_amd import _native, synth
    samp, model, _ = _model()
    kw = dict(mean_dwell=9, min_dwell=2, scale=12.0, offset=90.0, noise_sd=0.25, n_lead=200, n_trail=100)
    if reverse:
        kw = dict(mean_dwell=43, min_dwell=6, scale=80.0, offset=500.0, noise_sd=0.25, n_lead=37, n_trail=0)
    n_bases = [1, 2, 257, 300, 1000, 2500, 64, 255, 256]
    g = _native.Synth(model, 0)
    sp = _native.make_synth_params(reverse=reverse, **kw)
    raw, raw_off, seq, seq_off = g.generate(sp, 0xfeedfacecafebeef, n_bases, raw_dtype=dtype, first_read=12345678901)
    h_raw, h_seq = g.download()
    want_raw, want_seq = synth.device_reads_reference(model, 0xfeedfacecafebeef, n_bases, raw_dtype=dtype,
                                                      first_read=12345678901, reverse=reverse, **kw)
    assert raw.size == raw_off[-1] == sum(len(r) for r in want_raw)
    for i in range(len(n_bases)):
        assert np.array_equal(h_seq[seq_off[i]:seq_off[i + 1]], want_seq[i]), i
        got = h_raw[raw_off[i]:raw_off[i + 1]]
        assert got.dtype == np.dtype(dtype) and np.array_equal(got, want_raw[i]), i
    g.close()


def test_a_read_is_a_function_of_seed_and_job_index_alone():
    from tombo_amd import _native
    samp, model, _ = _model()
    sp = _native.make_synth_params()
    g, g2 = _native.Synth(model, 0), _native.Synth(model, 0)
    nb = [400, 900, 400, 1300, 700, 400, 400, 2000]
    _, ro, _, so = g.generate(sp, 99, nb)
    raw_all, seq_all = g.download()
    _, ro_a, _, so_a = g2.generate(sp, 99, nb[:3])
    raw_a, seq_a = g2.download()
    _, ro_b, _, so_b = g2.generate(sp, 99, nb[3:], first_read=3)
    raw_b, seq_b = g2.download()
    assert np.array_equal(np.concatenate([raw_a, raw_b]), raw_all)
    assert np.array_equal(np.concatenate([seq_a, seq_b]), seq_all)
    # ... and every read of a job is another read, every seed another job
    _, _, _, _ = g2.generate(sp, 100, nb)
    raw_c, seq_c = g2.download()
    assert not np.array_equal(seq_c[:400], seq_all[:400])
    reads = [seq_all[so[i]:so[i] + 300].tobytes() for i in range(len(nb))]
    assert len(set(reads)) == len(nb)
    g.close(), g2.close()


def test_device_batch_through_the_engine():
    """device pointers in place of host arrays: same results as the host upload of the same reads,
    and the oracle's on the same reads"""
    import oracle
    from tombo_amd import _native, tombo_stats as ts
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    samp, model, params = _model()
    sp = _native.make_synth_params()
    g = _native.Synth(model, 0)
    nb = [300, 700, 450, 999, 1000, 620]
    raw, raw_off, seq, seq_off = g.generate(sp, 5, nb)
    h_raw, h_seq = g.download()
    p = _native.make_params(params)
    o = _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA'])
    outs = []
    for r_, s_ in ((raw, seq), (h_raw, h_seq)):
        eng = _native.Engine(0)
        eng.ensure_model(model)
        eng.upload_packed(p, o, r_, raw_off, s_, seq_off)
        eng.run()
        d = eng.download(want_norm=True)
        outs.append((d, d['segs']))
        eng.close()
    (d0, segs0), (d1, segs1) = outs
    assert np.array_equal(d0['status'], d1['status']) and (d0['status'] == 0).sum() >= 5
    assert np.array_equal(segs0, segs1) and np.array_equal(d0['norm'], d1['norm'])
    seg_off = np.concatenate([[0], np.cumsum(np.asarray(nb) + 1)])
    for i in range(len(nb)):
        want = oracle.resquiggle_read(
            h_raw[raw_off[i]:raw_off[i + 1]].astype(np.float64), h_seq[seq_off[i]:seq_off[i + 1]],
            model.level_means, model.level_sds, oracle.make_params(params),
            oracle.make_opts(model.kmer_width, model.central_pos, outlier_thresh=5.0,
                             sig_match_thresh=SIG_MATCH_THRESH['DNA']))
        assert want['status'] == d0['status'][i], i
        if want['status'] == 0:
            assert np.array_equal(want['segs'], segs0[seg_off[i]:seg_off[i + 1]]), i
    g.close()


def test_a_batch_of_a_job_gives_the_same_result_whoever_runs_it():
    """what the job checksum of bench.py --preset cfg5 rests on: device-made batches keyed by their
    first read, the Theil-Sen draw keyed by the batch (ReadBatch.subsample_seed) -- submission order
    and slot count do not show in the results"""
    from tombo_amd import _native, streaming
    samp, model, params = _model()
    sp = _native.make_synth_params()
    gens = [_native.Synth(model, 0) for _ in range(4)]
    n_b, per = 4, 6

    def run(order, n_slots):
        pipe = streaming.StreamPipeline(model, params, n_slots=n_slots, outlier_thresh=5.0, seq_samp_type=samp,
                                        want_norm=False, segs_dtype=np.int32, subsample_seed=77)
        out = {}
        batches = []
        for i, k in enumerate(order):
            raw, raw_off, seq, seq_off = gens[i % 4].generate(sp, 77, [1500, 900, 2100, 1200, 1800, 1000][:per],
                                                              first_read=k * per)
            b = streaming.ReadBatch(raw, raw_off, seq, seq_off, tag=k)
            b.subsample_seed = 1000 + k
            batches.append(b)
        for res in pipe.run(batches):
            out[res.tag] = (res.results.copy(), np.asarray(res.segs).copy())
        pipe.close()
        return out

    a = run([0, 1, 2, 3], 2)
    b = run([2, 0, 3, 1], 3)
    assert sorted(a) == sorted(b) == list(range(n_b))
    n_ok = 0
    for k in range(n_b):
        assert np.array_equal(a[k][0], b[k][0]) and np.array_equal(a[k][1], b[k][1]), k
        n_ok += int((a[k][0]['status'] == 0).sum())
    assert n_ok >= n_b * per - 2
    # ... and another key is another draw for the reads that have one (more than 1000 bases)
    assert not np.array_equal(a[0][0]['shift'], a[1][0]['shift'])
    for g in gens:
        g.close()
