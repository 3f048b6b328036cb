"""Streaming path on the GPU: typed raw input (int16 / float32 / float64), enqueue-only upload /
download through page-locked buffers, several engine slots in flight, skip_norm_out, automatic
sub-batching.  Every result is compared bit for bit with the plain one-batch path
(`resquiggle_batch`, itself checked against the oracle in test_gpu_parity.py) and, for the
compact records, with the oracle directly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(n_reads=23, seed0=500, dac=False):
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    rng = np.random.RandomState(seed0)
    reads = []
    for i in range(n_reads):
        nb = int(rng.choice([300, 700, 1200, 1800]))
        seq, raw, _ = synth.synth_read(model, nb, seed0 + i, **synth.DNA_SYNTH)
        if dac:
            raw = np.round(raw / 0.1709 + 10.0).astype(np.int16)
        si = rng.choice(nb, 1000, replace=False).astype(np.int64) if nb > 1000 else None
        reads.append((seq, raw, si))
    return samp, model, params, reads


def _map_results(reads):
    from tombo_amd import tombo_helper as th
    return [th.resquiggleResults(
        align_info=th.alignInfo('r%d' % i, 'BaseCalled_template', 0, 0, 0, 0, len(s) - 5, 0),
        genome_loc=th.genomeLocation(0, '+', 'synth'), genome_seq=s, mean_q_score=10.0,
        raw_signal=r) for i, (s, r, _) in enumerate(reads)]


@pytest.mark.parametrize('dac', [False, True])
def test_stream_pipeline_equals_single_batches(dac):
    from tombo_amd import resquiggle as rq, streaming, tombo_stats as ts
    samp, model, params, reads = _setup(dac=dac)
    ref = rq.resquiggle_batch(_map_results(reads), model, params, outlier_thresh=5.0,
                              seq_samp_type=samp, samp_inds=[si for _, _, si in reads])
    cuts = [0, 5, 6, 13, 16, 20, 23]   # ragged batches, more batches than slots
    for want_norm, sd in ((False, np.int32), (True, np.int64)):
        pipe = streaming.StreamPipeline(model, params, n_slots=3, outlier_thresh=5.0,
                                        seq_samp_type=samp, want_norm=want_norm, segs_dtype=sd)
        batches = [streaming.ReadBatch.from_lists(
            [r for _, r, _ in reads[a:b]], [ts.encode_seq(s) for s, _, _ in reads[a:b]],
            samp_inds=[si for _, _, si in reads[a:b]], tag=(a, b), pinned=(a % 2 == 0))
            for a, b in zip(cuts[:-1], cuts[1:])]
        seen = []
        for res in pipe.run(batches):
            a, b = res.tag
            seen.append(res.tag)
            assert res.n == b - a
            for k in range(b - a):
                want = ref[a + k]
                rec = res.results[k]
                if isinstance(want, Exception):
                    assert rec['status'] != 0
                    continue
                assert rec['status'] == 0
                np.testing.assert_array_equal(res.segs_of(k).astype(np.int64), want.segs)
                assert rec['read_start_rel_to_raw'] == want.read_start_rel_to_raw
                assert rec['sig_match_score'] == want.sig_match_score
                assert rec['shift'] == want.scale_values.shift
                assert rec['scale'] == want.scale_values.scale
                assert rec['lower_lim'] == want.scale_values.lower_lim
                assert bool(rec['norm_params_changed']) == want.norm_params_changed
                assert rec['norm_len'] == want.raw_signal.shape[0]
                if want_norm:
                    np.testing.assert_array_equal(res.norm_of(k), want.raw_signal)
        assert seen == [(a, b) for a, b in zip(cuts[:-1], cuts[1:])]   # submission order
        pipe.close()


def test_compact_records_equal_oracle_on_int16():
    import oracle
    from tombo_amd import streaming, tombo_stats as ts
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    samp, model, params, reads = _setup(n_reads=9, seed0=900, dac=True)
    pipe = streaming.StreamPipeline(model, params, n_slots=2, outlier_thresh=5.0, seq_samp_type=samp)
    b = streaming.ReadBatch.from_lists([r for _, r, _ in reads], [ts.encode_seq(s) for s, _, _ in reads],
                                       samp_inds=[si for _, _, si in reads], pinned=True)
    res = list(pipe.run([b]))[0]
    p = oracle.make_params(params)
    o = oracle.make_opts(model.kmer_width, model.central_pos, outlier_thresh=5.0,
                         sig_match_thresh=SIG_MATCH_THRESH['DNA'])
    for k, (s, r, si) in enumerate(reads):
        want = oracle.resquiggle_read(r.astype(np.float64), ts.encode_seq(s), model.level_means,
                                      model.level_sds, p, o, samp_ind=si)
        assert res.results['status'][k] == want['status']
        if want['status'] == 0:
            np.testing.assert_array_equal(res.segs_of(k), want['segs'])
            assert res.results['sig_match_score'][k] == want['sig_match_score']
            assert res.results['read_start_rel_to_raw'][k] == want['read_start_rel_to_raw']
    pipe.close()


def test_skip_norm_out_keeps_events_table_and_refuses_norm_download():
    from tombo_amd import _native, resquiggle as rq, tombo_stats as ts
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    samp, model, params, reads = _setup(n_reads=6, seed0=40)
    eng = rq.get_engine(0)
    eng.ensure_model(model)
    p = _native.make_params(params)
    si = np.zeros((len(reads), 1000), np.int64)
    for i, (_, _, s) in enumerate(reads):
        if s is not None:
            si[i] = s
    outs = {}
    for skip in (False, True):
        o = _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA'],
                              skip_norm_out=skip)
        eng.upload(p, o, [r for _, r, _ in reads], [ts.encode_seq(s) for s, _, _ in reads], samp_ind=si)
        eng.run()
        outs[skip] = (eng.download(want_norm=False), eng.base_stats())
        if skip:
            with pytest.raises(_native.EngineError):
                eng.download_async(norm=np.zeros(eng.n_raw_total))
    a, b = outs[False], outs[True]
    np.testing.assert_array_equal(a[0]['segs'], b[0]['segs'])
    np.testing.assert_array_equal(a[0]['score'], b[0]['score'])
    np.testing.assert_array_equal(a[1][0], b[1][0])   # per-base means
    np.testing.assert_array_equal(a[1][1], b[1][1])   # per-base sds


def test_auto_sub_batching_gives_the_one_batch_result():
    from tombo_amd import resquiggle as rq
    samp, model, params, reads = _setup(n_reads=17, seed0=70)
    mrs = _map_results(reads)
    sis = [si for _, _, si in reads]
    one = rq.resquiggle_batch(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp, samp_inds=sis)
    # a budget that only fits a handful of these reads at a time
    cut = rq.resquiggle_batch(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp,
                              samp_inds=sis, mem_budget=5.6e8)
    assert len(one) == len(cut) == len(mrs)
    for x, y in zip(one, cut):
        assert isinstance(x, Exception) == isinstance(y, Exception)
        if not isinstance(x, Exception):
            np.testing.assert_array_equal(x.segs, y.segs)
            np.testing.assert_array_equal(x.raw_signal, y.raw_signal)
            assert x.sig_match_score == y.sig_match_score


def test_streamed_resquiggle_batch_gives_the_one_batch_result(monkeypatch):
    """a long list goes through three engines in rotation (pack / compute / download + unpack
    overlapped, resquiggle._stream_batches): same results, same order, failures included; with the
    host draw the Theil-Sen subsamples come off numpy's RNG in read order either way"""
    from tombo_amd import resquiggle as rq
    samp, model, params, reads = _setup(n_reads=41, seed0=170)
    mrs = _map_results(reads)
    mrs[5] = mrs[5]._replace(raw_signal=mrs[5].raw_signal[:400])     # a read that fails
    monkeypatch.setenv('TBA_API_STREAM', '0')
    np.random.seed(11)
    one = rq.resquiggle_batch(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp)
    monkeypatch.setenv('TBA_API_STREAM', '1')
    monkeypatch.setenv('TBA_API_STREAM_MIN', '12')
    np.random.seed(11)
    cut = rq.resquiggle_batch(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp)
    nosig = rq.resquiggle_batch(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp,
                                subsample_seed=3, return_signal=False)
    assert len(one) == len(cut) == len(nosig) == len(mrs)
    assert any(isinstance(x, Exception) for x in one)
    for x, y, z in zip(one, cut, nosig):
        assert isinstance(x, Exception) == isinstance(y, Exception) == isinstance(z, Exception)
        if isinstance(x, Exception):
            assert str(x) == str(y)
            continue
        np.testing.assert_array_equal(x.segs, y.segs)
        np.testing.assert_array_equal(x.raw_signal, y.raw_signal)
        assert x.sig_match_score == y.sig_match_score and x.scale_values == y.scale_values
        assert x.genome_seq == y.genome_seq and x.read_start_rel_to_raw == y.read_start_rel_to_raw
        assert z.raw_signal is None and z.segs.shape == x.segs.shape
    # the device-side draw of a read is keyed by its index in the LIST: cut into sub-batches or not (and
    # however), a seeded call gives the same subsamples, hence the same results bit for bit
    monkeypatch.setenv('TBA_API_STREAM', '0')
    one_dev = rq.resquiggle_batch(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp, subsample_seed=3)
    monkeypatch.setenv('TBA_API_STREAM', '1')
    for cuts in ('3', '5'):
        monkeypatch.setenv('TBA_API_CUTS', cuts)
        cut_dev = rq.resquiggle_batch(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp, subsample_seed=3)
        for x, y in zip(one_dev, cut_dev):
            assert isinstance(x, Exception) == isinstance(y, Exception)
            if not isinstance(x, Exception):
                np.testing.assert_array_equal(x.segs, y.segs)
                np.testing.assert_array_equal(x.raw_signal, y.raw_signal)
                assert x.scale_values == y.scale_values and x.sig_match_score == y.sig_match_score


def test_results_are_views_of_pooled_page_locked_blocks(monkeypatch):
    """resquiggle_batch hands out views into page-locked blocks of the process-wide result pool (no
    second copy): the blocks stay leased while any result looks into them -- a later call does not
    overwrite them -- return to the pool when the results go, and are reused by the next call; a stream
    that dies half way leaves the engines idle and the default engine's sharing hint reset"""
    import gc
    from tombo_amd import resquiggle as rq, _native
    samp, model, params, reads = _setup(n_reads=72, seed0=4100)
    mrs = _map_results(reads)
    pool = _native.result_pool()
    gc.collect()
    leased0 = pool.leased_bytes
    monkeypatch.setenv('TBA_API_STREAM_MIN', '24')
    monkeypatch.setenv('TBA_API_ZERO_COPY_MIN', '1')    # (default: sub-batches of 32 reads and more)
    a = rq.resquiggle_batch(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp, subsample_seed=9)
    ok = [r for r in a if not isinstance(r, Exception)]
    assert len(ok) >= 60
    assert all(r.raw_signal.base is not None and r.segs.base is not None for r in ok)   # views, not copies
    assert pool.leased_bytes > leased0
    snap = [(r.segs.copy(), r.raw_signal.copy()) for r in ok]
    b = rq.resquiggle_batch(mrs[::-1], model, params, outlier_thresh=5.0, seq_samp_type=samp, subsample_seed=10)
    for r, (sg, sig) in zip(ok, snap):       # the first call's results are untouched by the second
        np.testing.assert_array_equal(r.segs, sg)
        np.testing.assert_array_equal(r.raw_signal, sig)
    keep = ok[3].raw_signal                  # one view keeps its block leased
    del a, b, ok, r
    gc.collect()
    assert pool.leased_bytes > leased0
    idle_before = pool.idle_bytes
    assert idle_before > 0
    del keep
    gc.collect()
    assert pool.leased_bytes == leased0
    c = rq.resquiggle_batch(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp, subsample_seed=9)
    assert pool.idle_bytes < idle_before + (1 << 20)    # served from the idle blocks, nothing new allocated
    for r, (sg, sig) in zip([x for x in c if not isinstance(x, Exception)], snap):
        np.testing.assert_array_equal(r.segs, sg)
        np.testing.assert_array_equal(r.raw_signal, sig)
    # a stream that raises half way
    calls = {'n': 0}
    real = rq._submit_batch

    def boom(*a_, **k_):
        calls['n'] += 1
        if calls['n'] == 2:
            raise _native.EngineError('injected')
        return real(*a_, **k_)
    monkeypatch.setattr(rq, '_submit_batch', boom)
    with pytest.raises(_native.EngineError, match='injected'):
        rq.resquiggle_batch(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp, subsample_seed=9)
    monkeypatch.setattr(rq, '_submit_batch', real)
    assert not rq.get_engine(0).query()                      # nothing in flight
    d = rq.resquiggle_batch(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp, subsample_seed=9)
    for x, y in zip(c, d):
        assert isinstance(x, Exception) == isinstance(y, Exception)
        if not isinstance(x, Exception):
            np.testing.assert_array_equal(x.segs, y.segs)
    rq.release_stream_engines()
    assert rq._STREAM_ENGINES == {}


def test_put_rejects_indices_outside_the_signal():
    from tombo_amd import _native, resquiggle as rq, tombo_stats as ts
    samp, model, params, reads = _setup(n_reads=1, seed0=11)
    eng = rq.get_engine(0)
    eng.ensure_model(model)
    seq, raw, _ = reads[0]
    eng.set_num_events([40])
    eng.upload(_native.make_params(params), _native.make_opts(), [raw], [ts.encode_seq(seq)])
    good = np.arange(0, 400, 10, dtype=np.int64)
    for bad in (good[::-1].copy(), good + raw.shape[0], np.where(np.arange(40) == 7, good[6], good)):
        with pytest.raises(_native.EngineError):
            eng.put(_native.PUT_VALID_CPTS, bad, per_read=[40])
    eng.put(_native.PUT_VALID_CPTS, good, per_read=[40])
    nseg = int(eng.seg_off[-1])
    segs = np.linspace(0, 300, nseg).astype(np.int64)
    with pytest.raises(_native.EngineError):
        eng.put(_native.PUT_DP_SEGS, segs[::-1].copy(), per_read=[0, 300])
    with pytest.raises(_native.EngineError):
        eng.put(_native.PUT_DP_SEGS, segs, per_read=[0, 200])      # boundaries past norm_len
    eng.put(_native.PUT_DP_SEGS, segs, per_read=[0, 300])


def test_failed_upload_does_not_leave_forced_event_counts_armed():
    from tombo_amd import _native, resquiggle as rq, tombo_stats as ts
    samp, model, params, reads = _setup(n_reads=1, seed0=12)
    eng = rq.get_engine(0)
    eng.ensure_model(model)
    seq, raw, _ = reads[0]
    eng.set_num_events([7])
    with pytest.raises(_native.EngineError):   # start bandwidth above TBA_MAX_BAND: refused
        eng.upload(_native.make_params(params._replace(start_bw=5000)), _native.make_opts(),
                   [raw], [ts.encode_seq(seq)])
    eng.upload(_native.make_params(params), _native.make_opts(outlier_thresh=5.0), [raw],
               [ts.encode_seq(seq)])
    eng.run_stages(_native.STAGE_SEGMENT, _native.STAGE_SEGMENT)
    assert int(eng.get(_native.GET_N_CPTS)[0]) == int(eng.num_events[0]) != 7


def test_long_read_lists_are_cut_and_oversized_batches_refused():
    """more reads than one batch may hold (a launch-grid dimension): resquiggle_batch cuts the
    list on its own, the raw engine call refuses"""
    from tombo_amd import _native, resquiggle as rq, planner, synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    base = [synth.synth_map_res(model, 150, 9000 + i, **synth.DNA_SYNTH) for i in range(8)]
    n = planner.MAX_READS + 700
    mrs = [base[i % 8] for i in range(n)]
    res = rq.resquiggle_batch(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp)
    assert len(res) == n
    ref = rq.resquiggle_batch(base, model, params, outlier_thresh=5.0, seq_samp_type=samp)
    for i in (0, 5, planner.MAX_READS - 1, planner.MAX_READS, n - 1):
        a, b = res[i], ref[i % 8]
        assert isinstance(a, Exception) == isinstance(b, Exception)
        if not isinstance(a, Exception):
            np.testing.assert_array_equal(a.segs, b.segs)
            assert a.sig_match_score == b.sig_match_score
    eng = rq.get_engine(0)
    raws = [np.zeros(64)] * 65536
    seqs = [np.zeros(8, np.uint8)] * 65536
    with pytest.raises(_native.EngineError):
        eng.upload(_native.make_params(params), _native.make_opts(), raws, seqs)
