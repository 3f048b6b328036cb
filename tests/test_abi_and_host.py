"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header
declares, the host mirror of the reference interface behaves, errors map to the reference's
strings, and the product path fails loudly without a GPU."""
import os
import re
import ctypes

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def test_library_exports_every_declared_symbol():
    from tombo_amd import _native
    _native.build()
    hdr = open(os.path.join(ROOT, 'include', 'tombo_amd.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = sorted(set(re.findall(r'\b(tba_[a-z0-9_]+)\s*\(', hdr)))
    assert len(declared) >= 18
    lib = ctypes.CDLL(_native.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, 'declared in include/tombo_amd.h but not exported: %s' % missing


def test_struct_layouts_match_header():
    """ctypes mirrors of tba_params / tba_opts have the header's field order"""
    from tombo_amd import _native
    hdr = open(os.path.join(ROOT, 'include', 'tombo_amd.h')).read()
    body = hdr[hdr.index('typedef struct {', hdr.index('th.resquiggleParams')):hdr.index('} tba_params;')]
    names = re.findall(r'\b([a-z_]+)\s*[,;]', re.sub(r'/\*.*?\*/', '', body, flags=re.S))
    assert names == [f[0] for f in _native.Params._fields_]
    body = hdr[hdr.index('typedef struct {', hdr.index('} tba_params;')):hdr.index('} tba_opts;')]
    names = re.findall(r'\b([a-z_]+)\s*[,;]', re.sub(r'/\*.*?\*/', '', body, flags=re.S))
    assert names == [f[0] for f in _native.Opts._fields_]


def test_struct_sizes_match_the_built_library():
    import ctypes as C
    from tombo_amd import _native
    out = (C.c_int64 * 3)()
    assert _native.lib().tba_abi_sizes(out, C.c_int64(3)) == 0
    assert list(out) == [C.sizeof(_native.Params), C.sizeof(_native.Opts), C.sizeof(_native.ReadResult)]


def test_pack_reads_host_packer():
    """tba_pack_reads (host only): CSR copy of per-read arrays by native threads, optional flip,
    ACGT -> 0..3 (anything else 255), every boundary dtype, ragged / empty reads"""
    from tombo_amd import _native, tombo_stats as ts
    rng = np.random.default_rng(11)
    for dt in (np.int16, np.float32, np.float64):
        lens = [0, 1, 5, 1000, 3, 77777, 12, 0, 40000] + [int(x) for x in rng.integers(1, 3000, 200)]
        raws = [(rng.normal(0, 100, n)).astype(dt) for n in lens]
        seqs = [''.join(rng.choice(list('ACGT'), int(rng.integers(0, 300)))) for _ in lens]
        seqs[3] = 'ACGTNacgtRYACGT'
        for rev in (False, True):
            for nt in (1, 7):
                raw, raw_off, seq, seq_off, _ = _native.pack_reads(raws, seqs, reverse=rev, n_threads=nt)
                assert raw.dtype == dt
                want = np.concatenate([r[::-1] if rev else r for r in raws])
                assert np.array_equal(raw, want)
                assert np.array_equal(np.diff(raw_off), lens)
                assert np.array_equal(seq, np.concatenate([ts.encode_seq(s) for s in seqs]))
                assert np.array_equal(np.diff(seq_off), [len(s) for s in seqs])
    # mixed dtypes fall back to float64; bytes sequences are taken as they are
    raw, raw_off, seq, seq_off, _ = _native.pack_reads(
        [np.arange(5, dtype=np.int16), np.arange(3, dtype=np.float32)], [b'ACGT', 'TTG'])
    assert raw.dtype == np.float64 and np.array_equal(raw, [0, 1, 2, 3, 4, 0, 1, 2])
    assert np.array_equal(seq, [0, 1, 2, 3, 3, 3, 2])


def test_no_cpu_fallback_without_gpu():
    from tombo_amd import _native
    lib = _native.lib()
    if lib.tba_device_count() > 0:
        pytest.skip('a GPU is visible here')
    with pytest.raises(_native.EngineError, match='no CPU fallback'):
        _native.Engine(0)
    from tombo_amd import resquiggle as rq, tombo_stats as ts, tombo_helper as th, synth
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    mr = synth.synth_map_res(model, 300, 1, **synth.DNA_SYNTH)
    with pytest.raises(_native.EngineError):
        rq.resquiggle_read(mr, model, ts.load_resquiggle_parameters(samp), 5.0)


def test_error_table_is_the_reference_taxonomy():
    from tombo_amd import errors, tombo_helper as th
    hdr = open(os.path.join(ROOT, 'include', 'tombo_amd.h')).read()
    codes = dict((int(v), k) for k, v in re.findall(r'(TBA_[A-Z_]+) = (\d+),', hdr))
    for code in errors.MESSAGES:
        assert code in codes
    assert errors.MESSAGES[10] == 'Adaptive signal to seqeunce alignment extended beyond raw signal'
    assert errors.MESSAGES[12] == 'Discordant reference and seqeunce lengths.'
    with pytest.raises(th.TomboError, match='extends beyond bandwidth'):
        errors.raise_for_status(11)
    with pytest.raises(RuntimeError):
        errors.raise_for_status(100)
    errors.raise_for_status(0)


def test_params_and_model_match_reference_values():
    from tombo_amd import tombo_stats as ts, tombo_helper as th
    dna = ts.load_resquiggle_parameters(th.seqSampleType('DNA', False))
    assert dna == th.resquiggleParams(
        4.2, 4.2, 300, 20.0, 5, 3, 1, 5, 4.9978845608028655, 4.2, False, 40, 750, 2500, 250)
    rna = ts.load_resquiggle_parameters(th.seqSampleType('RNA', False), use_save_bandwidth=True)
    assert (rna.bandwidth, rna.use_t_test_seg, rna.start_bw) == (1500, True, 1000)
    assert float.hex(rna.z_shift - 6) == float.hex(float.fromhex('0x1.9884533d43651p-1') + 6 - 6)
    m = ts.TomboModel(seq_samp_type=th.seqSampleType('DNA', False))
    assert (m.kmer_width, m.central_pos) == (6, 2)
    mu, sd = m.get_exp_levels_from_seq('ACGTACGTAC')
    assert mu.shape == (5,) and mu[0] == m.means['ACGTAC'] and sd[4] == m.sds['ACGTAC']
    assert ts.compute_num_events(92067, 10000, 5) == 18413
    assert ts.compute_num_events(1000, 10000, 5) == 11000
    with pytest.raises(th.TomboError):
        m.get_exp_levels_from_seq('ACGTNCGTAC')
    r = ts.TomboModel(seq_samp_type=th.seqSampleType('RNA', False))
    assert (r.kmer_width, r.central_pos) == (5, 1)


def test_resquiggle_results_tuple_is_field_compatible():
    from tombo_amd import tombo_helper as th
    assert th.resquiggleResults._fields == (
        'align_info', 'genome_loc', 'genome_seq', 'mean_q_score', 'raw_signal', 'channel_info',
        'read_start_rel_to_raw', 'segs', 'scale_values', 'sig_match_score',
        'norm_params_changed', 'start_clip_bases', 'stall_ints')
    assert th.scaleValues._fields == ('shift', 'scale', 'lower_lim', 'upper_lim', 'outlier_thresh')
    assert th.dpResults._fields == ('read_start_rel_to_raw', 'segs', 'ref_means', 'ref_sds',
                                    'genome_seq')
    mr = th.resquiggleResults('a', 'b', 'ACGT', 1.0)
    assert mr.raw_signal is None and mr.stall_ints is None


def test_remove_stall_cpts_walk():
    from tombo_amd import tombo_stats as ts
    cpts = np.array([5, 10, 20, 30, 40, 50, 60, 100], dtype=np.int64)
    out = ts.remove_stall_cpts([(8, 25), (45, 60)], cpts)
    assert out.tolist() == [5, 30, 40, 60, 100]
    assert ts.remove_stall_cpts([], cpts) is cpts


def test_worker_loop_control_flow_without_gpu(monkeypatch):
    """resquiggle_batch_iters = `_resquiggle_worker`'s per-read loop (resquiggle.py:1492-1504,
    1578-1589): re-runs while norm_params_changed (at most max_scaling_iters passes, fitted scale
    values handed on, const_scale / skip_seq_scaling only on the first pass), then every failed
    read started over with the save parameters.  The engine is replaced by a scripted stub."""
    from tombo_amd import resquiggle as rq, tombo_helper as th
    calls = []
    # per read: outcomes of successive passes with the main parameters, then with the save ones
    script = {
        'a': (['ok'], []),
        'b': (['changed', 'changed', 'ok'], []),
        'c': (['changed', 'changed', 'changed', 'changed'], []),       # capped at 3 passes
        'd': (['fail'], ['changed', 'ok']),
        'e': (['changed', 'fail'], ['fail']),
    }
    seen = {}

    def fake_batch(map_results, std_ref, params, outlier_thresh=None, all_raw_signals=None,
                   const_scale=None, skip_seq_scaling=False, seq_samp_type=None, engine=None, **kw):
        calls.append((params, [m.align_info for m in map_results], const_scale, skip_seq_scaling,
                      [m.scale_values for m in map_results], all_raw_signals is not None))
        out = []
        for m in map_results:
            rid = m.align_info
            k = (rid, params)
            seen[k] = seen.get(k, 0) + 1
            plan = script[rid][0 if params == 'main' else 1]
            what = plan[seen[k] - 1]
            if what == 'fail':
                out.append(th.TomboError('boom %s' % rid))
            else:
                out.append(m._replace(segs=[0, 1], scale_values=('sv', rid, params, seen[k]),
                                      norm_params_changed=(what == 'changed')))
        return out
    monkeypatch.setattr(rq, 'resquiggle_batch', fake_batch)
    mrs = [th.resquiggleResults(align_info=r, genome_loc=None, genome_seq='ACGT', mean_q_score=1.0,
                                raw_signal='raw-' + r) for r in 'abcde']
    res, passes = rq.resquiggle_batch_iters(mrs, None, 'main', 'save', outlier_thresh=5.0,
                                            const_scale=12.0, skip_seq_scaling=True,
                                            return_passes=True)
    assert passes == [1, 3, 3, 1 + 2, 2 + 1]
    assert [isinstance(r, Exception) for r in res] == [False, False, False, False, True]
    assert res[2].norm_params_changed and res[2].scale_values == ('sv', 'c', 'main', 3)
    assert res[3].scale_values == ('sv', 'd', 'save', 2) and str(res[4]) == 'boom e'
    # first pass: everyone, with const_scale / skip_seq_scaling; re-runs: only the changed ones,
    # carrying the fitted scale values and the un-normalised signal, options not forwarded
    assert calls[0][:4] == ('main', list('abcde'), 12.0, True) and calls[0][4] == [None] * 5
    assert calls[1][:4] == ('main', list('bce'), None, False) and calls[1][5]
    assert calls[1][4] == [('sv', r, 'main', 1) for r in 'bce']
    assert calls[2][:2] == ('main', list('bc'))
    # the save-parameter round starts the failed reads over (d failed pass 1, e failed pass 2)
    assert calls[3][:4] == ('save', list('de'), 12.0, True) and calls[3][4] == [None, None]
    assert calls[4][:2] == ('save', ['d']) and len(calls) == 5


def test_events_table_layout():
    """the Events structured array of write_new_fast5_group (tombo_helper.py:2353-2360)"""
    import numpy as np
    from tombo_amd import tombo_helper as th
    res = th.resquiggleResults(align_info=None, genome_loc=None, genome_seq='ACGTA',
                               mean_q_score=1.0, segs=np.array([0, 3, 7, 8, 15, 20]))
    tab = th.events_table(res, np.arange(5) * 0.5, np.arange(5) * 0.1)
    assert tab.dtype == np.dtype([('norm_mean', 'f8'), ('norm_stdev', 'f8'), ('start', 'u4'),
                                  ('length', 'u4'), ('base', 'S1')])
    assert tab['start'].tolist() == [0, 3, 7, 8, 15] and tab['length'].tolist() == [3, 4, 1, 7, 5]
    assert b''.join(tab['base']) == b'ACGTA' and tab['norm_mean'][4] == 2.0
    assert np.isnan(th.events_table(res, np.zeros(5))['norm_stdev']).all()


def test_planner_cuts_by_memory_and_orders_by_length():
    """host planner (no GPU needed: tba_batch_footprint is a host-only entry point)"""
    import numpy as np
    from tombo_amd import planner, _native, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
    p = _native.make_params(params)
    o = _native.make_opts(outlier_thresh=5.0, skip_norm_out=True)
    rng = np.random.default_rng(3)
    B = np.clip(np.exp(rng.normal(np.log(8000), 0.9, 3000)), 1000, 100000).astype(np.int64)
    S, L = B * 9 + 300, B + 5
    budget = 8e9
    plan = planner.plan_batches(S, L, p, o, 6, budget, np.int16)
    assert sorted(np.concatenate(plan).tolist()) == list(range(3000))       # every read once
    for idx in plan:
        assert planner.exact_bytes(S[idx], L[idx], p, o, 6, np.int16) <= budget or len(idx) == 1
        assert np.all(np.diff(B[idx]) <= 0)                                 # longest first inside
    longest = [int(B[idx].max()) for idx in plan]
    assert longest == sorted(longest, reverse=True)        # longest critical path first
    assert B[plan[0]].max() == B.max()
    # the vectorised estimate tracks the engine's own figure
    est = planner.estimate_bytes(S, L, p, o, 6, np.int16).sum()
    exact = planner.exact_bytes(S, L, p, o, 6, np.int16)
    assert abs(est - exact) / exact < 0.05
    # input order kept on request; int16 input is smaller than float64
    plan2 = planner.plan_batches(S, L, p, o, 6, budget, np.int16, sort=False)
    assert np.array_equal(np.concatenate(plan2), np.arange(3000))
    assert planner.exact_bytes(S, L, p, o, 6, np.int16) < planner.exact_bytes(S, L, p, o, 6, np.float64)


def test_stream_pipeline_slot_logic_with_stub_engines(monkeypatch):
    """host logic of streaming.StreamPipeline without a GPU: slots are used round-robin, a slot's
    previous batch is finished before it is reused, results come back in submission order and the
    two output sets of a slot alternate (a handed-out result is not overwritten by the next batch
    of its slot)"""
    import numpy as np
    from tombo_amd import streaming, _native, tombo_stats as ts, tombo_helper as th

    log = []

    class StubEngine(object):
        n_made = 0

        def __init__(self, device):
            self.id = StubEngine.n_made
            StubEngine.n_made += 1
            self.kmer_width = 6

        def ensure_model(self, m):
            pass

        def set_sharing(self, n_engines):
            pass

        def upload_packed(self, p, o, raw, raw_off, seq, seq_off, **kw):
            self.n = len(raw_off) - 1
            self.raw_off = np.asarray(raw_off)
            self.seg_off = np.arange(self.n + 1) * 3
            self.n_raw_total = int(raw_off[-1])
            self.tag = int(raw[0])
            log.append(('up', self.id, self.tag))

        def enqueue(self):
            log.append(('run', self.id, self.tag))

        def download_async(self, results=None, segs32=None, segs64=None, norm=None):
            self._out = (results, segs32)
            results['status'][:self.n] = 0
            results['read_start_rel_to_raw'][:self.n] = self.tag   # "computed" at enqueue time
            segs32[:int(self.seg_off[-1])] = self.tag

        def sync(self):
            log.append(('sync', self.id, self.tag))

        def get(self, what):
            return np.zeros(32, np.float32)

        def close(self):
            pass

    class StubPinned(object):
        def __init__(self, shape, dtype):
            self.a = np.zeros(shape, dtype)

        def close(self):
            pass

    monkeypatch.setattr(_native, 'Engine', StubEngine)
    monkeypatch.setattr(_native, 'PinnedArray', StubPinned)
    samp = th.seqSampleType('DNA', False)
    params = ts.load_resquiggle_parameters(samp)
    pipe = streaming.StreamPipeline(None, params, n_slots=2, device=0, outlier_thresh=5.0)
    batches = [streaming.ReadBatch(np.full(4, t, np.int16), np.array([0, 2, 4]), np.zeros(8, np.uint8),
                                   np.array([0, 4, 8]), tag=t) for t in range(5)]
    held = []
    for res in pipe.run(batches):
        held.append(res)
        # every result still shows its own batch, also the ones handed out earlier that are at
        # most n_slots batches old
        for r in held[-2:]:
            assert int(r.results['read_start_rel_to_raw'][0]) == r.tag
            assert int(r.segs[0]) == r.tag
    assert [r.tag for r in held] == [0, 1, 2, 3, 4]
    ups = [e for e in log if e[0] == 'up']
    assert [e[1] for e in ups] == [0, 1, 0, 1, 0]                  # round-robin over two slots
    # a slot is synced (its old batch finished) before it is uploaded again
    for k, e in enumerate(log):
        if e[0] == 'up' and e[2] >= 2:
            assert ('sync', e[1], e[2] - 2) in log[:k]
    pipe.close()


def test_planner_tail_batches():
    """reads above tail_bases get batches of their own, in front; every read exactly once"""
    from tombo_amd import planner, _native, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    p = _native.make_params(ts.load_resquiggle_parameters(samp)._replace(bandwidth=500))
    o = _native.make_opts(outlier_thresh=5.0)
    rng = np.random.default_rng(4)
    B = np.clip(np.exp(rng.normal(np.log(8000.0), 0.9, 3000)), 1000, 200000).astype(np.int64)
    S, L = B * 9 + 300, B + 5
    plan = planner.plan_batches(S, L, p, o, 6, 3e9, tail_bases=30000)
    assert np.array_equal(np.sort(np.concatenate(plan)), np.arange(3000))
    n_tail = int((B > 30000).sum())
    assert 0 < n_tail < 3000
    seen = 0
    for idx in plan:   # tail batches first, and never mixed with ordinary reads
        is_tail = B[idx] > 30000
        assert is_tail.all() or not is_tail.any()
        if seen < n_tail:
            assert is_tail.all()
        seen += len(idx)
    assert planner.plan_batches(S, L, p, o, 6, 3e9, tail_bases=10 ** 9)[0].shape[0] > 0   # no tail: plain plan


def test_stream_pipeline_any_slot_and_feeder_staging(monkeypatch):
    """in_order=False: a batch takes the first slot that is free or has finished (never waiting
    behind a long one while another is idle), results come back as they complete; ReadFeeder hands
    a staging set out again only after the batch packed into it was finished"""
    from tombo_amd import streaming, _native, tombo_stats as ts, tombo_helper as th
    clock = [0.0]

    class StubEngine(object):
        def __init__(self, device):
            self.kmer_width = 6
            self.busy_until = 0.0

        def ensure_model(self, m):
            pass

        def set_sharing(self, n_engines):
            pass

        def upload_packed(self, p, o, raw, raw_off, seq, seq_off, **kw):
            self.n = len(raw_off) - 1
            self.raw_off = np.asarray(raw_off)
            self.seg_off = np.arange(self.n + 1) * 3
            self.n_raw_total = int(raw_off[-1])
            self.dur = float(raw[0])          # the batch's first sample says how long it "runs"

        def enqueue(self):
            self.busy_until = clock[0] + self.dur

        def download_async(self, results=None, **kw):
            results['status'][:self.n] = 0

        def query(self):
            return clock[0] < self.busy_until

        def sync(self):
            clock[0] = max(clock[0], self.busy_until)

        def get(self, what):
            return np.zeros(32, np.float32)

        def close(self):
            pass

    class StubPinned(object):
        def __init__(self, shape, dtype):
            self.a = np.zeros(shape, dtype)

        def close(self):
            pass
    monkeypatch.setattr(_native, 'Engine', StubEngine)
    monkeypatch.setattr(_native, 'PinnedArray', StubPinned)
    monkeypatch.setattr(streaming.time, 'sleep', lambda dt: clock.__setitem__(0, clock[0] + 1.0))
    params = ts.load_resquiggle_parameters(th.seqSampleType('DNA', False))
    pipe = streaming.StreamPipeline(None, params, n_slots=3, device=0, outlier_thresh=5.0, in_order=False)
    pipe.reserve(8, 64)
    feeder = streaming.ReadFeeder(n_slots=3)
    assert len(feeder.stages) == 5
    done_tags = []
    durs = [100, 5, 5, 5, 5, 5, 5]          # batch 0 is the long one
    batches = []
    for k, d in enumerate(durs):
        b = feeder.pack([np.full(4, d, np.int16), np.full(3, d, np.int16)], ['ACGTACGTAC', 'ACGTACGTACG'], tag=k)
        batches.append(b)
        out = pipe.submit(b)
        if out is not None:
            done_tags.append(out.tag)
    done_tags += [r.tag for r in pipe.flush()]
    assert sorted(done_tags) == list(range(len(durs)))
    assert done_tags[:4] == [1, 2, 3, 4] and done_tags.index(0) >= 4   # the short ones overtook the long one
    assert clock[0] < 140                                           # nobody queued up behind it
    # staging: 7 batches went through 5 sets without growing -- sets of finished batches were reused,
    # and the set of the long batch was not handed out while it was still in flight
    assert len(feeder.stages) == 5 and not any(feeder._busy)
    feeder.close()
    pipe.close()


def test_pinned_pool_budget_counts_what_it_allocates_and_divides_by_the_node_ranks(monkeypatch):
    """_native.PinnedPool without a device (the page-locked allocation stubbed): the budget is checked against the bytes a
    fresh block really takes (need + 1/16 + 4096), idle blocks of the wrong size make room, a block comes back when its
    last view dies, and the default budget is a quarter of the host divided by the ranks of the node (page-locked memory
    cannot be swapped: round-5 advisor finding)"""
    import ctypes
    import gc
    from tombo_amd import _native

    made, closed = [], []

    class FakePinned(object):
        def __init__(self, n, dtype):
            self.nbytes = int(n)
            self._buf = ctypes.create_string_buffer(self.nbytes)
            self._ptr = ctypes.c_void_p(ctypes.addressof(self._buf))
            made.append(self.nbytes)

        def close(self):
            closed.append(self.nbytes)

    monkeypatch.setattr(_native, 'PinnedArray', FakePinned)
    monkeypatch.setenv('TBA_PINNED_POOL_BYTES', str(3 << 20))
    pool = _native.PinnedPool()
    assert pool.budget == 3 << 20
    a = pool.lease(1 << 18, np.float64)              # 2 MiB + 1/16 + 4096
    assert a is not None and made == [(2 << 20) + (2 << 20) // 16 + 4096] and pool.leased_bytes == made[0]
    assert pool.lease(1 << 17, np.float64) is None   # 1 MiB + ... would pass a check against `need` alone: 2 + 1 <= 3
    assert pool.leased_bytes == made[0] and len(made) == 1
    del a
    gc.collect()
    assert pool.leased_bytes == 0 and pool.idle_bytes == made[0]
    b = pool.lease(1 << 18, np.float64)              # the idle block again
    assert b is not None and len(made) == 1 and pool.idle_bytes == 0
    del b
    gc.collect()
    c = pool.lease(327680, np.float64)               # 2.5 MiB: too large for the idle block, and no room beside it: it goes
    assert c is not None and closed == [made[0]] and pool.idle_bytes == 0 and pool.leased_bytes == made[1]
    monkeypatch.delenv('TBA_PINNED_POOL_BYTES')
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    one = _native.PinnedPool().budget
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '1')
    assert _native.PinnedPool().budget >= one and one <= (16 << 30)
    host = os.sysconf('SC_PHYS_PAGES') * os.sysconf('SC_PAGE_SIZE')
    assert one == min(host // 4 // 8, 16 << 30)
