"""Row P10 on the device: the worker's per-read preparation (resquiggle.py:1506-1530) -- RNA flip,
ts.identify_stalls (tombo_stats.py:269-368), and the opt-in device-side Theil-Sen subsample --
against the numpy restatement in oracle/ (pinned on the `stall_ints` the live reference left in
the RNA fixtures) and against batches that are handed the same inputs from the host."""
import numpy as np
import pytest

import oracle
from conftest import golden_names

pytestmark = pytest.mark.gpu


def _engine():
    from tombo_amd import resquiggle as rq
    return rq.get_engine(0)


def _ints(x):
    return np.array([[int(a), int(b)] for a, b in x], dtype=np.int64).reshape(-1, 2)


def _rna_like(rng, n, n_stalls, scale=90.0):
    from tombo_amd import synth
    return synth.stalled_signal(rng, n, n_stalls, scale)


def test_identify_stalls_matches_restatement_on_random_reads():
    """>= 100 random reads incl. inserted stalls, float64 / float32 / int16 boundary types; very
    short reads (below the window: []), stalls at the read ends"""
    from tombo_amd import tombo_stats as ts
    rng = np.random.default_rng(20260927)
    n_with = 0
    for k in range(120):
        n = int(rng.integers(200, 60000)) if k % 10 else int(rng.integers(1, 700))
        raw = _rna_like(rng, n, int(rng.integers(0, 4)))
        if k % 7 == 0 and raw.shape[0] > 2000:   # a stall running into the end / from the start
            raw[-900:] = raw[-900] + rng.normal(0, 2.0, 900)
            raw[:700] = raw[0] + rng.normal(0, 2.0, 700)
        want = _ints(oracle.identify_stalls(raw))
        got = _ints(ts.identify_stalls(raw))
        assert np.array_equal(got, want), (k, got, want)
        n_with += len(want) > 0
        if k % 3 == 0:   # the DAC form: what the FAST5 `Signal` dataset holds
            dac = np.round(raw).astype(np.int16)
            want = _ints(oracle.identify_stalls(dac.astype(np.float64)))
            assert np.array_equal(_ints(ts.identify_stalls(dac)), want), k
            f32 = raw.astype(np.float32)
            want = _ints(oracle.identify_stalls(f32.astype(np.float64)))
            assert np.array_equal(_ints(ts.identify_stalls(f32)), want), k
    assert n_with >= 40


def test_identify_stalls_other_parameters():
    """generic n_windows path of the metric kernel, no widening (expand <= 0: unmerged runs)"""
    from tombo_amd import tombo_stats as ts, tombo_helper as th
    rng = np.random.default_rng(5)
    for sp in (th.stallParams(window_size=200, threshold=30.0, min_consecutive_obs=120,
                              edge_buffer=100, mini_window_size=40, n_windows=5),
               th.stallParams(window_size=90, threshold=50.0, min_consecutive_obs=60,
                              edge_buffer=10, mini_window_size=30, n_windows=3),
               th.stallParams(window_size=64, threshold=45.0, min_consecutive_obs=0,
                              edge_buffer=40, mini_window_size=4, n_windows=16)):
        for _ in range(6):
            raw = _rna_like(rng, int(rng.integers(3000, 30000)), 3)
            want = _ints(oracle.identify_stalls(raw, sp))
            got = _ints(ts.identify_stalls(raw, sp))
            assert np.array_equal(got, want), (sp, got[:5], want[:5])


@pytest.mark.parametrize('name', [n for n in golden_names() if n.startswith('rna_')])
def test_identify_stalls_vs_reference_recorded_intervals(golden_case, name):
    from tombo_amd import tombo_stats as ts
    c = golden_case(name)
    want = c.g['stall_ints'] if 'stall_ints' in c.g else np.zeros((0, 2), np.int64)
    assert np.array_equal(_ints(ts.identify_stalls(c.raw)), want)


def _rna_batch(n_reads=10, seed0=77000):
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('RNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    reads = []
    rng = np.random.default_rng(3)
    for k in range(n_reads):
        nb = int(rng.integers(300, 2600))
        seq, raw, _ = synth.synth_read(model, nb, seed0 + k, **synth.RNA_SYNTH)
        if k % 2 == 0:   # stalled stretches
            a = int(rng.integers(1000, raw.shape[0] - 4000))
            raw = np.concatenate([raw[:a], raw[a] + rng.normal(0, 3.0, 1800), raw[a:]])
        st = np.random.get_state()
        np.random.seed(k)
        si = np.random.choice(nb, 1000, replace=False).astype(np.int64) if nb > 1000 else None
        np.random.set_state(st)
        reads.append((raw, seq, si))
    return samp, model, params, reads


@pytest.mark.parametrize('dtype', [np.float64, np.int16])
def test_batch_with_device_prep_equals_host_prepared_batch(dtype):
    """acquisition-order samples + reverse_raw + detect_stalls == a batch handed the flipped signal
    and the restatement's stall intervals; and both == the oracle"""
    from tombo_amd import _native as N, tombo_stats as ts, tombo_helper as th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH, STALL_PARAMS
    samp, model, params, reads = _rna_batch()
    if dtype == np.int16:
        reads = [(np.round(r / 0.1709 + 10.0).astype(np.int16), s, si) for r, s, si in reads]
    eng = _engine()
    eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    p = N.make_params(params)
    si = np.zeros((len(reads), 1000), np.int64)
    for i, r in enumerate(reads):
        if r[2] is not None:
            si[i] = r[2]
    stalls = [oracle.identify_stalls(np.asarray(r[0], np.float64)) for r in reads]
    assert sum(len(s) > 0 for s in stalls) >= 3
    seqs = [ts.encode_seq(r[1]) for r in reads]
    # host-prepared
    o_host = N.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['RNA'])
    eng.upload(p, o_host, [r[0] for r in reads], seqs, samp_ind=si, stall_ints=stalls)
    eng.run()
    a = eng.download()
    a_cpts, a_n = eng.get(N.GET_VALID_CPTS).copy(), eng.get(N.GET_N_CPTS).copy()
    # device-prepared, from the samples as acquired (3'->5')
    o_dev = N.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['RNA'],
                        reverse_raw=True, stall_params=th.stallParams(**STALL_PARAMS))
    eng.upload(p, o_dev, [np.ascontiguousarray(r[0][::-1]) for r in reads], seqs, samp_ind=si)
    eng.run()
    b = eng.download()
    got = eng.stall_ints()
    for i, s in enumerate(stalls):
        assert np.array_equal(got[i], _ints(s)), i
    assert np.array_equal(eng.get(N.GET_N_CPTS), a_n)
    assert np.array_equal(eng.get(N.GET_VALID_CPTS), a_cpts)
    for k in ('status', 'segs', 'read_start', 'norm_len', 'sv', 'score', 'changed', 'norm'):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert (a['status'] == 0).sum() >= len(reads) - 1
    # a second run of the same resident batch detects the same intervals again
    eng.run()
    again = eng.stall_ints()
    assert all(np.array_equal(x, y) for x, y in zip(got, again))
    # the oracle, on the flipped signal with the restatement's intervals
    from test_gpu_parity import _oracle_read
    for i, r in enumerate(reads):
        o = _oracle_read(model, params, np.asarray(r[0], np.float64), r[1], 5.0, 'RNA',
                         stall_ints=stalls[i], samp_ind=r[2])
        assert o['status'] == b['status'][i]
        if o['status'] == 0:
            assert np.array_equal(o['segs'], b['segs'][eng.seg_off[i]:eng.seg_off[i + 1]])


def test_reverse_raw_dna_float32():
    """the flip alone (any boundary type)"""
    from tombo_amd import _native as N, synth, tombo_stats as ts, tombo_helper as th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    rs = [synth.synth_read(model, nb, 4100 + nb, **synth.DNA_SYNTH) for nb in (700, 401, 950)]
    raws = [r[1].astype(np.float32) for r in rs]
    seqs = [ts.encode_seq(r[0]) for r in rs]
    eng = _engine()
    eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    p = N.make_params(params)
    outs = []
    for rev in (False, True):
        o = N.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA'], reverse_raw=rev)
        eng.upload(p, o, [np.ascontiguousarray(r[::-1]) if rev else r for r in raws], seqs)
        eng.run()
        outs.append(eng.download())
    assert (outs[0]['status'] == 0).all()
    for k in ('status', 'segs', 'read_start', 'norm', 'sv', 'score'):
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), k


def test_device_subsample_is_a_valid_draw_and_feeds_the_same_fit():
    """tba_opts.device_subsample: 1000 distinct in-range indices per read, a function of (seed,
    read index) only; the fit they produce is what the oracle computes from the same indices"""
    from tombo_amd import _native as N, synth, tombo_stats as ts, tombo_helper as th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    from test_gpu_parity import _oracle_read
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    nbs = (1500, 900, 2600, 1001, 4000)
    rs = [synth.synth_read(model, nb, 5200 + nb, **synth.DNA_SYNTH) for nb in nbs]
    eng = _engine()
    eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    p = N.make_params(params)
    drawn = []
    for seed in (1, 1, 2):
        o = N.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA'], subsample_seed=seed)
        eng.upload(p, o, [r[1] for r in rs], [ts.encode_seq(r[0]) for r in rs])
        eng.run()
        out = eng.download()
        drawn.append(eng.get(N.GET_SAMP_IND).copy())
    assert (out['status'] == 0).all()
    assert np.array_equal(drawn[0], drawn[1])
    for i, nb in enumerate(nbs):
        if nb <= 1000:
            continue
        si = drawn[2][i]
        assert si.min() >= 0 and si.max() < nb and np.unique(si).shape[0] == 1000
        assert not np.array_equal(drawn[0][i], si)            # another seed, another draw
        o = _oracle_read(model, params, rs[i][1], rs[i][0], 5.0, 'DNA', samp_ind=si)
        assert o['status'] == 0
        assert np.array_equal(o['segs'], out['segs'][eng.seg_off[i]:eng.seg_off[i + 1]])
        assert np.array_equal(o['scale_values'][:2], out['sv'][i][:2])
        nl = int(out['norm_len'][i])
        assert np.array_equal(o['norm_signal'], out['norm'][eng.raw_off[i]:eng.raw_off[i] + nl])
    # reads 0 and 2 got different keys although in the same batch
    assert not np.array_equal(drawn[0][0] % 1000, drawn[0][2] % 1000)


def test_device_subsample_uniformity():
    """the keyed permutation: a bijection of [0, n) for awkward n, and its first 1000 images are
    spread like a uniform draw without replacement (chi-square over 50 bins, 400 keys; marginal
    inclusion frequency of single indices)"""
    import ctypes as C
    eng = _engine()
    L = eng._L

    def perm(n, seed, ri, count):
        out = np.zeros(count, np.int64)
        rc = L.tba_selftest_subsample(eng._h, C.c_int64(n), C.c_uint64(seed), C.c_int64(ri),
                                      C.c_int64(count), out.ctypes.data_as(C.POINTER(C.c_int64)))
        assert rc == 0
        return out
    for n in (1, 2, 3, 1001, 4096, 4097, 10000, 65537, 1 << 20):
        pm = perm(n, 99, 5, min(n, 200000))
        if pm.shape[0] == n:
            assert np.array_equal(np.sort(pm), np.arange(n)), n
        else:
            assert np.unique(pm).shape[0] == pm.shape[0] and pm.min() >= 0 and pm.max() < n
    n, bins, keys = 10000, 50, 400
    cnt = np.zeros(bins)
    hits = np.zeros(n)
    for k in range(keys):
        si = perm(n, 12345, k, 1000)
        cnt += np.bincount(si * bins // n, minlength=bins)
        hits[si] += 1
    exp = keys * 1000.0 / bins
    chi2 = ((cnt - exp) ** 2 / exp).sum()
    assert chi2 < 100.0, chi2            # 49 dof: mean 49, sd 9.9; 100 is > 5 sd
    # every index is drawn with probability 0.1 per key: binomial(400, 0.1), mean 40, sd 6
    assert abs(hits.mean() - 40.0) < 1e-9 and 4.5 < hits.std() < 7.5, (hits.mean(), hits.std())
    assert hits.max() < 80 and hits.min() > 10


def test_stall_detection_of_a_long_read_in_a_batch():
    """a read past TBA_LONG_RAW samples takes the workgroup-per-read cumulative sum
    (k_cumsum_scores_long, MODE 1) under the stall detector: float64 and int16"""
    from tombo_amd import _native as N, synth, tombo_stats as ts, tombo_helper as th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH, STALL_PARAMS
    samp = th.seqSampleType('RNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    rng = np.random.default_rng(8)
    reads = []
    for k, nb in enumerate((7000, 500, 6400)):
        seq, raw, _ = synth.synth_read(model, nb, 31000 + k, **synth.RNA_SYNTH)
        for _ in range(3):
            a = int(rng.integers(1000, raw.shape[0] - 5000))
            raw = np.concatenate([raw[:a], raw[a] + rng.normal(0, 3.0, int(rng.integers(300, 2500))), raw[a:]])
        reads.append((raw, seq))
    assert max(r[0].shape[0] for r in reads) > 262144
    eng = _engine()
    eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    for dt in (np.float64, np.int16):
        raws = [r[0] if dt == np.float64 else np.round(r[0]).astype(np.int16) for r in reads]
        o = N.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['RNA'],
                        stall_params=th.stallParams(**STALL_PARAMS), subsample_seed=3)
        eng.upload(N.make_params(params), o, raws, [ts.encode_seq(r[1]) for r in reads])
        eng.run()
        got = eng.stall_ints()
        for i, raw in enumerate(raws):
            want = _ints(oracle.identify_stalls(np.asarray(raw, np.float64)))
            assert len(want) >= 1 and np.array_equal(got[i], want), (dt, i)
