"""GPU: the per-kernel C ABI entry points (tba_c_*, bound with the Cython module's names in
tombo_amd/_c_dynamic_programming.py and _c_helper.py) against the oracle's kernel-level
restatements, on data from a synthetic read."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def read():
    import oracle
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    seq, raw, _ = synth.synth_read(model, 700, 4242, **synth.DNA_SYNTH)
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    o = oracle.resquiggle_read(
        raw, ts.encode_seq(seq), model.level_means, model.level_sds, oracle.make_params(params),
        oracle.make_opts(6, 2, outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA']),
        debug=True)
    assert o['status'] == 0
    mu, sd = model.get_exp_levels_from_seq(seq)
    return dict(model=model, params=params, raw=raw, seq=seq, o=o, mu=mu, sd=sd)


def test_elementwise_kernels(read):
    import oracle
    from tombo_amd import _c_dynamic_programming as cdp, _c_helper as ch
    d = read['o']['dbg']
    sig = d['seg_norm_signal']
    z = cdp.c_base_z_scores(sig[:5000], 0.3, 0.35, True, 2.5)
    np.testing.assert_array_equal(z, oracle.base_z_scores(sig[:5000], 0.3, 0.35, True, 2.5))
    z = cdp.c_base_z_scores(sig[:100], -1.0, 0.7)
    np.testing.assert_array_equal(z, oracle.base_z_scores(sig[:100], -1.0, 0.7, False, 10.0))
    np.testing.assert_array_equal(ch.c_new_means(sig, d['valid_cpts']), d['event_means'])
    np.testing.assert_array_equal(ch.c_apply_outlier_thresh(sig, -0.5, 0.75),
                                  oracle.apply_outlier_thresh(sig, -0.5, 0.75))
    with pytest.raises(ValueError):
        ch.c_new_means(sig.astype(np.float32), d['valid_cpts'])


def test_change_point_kernels(read):
    import oracle
    from tombo_amd import _c_helper as ch
    d = read['o']['dbg']
    sig = d['seg_norm_signal']
    n = len(d['valid_cpts'])
    np.testing.assert_array_equal(ch.c_valid_cpts_w_cap(sig, 3, 5, n), d['valid_cpts'])
    rc, want = oracle.valid_cpts(read['raw'], 6, 12, 400, ttest=True)
    assert rc == 0
    np.testing.assert_array_equal(ch.c_valid_cpts_w_cap_t_test(read['raw'], 6, 12, 400), want)
    # more change points than the signal can hold: the reference's error
    with pytest.raises(NotImplementedError, match='Fewer changepoints found than requested'):
        ch.c_valid_cpts_w_cap(sig[:600], 3, 5, 400)
    rc, _ = oracle.valid_cpts(sig[:600], 3, 5, 400)
    assert rc == 2


def test_forward_pass_and_traceback_kernels(read):
    import oracle
    from tombo_amd import _c_dynamic_programming as cdp
    p = read['params']
    ev = read['o']['dbg']['event_means']
    mu, sd = read['mu'], read['sd']
    # static band, as find_seq_start_in_events builds it (resquiggle.py:708-724)
    nb, bw = 120, 300
    z = np.empty((nb, bw))
    for r in range(nb):
        z[r] = p.z_shift - np.minimum(p.max_half_z_score, np.abs(ev[r:r + bw] - mu[r]) / sd[r])
    starts = np.arange(nb, dtype=np.int64)
    fwd, tb = cdp.c_banded_forward_pass(z, starts, p.skip_pen, p.stay_pen)
    ofwd, otb = oracle.banded_forward_pass(z, starts, p.skip_pen, p.stay_pen)
    np.testing.assert_array_equal(fwd, ofwd)
    np.testing.assert_array_equal(tb[1:], otb[1:])
    top = int(np.argmax(fwd[-1]))
    rc, want = oracle.banded_traceback(otb, starts, top)
    assert rc == 0
    np.testing.assert_array_equal(cdp.c_banded_traceback(tb, starts, top), want)
    with pytest.raises(NotImplementedError, match='extends beyond bandwidth'):
        cdp.c_banded_traceback(tb, starts, top, 140)
    # adaptive continuation from the static rows, in place (pyx:314-412)
    n_bases = 500
    fwd_a = np.zeros((n_bases + 1, bw))
    tb_a = np.zeros((n_bases + 1, bw), dtype=np.int64)
    st_a = np.zeros(n_bases, dtype=np.int64)
    fwd_a[:nb + 1], tb_a[:nb + 1], st_a[:nb] = fwd, tb, starts
    o_fwd, o_tb, o_st = fwd_a.copy(), tb_a.astype(np.int8), st_a.copy()
    rc = oracle.adaptive_banded_forward_pass(o_fwd, o_tb, o_st, ev, mu[:n_bases], sd[:n_bases],
                                             p.z_shift, p.skip_pen, p.stay_pen, nb, -15.0, True,
                                             p.max_half_z_score)
    assert rc == 0
    ret = cdp.c_adaptive_banded_forward_pass(
        fwd_a, tb_a, st_a, ev, mu[:n_bases], sd[:n_bases], p.z_shift, p.skip_pen, p.stay_pen, nb,
        -15.0, True, p.max_half_z_score)
    assert ret is None
    np.testing.assert_array_equal(st_a, o_st)
    np.testing.assert_array_equal(fwd_a, o_fwd)
    np.testing.assert_array_equal(tb_a[1:], o_tb[1:].astype(np.int64))
    # band start pushed past the last event (argmax at the right edge of a row that already sits
    # at the end of the events): the reference's error, pyx:349-357
    fwd_b, tb_b, st_b = fwd_a.copy(), tb_a.copy(), st_a.copy()
    fwd_b[nb] = np.arange(bw, dtype=np.float64)
    st_b[nb - 1] = ev.shape[0] - 10
    o_fwd, o_tb, o_st = fwd_b.copy(), tb_b.astype(np.int8), st_b.copy()
    rc = oracle.adaptive_banded_forward_pass(o_fwd, o_tb, o_st, ev, mu[:n_bases], sd[:n_bases],
                                             p.z_shift, p.skip_pen, p.stay_pen, nb, -15.0, True,
                                             p.max_half_z_score)
    assert rc == 10
    with pytest.raises(NotImplementedError, match='extended beyond raw signal'):
        cdp.c_adaptive_banded_forward_pass(
            fwd_b, tb_b, st_b, ev, mu[:n_bases], sd[:n_bases], p.z_shift, p.skip_pen,
            p.stay_pen, nb, -15.0, True, p.max_half_z_score)


@pytest.mark.parametrize('bw', [64, 100, 257, 500, 700, 1000, 1500, 1610, 2000, 2500, 3072])
def test_row_engine_every_band_class(bw):
    """c_banded_forward_pass over random z-scores and random band offsets for every
    cells-per-lane instantiation of the wave-per-read kernel (CPL 4..48), noise-like scores with
    many exact ties (winsorised z) included"""
    import oracle
    from tombo_amd import _c_dynamic_programming as cdp
    rng = np.random.default_rng(bw)
    nb = 40
    z = 5.0 - np.abs(rng.normal(0, 6, size=(nb, bw)))
    z = np.maximum(z, -15.0)                       # winsorised floor -> exact ties
    z[rng.random(z.shape) < 0.02] = -15.0
    starts = np.cumsum(rng.integers(0, 4, size=nb)).astype(np.int64)
    starts[:5] = 0
    fwd, tb = cdp.c_banded_forward_pass(z, starts, 4.2, 4.2)
    ofwd, otb = oracle.banded_forward_pass(z, starts, 4.2, 4.2)
    np.testing.assert_array_equal(fwd, ofwd)
    np.testing.assert_array_equal(tb[1:], otb[1:])
    top = int(np.argmax(fwd[-1]))
    rc, want = oracle.banded_traceback(otb, starts, top)
    assert rc == 0
    np.testing.assert_array_equal(cdp.c_banded_traceback(tb, starts, top), want)


def test_row_constant_division_is_ieee():
    """div_by_recip (k_dp's z-score division by a per-row sd) == a / b bit for bit"""
    import ctypes as C
    from tombo_amd import resquiggle as rq
    eng = rq.get_engine(0)
    rng = np.random.default_rng(3)
    n = 1 << 21
    a = np.concatenate([
        rng.normal(0, 3, n // 4), rng.normal(0, 1e-3, n // 4), rng.uniform(-30, 30, n // 4),
        rng.normal(0, 1, n // 4) * 10.0 ** rng.uniform(-12, 6, n // 4)])
    b = np.concatenate([
        np.full(n // 4, 0.3528720791815676), np.full(n // 4, 0.2252531482690988),
        rng.uniform(0.05, 3.0, n // 4), 10.0 ** rng.uniform(-3, 3, n // 4)])
    a[:64] = 0.0
    a[64:128] = np.nextafter(b[64:128], 10)          # quotients next to 1
    a[128:192] = b[128:192] * (1 + 2.0 ** -52)
    out = np.empty(n)
    pd = C.POINTER(C.c_double)
    rc = eng._L.tba_selftest_division(eng._h, a.ctypes.data_as(pd), b.ctypes.data_as(pd),
                                      C.c_int64(n), out.ctypes.data_as(pd))
    assert rc == 0
    want = a / b
    bad = np.flatnonzero(out != want)
    assert bad.size == 0, (bad[:5], a[bad[:5]], b[bad[:5]], out[bad[:5]], want[bad[:5]])
