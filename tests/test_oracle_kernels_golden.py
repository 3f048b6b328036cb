"""CPU: the oracle's stand-alone raw-signal DP / helper kernels against vectors recorded from the
reference's compiled Cython functions (tests/golden/gen_golden_kernels.py)."""
import os
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def kt():
    return np.load(os.path.join(HERE, 'golden', 'kernels_tail.npz'))


def rag(g, key, i):
    o = g[key + '_off']
    return g[key][o[i]:o[i + 1]]


def region_cases(g):
    for name in g['rz_names']:
        p = 'rz_%s_' % name
        rs, re_, mbs, m = (int(x) for x in g[p + 'args'])
        mh = float(g[p + 'mh'][0])
        yield p, rs, re_, mbs, m, None if np.isnan(mh) else mh


def test_reg_z_scores(kt):
    import oracle
    for p, rs, re_, mbs, m, mh in region_cases(kt):
        res = oracle.reg_z_scores(kt[p + 'sig'], kt[p + 'means'], kt[p + 'sds'], kt[p + 'starts'],
                                  rs, re_, mbs, m, mh)
        assert len(res) == re_ - rs
        for i, (z, b) in enumerate(res):
            assert tuple(kt[p + 'bounds'][i]) == b, p
            np.testing.assert_array_equal(z, rag(kt, p + 'z', i))


def test_base_forward_pass_and_traceback(kt):
    import oracle
    for p, rs, re_, mbs, m, mh in region_cases(kt):
        bounds = kt[p + 'bounds']
        fwd, ld = kt[p + 'fp_first_fwd'], kt[p + 'fp_first_last_diag']
        rows = [fwd]
        for k, (bs, be, ps, pe, mm) in enumerate(kt[p + 'fp_args']):
            assert (bs, be) == tuple(bounds[k + 1]) and (ps, pe) == tuple(bounds[k]) and mm == m
            rc, nf, nl = oracle.base_forward_pass(
                rag(kt, p + 'z', k + 1), bs, be, rag(kt, p + 'z', k), ps, pe, fwd, ld, m)
            assert rc == 0
            np.testing.assert_array_equal(nf, rag(kt, p + 'fp_fwd', k))
            np.testing.assert_array_equal(nl, rag(kt, p + 'fp_last_diag', k))
            fwd, ld = nf, nl
            rows.append(nf)
        n = len(rows)
        for k, (cs, ns, ne, ss, mm, want) in enumerate(kt[p + 'tb_args']):
            b = n - 1 - k
            assert cs == bounds[b][0] and (ns, ne) == tuple(bounds[b - 1])
            got = oracle.base_traceback(rows[b], cs, rows[b - 1], ns, ne, ss, mm)
            assert got == want
        np.testing.assert_array_equal(kt[p + 'tb_args'][::-1, 5], kt[p + 'new_segs'])


def test_slopes_and_mean_stds(kt):
    import oracle
    np.testing.assert_array_equal(oracle.compute_slopes(kt['sl_ev'], kt['sl_md']), kt['sl_out'])
    np.testing.assert_array_equal(oracle.compute_slopes(kt['sl_ev'], kt['sl_md'], 5.0),
                                  kt['sl_out_max5'])
    assert (kt['sl_out'] == 1000.0).sum() == 2
    m, s = oracle.new_mean_stds(kt['ms_sig'], kt['ms_segs'])
    np.testing.assert_array_equal(m, kt['ms_means'])
    np.testing.assert_array_equal(s, kt['ms_stds'])


def test_llh_ratio_kernels(kt):
    """row N4: the per-position log-likelihood ratio kernels (_c_helper.pyx:277-358)"""
    import oracle
    kw = int(kt['llh_kw'][0])
    m, r, a = kt['llh_means'], kt['llh_ref_means'], kt['llh_alt_means']
    rv, av = kt['llh_ref_vars'], kt['llh_alt_vars']
    sf, hf, hp = (float(x) for x in kt['llh_scaled_params'])
    for i in range(kt['llh_var'].shape[0]):
        sl = slice(i, i + kw)
        assert oracle.calc_llh_ratio(m[sl], r[sl], a[sl], rv[sl], av[sl]) == kt['llh_var'][i]
        assert oracle.calc_llh_ratio_const_var(m[sl], r[sl], a[sl], rv[i]) == kt['llh_const'][i]
        assert oracle.calc_scaled_llh_ratio_const_var(m[sl], r[sl], a[sl], rv[i], sf, hf, hp) == \
            kt['llh_scaled'][i]


def test_resolve_skipped_bases_off_the_default_windows():
    """rq.resolve_skipped_bases_with_raw with non-default del_fix_window / max_del_fix_window /
    extra_sig_factor (resquiggle.py:405-407): results and errors recorded from the live reference
    (tests/golden/gen_golden_skipwin.py)"""
    import json
    import oracle
    from tombo_amd import errors, tombo_stats as ts, tombo_helper as th
    g = np.load(os.path.join(HERE, 'golden', 'kernels_skipwin.npz'))
    settings = json.loads(str(g['settings']))
    n_err = 0
    for name in g['names']:
        p = 'sw_%s_' % name
        params = ts.load_resquiggle_parameters(th.seqSampleType(str(g[p + 'samp']), False))
        for k, (dfw, mdfw, esf, mrc) in enumerate(settings):
            rc, out = oracle.resolve_skipped_bases(g[p + 'segs'], g[p + 'norm'], g[p + 'means'], g[p + 'sds'],
                                                   oracle.make_params(params), mrc, dfw, mdfw, esf)
            err = str(g[p + 'err%d' % k])
            if err:
                n_err += 1
                assert rc != 0 and errors.MESSAGES[rc] == err, (name, settings[k], rc, err)
            else:
                assert rc == 0, (name, settings[k], rc)
                np.testing.assert_array_equal(out, g[p + 'res%d' % k], err_msg='%s %r' % (name, settings[k]))
    assert n_err >= 10
