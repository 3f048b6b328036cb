"""Row N4 on the GPU: the per-read statistics driver (tombo_amd.tombo_stats.compute_*_read_stats)
against vectors recorded from the live reference (tests/golden/gen_golden_stats.py), and the
resident-batch de novo statistic against the array form.

Tolerance: the p-values go through erfc / log / exp of the device library against scipy's
(cephes) norm.cdf / chi2.sf -- 1e-12 relative (observed ~1e-15); log-likelihood ratios: the
constant-variance form is bit-equal, the scaled form (exp, pow) 1e-12.  Positions are integers:
exact.  Error messages are the reference's strings."""
import os
import json

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _load():
    from tombo_amd import tombo_stats as ts, tombo_helper as th
    g = np.load(os.path.join(GOLDEN_DIR, 'stats_reads.npz'))
    meta = json.loads(str(g['meta']))
    model = ts.TomboModel(seq_samp_type=th.seqSampleType('DNA', False))
    alts = []
    for am in meta['alt_models']:
        rows = g[am['key']]
        alts.append((am['name'], ts.AltModel(
            [(r['kmer'], r['pos'], r['mean'], r['sd']) for r in rows], model.central_pos,
            am['alt_base'], name=am['name'], motif=th.TomboMotif(am['motif'], am['mod_pos']))))
    reads = []
    for ci, c in enumerate(meta['cases']):
        reads.append(th.resquiggledRead(
            start=c['start'], end=c['start'] + c['n'], filtered=False, read_start_rel_to_raw=0,
            strand=c['strand'], fn=c['fn'], corr_group='RawGenomeCorrected_000/BaseCalled_template',
            rna=False, read_id=c['read_id'], means=g['c%d_means' % ci], seq=str(g['c%d_seq' % ci])))
    return g, meta, model, alts, reads


class _Reg(object):
    def __init__(self, se):
        self.start, self.end = se


def _close(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ok = ~np.isnan(a)
    np.testing.assert_allclose(a[ok], b[ok], rtol=RTOL, atol=0)


def test_de_novo_and_sample_compare_match_the_reference():
    from tombo_amd import tombo_stats as ts, tombo_helper as th
    g, meta, model, alts, reads = _load()
    n_checked = 0
    for ci, (c, rd) in enumerate(zip(meta['cases'], reads)):
        for ri, reg in enumerate(c['regions']):
            regd = None if reg is None else _Reg(reg)
            for fm in meta['fm_offsets']:
                tag = 'c%d_r%d_fm%d' % (ci, ri, fm)
                err = str(g[tag + '_dn_err'])
                if err:
                    with pytest.raises(th.TomboError, match=err[:30]):
                        ts.compute_de_novo_read_stats(rd, model, fm, regd)
                else:
                    pv, ps, rid = ts.compute_de_novo_read_stats(rd, model, fm, regd)
                    _close(pv[ts.DE_NOVO_TXT], g[tag + '_dn_p'])
                    np.testing.assert_array_equal(ps[ts.DE_NOVO_TXT], g[tag + '_dn_pos'])
                    assert rid == c['read_id']
                err = str(g[tag + '_sc_err'])
                cm, cs = g[tag + '_sc_cm'], g[tag + '_sc_cs']
                if err:
                    with pytest.raises(th.TomboError, match=err[:30]):
                        ts.compute_sample_compare_read_stats(rd, cm, cs, fm, regd)
                else:
                    pv, ps, rid = ts.compute_sample_compare_read_stats(rd, cm, cs, fm, regd)
                    _close(pv[ts.SAMP_COMP_TXT], g[tag + '_sc_p'])
                    np.testing.assert_array_equal(ps[ts.SAMP_COMP_TXT], g[tag + '_sc_pos'])
                n_checked += 2
    assert n_checked >= 60


def test_alt_model_llhrs_match_the_reference():
    from tombo_amd import tombo_stats as ts, tombo_helper as th
    g, meta, model, alts, reads = _load()
    hits = 0
    for ci, (c, rd) in enumerate(zip(meta['cases'], reads)):
        for ri, reg in enumerate(c['regions']):
            regd = None if reg is None else _Reg(reg)
            for std_llhr in (False, True):
                tag = 'c%d_r%d_llhr%d' % (ci, ri, int(std_llhr))
                err = str(g[tag + '_am_err'])
                if err:
                    with pytest.raises(th.TomboError, match=err[:30]):
                        ts.compute_alt_model_read_stats(rd, model, alts, std_llhr, regd)
                    continue
                ll, ps, rid = ts.compute_alt_model_read_stats(rd, model, alts, std_llhr, regd)
                for name, _ in alts:
                    want = g[tag + '_am_%s_v' % name]
                    np.testing.assert_array_equal(ps[name], g[tag + '_am_%s_pos' % name])
                    if std_llhr:
                        np.testing.assert_array_equal(ll[name], want)   # constant variance: bit-equal
                    else:
                        np.testing.assert_allclose(ll[name], want, rtol=RTOL, atol=1e-300)
                    hits += want.shape[0]
    assert hits > 50


def test_batch_forms_equal_single_read_calls():
    from tombo_amd import tombo_stats as ts
    g, meta, model, alts, reads = _load()
    one = [ts.compute_de_novo_read_stats_batch([rd], model, 1)[0] for rd in reads]
    many = ts.compute_de_novo_read_stats_batch(reads, model, 1)
    for a, b in zip(one, many):
        assert isinstance(a, Exception) == isinstance(b, Exception)
        if not isinstance(a, Exception):
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
    am1 = [ts.compute_alt_model_read_stats_batch([rd], model, alts)[0] for rd in reads]
    amn = ts.compute_alt_model_read_stats_batch(reads, model, alts)
    for a, b in zip(am1, amn):
        assert isinstance(a, Exception) == isinstance(b, Exception)
        if not isinstance(a, Exception):
            for name, _ in alts:
                np.testing.assert_array_equal(a[0][name], b[0][name])


def test_resident_batch_de_novo_equals_the_array_form():
    """tba_batch_de_novo_stats on the finished resident batch (nothing uploaded) == the array form
    fed with the Events table of the same reads"""
    from tombo_amd import resquiggle as rq, synth, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    mrs = [synth.synth_map_res(model, nb, 4000 + nb, **synth.DNA_SYNTH) for nb in (400, 650, 20, 900)]
    res, tabs = rq.resquiggle_batch_events(mrs, model, params, outlier_thresh=5.0, seq_samp_type=samp)
    for fm in (0, 1, 3):
        got = rq.batch_de_novo_stats(fm_offset=fm)
        for i, r in enumerate(res):
            if isinstance(r, Exception):
                assert got[i] is None
                continue
            rd = th.read_from_results(r, tabs[i]['norm_mean'])
            try:
                want = ts.compute_de_novo_read_stats_batch([rd], model, fm)[0]
            except th.TomboError:
                want = None
            if isinstance(want, Exception) or want is None:
                assert got[i] is None or np.all(np.isnan(got[i][0]))
                continue
            np.testing.assert_array_equal(got[i][0], want[0])
            np.testing.assert_array_equal(got[i][1], want[1])
