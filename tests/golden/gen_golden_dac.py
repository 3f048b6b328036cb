"""Record what the REFERENCE does on int16-DAC-quantised reads (build container only).

    python tests/golden/gen_golden_dac.py        # writes tests/golden/dacq_*.npz

Event detection ranks change-point scores with `np.argsort(...)[::-1]`
(tombo/_c_helper.pyx:95-98); on quantised signal most scores tie exactly and the order inside a
tie is whatever numpy's unstable sort produces -- which depends on numpy's CPU dispatch
(AVX512 / AVX2 / scalar sorts give three different orders on the same array).  This script runs
`tombo.resquiggle.resquiggle_read` of the live reference on DAC-rounded versions of the cfg1 /
cfg2 / cfg4 reads once per dispatch (a subprocess per dispatch, NPY_DISABLE_CPU_FEATURES) and
stores `valid_cpts`, `segs`, `read_start_rel_to_raw` of each run.  Only data is written.  The
tests (tests/test_dac_quantised.py) measure how far the oracle / the engine (one fixed tie rule:
score descending, index descending) are from each recorded run, next to how far the reference's
own runs are from each other.
"""
import os
import sys
import json
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

DISPATCH = {
    'avx512': '',
    'avx2': 'AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR',
    'scalar': 'AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3',
}

CASES = [
    dict(name='dacq_dna_b2000_w100', samp_name='DNA', n_bases=2000, seeds=[0, 1, 2], bandwidth=100,
         band_bound_thresh=10),
    dict(name='dacq_dna_b10000_w500', samp_name='DNA', n_bases=10000, seeds=[100, 101],
         bandwidth=500),
    dict(name='dacq_rna_b3000_w500', samp_name='RNA', n_bases=3000, seeds=[21, 22]),
]


def to_dac(raw):
    """MinION-like digitisation (range 1400 pA over 8192 levels, offset 10): int16 DAC values"""
    return np.round(raw / 0.1709 + 10.0).astype(np.int16)


def worker(case_json):
    import ref_oracle
    from tombo_amd import synth, tombo_stats as my_ts, tombo_helper as my_th
    rq, ts, th = ref_oracle.load()
    c = json.loads(case_json)
    samp = th.seqSampleType(c['samp_name'], False)
    my_model = my_ts.TomboModel(seq_samp_type=my_th.seqSampleType(c['samp_name'], False))
    kmers = sorted(my_model.means.keys())
    std_ref = ts.TomboModel(kmer_ref=[(k, my_model.means[k], my_model.sds[k]) for k in kmers],
                            central_pos=my_model.central_pos, seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    if c.get('bandwidth'):
        params = params._replace(bandwidth=c['bandwidth'])
    if c.get('band_bound_thresh'):
        params = params._replace(band_bound_thresh=c['band_bound_thresh'])
    kw = dict(synth.DNA_SYNTH if c['samp_name'] == 'DNA' else synth.RNA_SYNTH)
    out = {}
    grabbed = {}
    orig = rq.segment_signal

    def seg(*a, **k):
        r = orig(*a, **k)
        grabbed['cpts'] = r[0].astype(np.int64)
        return r
    rq.segment_signal = seg
    for seed in c['seeds']:
        seq, raw, _ = synth.synth_read(my_model, c['n_bases'], seed, **kw)
        dac = to_dac(raw)
        stall = None
        if c['samp_name'] == 'RNA':
            stall = ts.identify_stalls(dac.astype(np.float64), rq.DEFAULT_STALL_PARAMS)
            out['s%d_stall_ints' % seed] = np.array(
                [[int(a), int(b)] for a, b in stall], dtype=np.int64).reshape(-1, 2)
        mr = th.resquiggleResults(
            align_info=th.alignInfo('r', 'BaseCalled_template', 0, 0, 0, 0, c['n_bases'], 0),
            genome_loc=th.genomeLocation(0, '+', 'synth'), genome_seq=seq, mean_q_score=10.0,
            raw_signal=dac, stall_ints=stall)
        np.random.seed(seed)
        try:
            res = rq.resquiggle_read(mr, std_ref, params, 5.0, seq_samp_type=samp)
            err = ''
        except th.TomboError as e:
            res, err = None, str(e)
        out['s%d_error' % seed] = np.array(err)
        out['s%d_valid_cpts' % seed] = grabbed.get('cpts', np.zeros(0, np.int64)).astype(np.int32)
        if res is not None:
            out['s%d_segs' % seed] = res.segs.astype(np.int32)
            out['s%d_read_start' % seed] = np.int64(res.read_start_rel_to_raw)
            sv = res.scale_values
            out['s%d_scale_values' % seed] = np.array([sv.shift, sv.scale, sv.lower_lim, sv.upper_lim])
            out['s%d_score' % seed] = np.float64(res.sig_match_score)
    np.savez_compressed(c['tmp'], **out)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == '--worker':
        worker(sys.argv[2])
        return
    for c in CASES:
        merged = {}
        for disp, disable in DISPATCH.items():
            tmp = '/tmp/dacq_%s_%s.npz' % (c['name'], disp)
            env = dict(os.environ)
            if disable:
                env['NPY_DISABLE_CPU_FEATURES'] = disable
            cj = dict(c, tmp=tmp)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), '--worker',
                                   json.dumps(cj)], env=env)
            d = np.load(tmp)
            for k in d.files:
                merged['%s__%s' % (disp, k)] = d[k]
        merged['meta'] = np.array(json.dumps(dict(
            c, dispatch=list(DISPATCH), dac='round(pA / 0.1709 + 10) as int16', np_seed='seed',
            outlier_thresh=5.0)))
        path = os.path.join(HERE, c['name'] + '.npz')
        np.savez_compressed(path, **merged)
        print(c['name'], '%.1f KB' % (os.path.getsize(path) / 1024.))
        for seed in c['seeds']:
            a = merged['avx512__s%d_segs' % seed]
            for disp in ('avx2', 'scalar'):
                b = merged['%s__s%d_segs' % (disp, seed)]
                ca, cb = merged['avx512__s%d_valid_cpts' % seed], merged['%s__s%d_valid_cpts' % (disp, seed)]
                print('  seed %d: reference avx512 vs %s: cpts identical %.4f, segs identical %.4f, '
                      'max shift %d' % (seed, disp, np.isin(ca, cb).mean(),
                                        (a == b).mean() if a.shape == b.shape else -1,
                                        np.abs(a - b).max() if a.shape == b.shape else -1))


if __name__ == '__main__':
    main()
