"""Generate golden vectors from the REFERENCE implementation (build container only).

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz

Imports the reference through tools/ref_oracle.py (scratch build outside the repo), runs
`tombo.resquiggle.resquiggle_read` on deterministic synthetic reads (tombo_amd/synth.py) and
records stage-wise intermediates by wrapping the reference's own functions.  Only DATA is
written: integer arrays in full, float arrays in full for the small cases and as
(sha256, head, tail, strided sample) for the large ones.  The raw input is regenerated from
the seed by the tests and checked against the recorded sha256.
"""
import os
import sys
import json
import hashlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import ref_oracle  # noqa: E402
from tombo_amd import synth, tombo_stats as my_ts, tombo_helper as my_th  # noqa: E402
import oracle  # noqa: E402  (the restatement being pinned)

rq, ts, th = ref_oracle.load()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


LITE = [False]   # (run_case(lite=True): float arrays as SHA-256 + length only)


def fsummary(name, a, out, full):
    """float array -> fixture entries"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    out[name + '__sha'] = sha(a)
    out[name + '__len'] = np.int64(a.shape[0] if a.ndim else 1)
    if LITE[0]:
        return
    if full or a.size <= 4096:
        out[name] = a
    else:
        out[name + '__head'] = a[:64].copy()
        out[name + '__tail'] = a[-64:].copy()
        out[name + '__stride'] = a[::max(1, a.size // 512)].copy()


def ref_model(my_model, samp):
    kmers = sorted(my_model.means.keys())
    return ts.TomboModel(kmer_ref=[(k, my_model.means[k], my_model.sds[k]) for k in kmers],
                         central_pos=my_model.central_pos, seq_samp_type=samp)


class Capture(object):
    """Wraps reference functions to record intermediates of one resquiggle_read call."""

    def __init__(self):
        self.d = {}
        self.calls = {}

    def wrap(self, mod, name, post):
        orig = getattr(mod, name)

        def f(*a, **k):
            r = orig(*a, **k)
            n = self.calls.get(name, 0)
            self.calls[name] = n + 1
            post(self.d, n, a, k, r)
            return r
        setattr(mod, name, f)
        return orig


def run_case(name, samp_name, n_bases, seed, bandwidth=None, band_bound_thresh=None,
             synth_kw=None, full=False, outlier_thresh=5.0, skip_seq_scaling=False,
             const_scale=None, second_iter=False, noise_body=False, sig_aln_params=None,
             seg_params=None, max_raw_cpts=None, edit=None, lite=False):
    """sig_aln_params / seg_params: --signal-align-parameters / --segmentation-parameters style
    overrides (tombo/_option_parsers.py:375-385,606-617 -> load_resquiggle_parameters)"""
    LITE[0] = bool(lite)
    samp = th.seqSampleType(samp_name, False)
    my_samp = my_th.seqSampleType(samp_name, False)
    my_model = my_ts.TomboModel(seq_samp_type=my_samp)
    std_ref = ref_model(my_model, samp)
    params = ts.load_resquiggle_parameters(samp, sig_aln_params, seg_params)
    if bandwidth is not None:
        params = params._replace(bandwidth=bandwidth)
    if band_bound_thresh is not None:
        params = params._replace(band_bound_thresh=band_bound_thresh)
    kw = dict(synth.DNA_SYNTH if samp_name == 'DNA' else synth.RNA_SYNTH)
    kw.update(synth_kw or {})
    seq, raw, true_starts = synth.synth_read(my_model, n_bases, seed, **kw)
    seq, raw = synth.edit_read(seq, raw, true_starts, edit)
    if noise_body:
        rng = np.random.default_rng(seed + 12345)
        raw = rng.normal(0.0, 1.0, size=raw.shape[0]) * kw['scale'] + kw['offset']
    stall_ints = None
    if samp_name == 'RNA':
        stall_ints = ts.identify_stalls(raw, rq.DEFAULT_STALL_PARAMS)
        mine = oracle.identify_stalls(raw)
        assert len(mine) == len(stall_ints) and all(
            int(a[0]) == int(b[0]) and int(a[1]) == int(b[1])
            for a, b in zip(mine, stall_ints)), 'identify_stalls restatement differs'
    map_res = th.resquiggleResults(
        align_info=th.alignInfo('r', 'BaseCalled_template', 0, 0, 0, 0, n_bases, 0),
        genome_loc=th.genomeLocation(0, '+', 'synth'), genome_seq=seq, mean_q_score=10.0,
        raw_signal=raw, stall_ints=stall_ints)

    out = {}
    meta = dict(name=name, samp=samp_name, n_bases=n_bases, seed=seed,
                bandwidth=int(params.bandwidth), band_bound_thresh=int(params.band_bound_thresh),
                synth_kw=kw, outlier_thresh=outlier_thresh, skip_seq_scaling=skip_seq_scaling,
                const_scale=const_scale, noise_body=noise_body, second_iter=second_iter,
                np_seed=seed, sig_aln_params=sig_aln_params, seg_params=seg_params,
                max_raw_cpts=max_raw_cpts, edit=edit)
    out['raw__sha'] = sha(raw)
    out['raw__len'] = np.int64(raw.shape[0])
    if stall_ints is not None:
        out['stall_ints'] = np.array([[int(a), int(b)] for a, b in stall_ints],
                                     dtype=np.int64).reshape(-1, 2)

    cap = Capture()
    origs = []

    def post_seg(d, n, a, k, r):
        d['valid_cpts'] = r[0].astype(np.int64)
        fsummary('seg_norm_signal', r[1], d, full)
        sv = r[2]
        d['seg_scale_values'] = np.array(
            [sv.shift, sv.scale, np.nan if sv.lower_lim is None else sv.lower_lim,
             np.nan if sv.upper_lim is None else sv.upper_lim], dtype=np.float64)
    origs.append((rq, 'segment_signal', cap.wrap(rq, 'segment_signal', post_seg)))

    def post_means(d, n, a, k, r):
        fsummary('base_means_call%d' % n, r, d, full)
    origs.append((ts, 'compute_base_means', cap.wrap(ts, 'compute_base_means', post_means)))

    def post_start(d, n, a, k, r):
        d['start_call%d' % n] = np.array([float(r[0]), float(r[1])])
        d['start_call%d_bw' % n] = np.int64(a[5])
    origs.append((rq, 'find_seq_start_in_events',
                  cap.wrap(rq, 'find_seq_start_in_events', post_start)))

    def post_mask(d, n, a, k, r):
        d['mask_band_event_starts'] = r[2].astype(np.int64)
        d['mask_z__sha'] = sha(r[3])
        d['mask_fwd_last'] = r[0][-1].copy()
        d['mask_args'] = np.array([float(a[3]), float(a[5])])  # mapped_start_offset, epb
    origs.append((rq, '_get_masked_start_fwd_pass',
                  cap.wrap(rq, '_get_masked_start_fwd_pass', post_mask)))

    def post_adapt(d, n, a, k, r):
        d['band_event_starts'] = a[2].astype(np.int64).copy()
        d['fwd_last_row'] = a[0][-1].copy()
        d['fwd_pass__sha'] = sha(a[0])
        d['adapt_start_seq_pos'] = np.int64(a[9] if len(a) > 9 else k['start_seq_pos'])
    origs.append((th, 'adaptive_banded_forward_pass',
                  cap.wrap(th, 'adaptive_banded_forward_pass', post_adapt)))

    def post_tb(d, n, a, k, r):
        d['traceback_call%d' % n] = r.astype(np.int64).copy()
    origs.append((th, 'banded_traceback', cap.wrap(th, 'banded_traceback', post_tb)))

    def post_static(d, n, a, k, r):
        d['static_read_tb'] = r.astype(np.int64).copy()
    origs.append((rq, 'find_static_base_assignment',
                  cap.wrap(rq, 'find_static_base_assignment', post_static)))

    def post_dp(d, n, a, k, r):
        d['dp_segs'] = r.segs.astype(np.int64)
        d['dp_read_start_rel_to_raw'] = np.int64(r.read_start_rel_to_raw)
    origs.append((rq, 'find_adaptive_base_assignment',
                  cap.wrap(rq, 'find_adaptive_base_assignment', post_dp)))

    def post_ts(d, n, a, k, r):
        d['theil_sen'] = np.array([float(v) for v in r])
    origs.append((ts, 'calc_kmer_fitted_shift_scale',
                  cap.wrap(ts, 'calc_kmer_fitted_shift_scale', post_ts)))

    # exact ties among the change-point scores: the reference ranks them with np.argsort
    # (_c_helper.pyx:95-98, 176-178), whose order inside a tie is numpy's (unstable, CPU-dispatch
    # dependent) -- a fixture whose picks hang on such a tie pins nothing.  The score array is
    # taken from the reference's own argsort call.
    seen_scores = []
    real_argsort = np.argsort

    def spy_argsort(a, *args, **kw):
        if not seen_scores:
            seen_scores.append(np.array(a, dtype=np.float64, copy=True))
        return real_argsort(a, *args, **kw)

    def do_call(mr, **kws):
        np.random.seed(seed)
        if max_raw_cpts is not None:
            kws['max_raw_cpts'] = max_raw_cpts
        return rq.resquiggle_read(mr, std_ref, params, outlier_thresh, seq_samp_type=samp, **kws)

    try:
        try:
            np.argsort = spy_argsort
            try:
                res = do_call(map_res, const_scale=const_scale, skip_seq_scaling=skip_seq_scaling)
            finally:
                np.argsort = real_argsort
            err = ''
        except th.TomboError as e:
            res, err = None, str(e)
        out.update(cap.d)
        if seen_scores and 'valid_cpts' in cap.d and len(cap.d['valid_cpts']):
            sc, w = seen_scores[0], int(params.running_stat_width)
            idx = cap.d['valid_cpts'] - w
            idx = idx[(idx >= 0) & (idx < sc.shape[0])]
            if idx.size:
                top = sc[sc >= sc[idx].min()]
                _, cnt = np.unique(top, return_counts=True)
                meta['score_ties'] = int(cnt[cnt > 1].sum())
        out['error'] = np.array(err)
        if res is not None:
            out['segs'] = res.segs.astype(np.int64)
            out['read_start_rel_to_raw'] = np.int64(res.read_start_rel_to_raw)
            fsummary('norm_signal', res.raw_signal, out, full)
            sv = res.scale_values
            out['scale_values'] = np.array([np.nan if v is None else v for v in (
                sv.shift, sv.scale, sv.lower_lim, sv.upper_lim)], dtype=np.float64)
            out['sig_match_score'] = np.float64(res.sig_match_score)
            out['norm_params_changed'] = np.bool_(res.norm_params_changed)
            meta['median_abs_boundary_err'] = -1.0 if edit else float(np.median(np.abs(
                res.read_start_rel_to_raw + res.segs - true_starts)))
            if second_iter:
                cap.d = {}
                cap.calls = {}
                try:
                    res2 = do_call(map_res._replace(scale_values=res.scale_values),
                                   all_raw_signal=raw)
                    err2 = ''
                except th.TomboError as e:
                    res2, err2 = None, str(e)
                out['it2_error'] = np.array(err2)
                if res2 is not None:
                    out['it2_valid_cpts'] = cap.d['valid_cpts']
                    out['it2_segs'] = res2.segs.astype(np.int64)
                    out['it2_read_start_rel_to_raw'] = np.int64(res2.read_start_rel_to_raw)
                    fsummary('it2_norm_signal', res2.raw_signal, out, full)
                    sv = res2.scale_values
                    out['it2_scale_values'] = np.array(
                        [np.nan if v is None else v for v in (
                            sv.shift, sv.scale, sv.lower_lim, sv.upper_lim)], dtype=np.float64)
                    out['it2_sig_match_score'] = np.float64(res2.sig_match_score)
                    out['it2_norm_params_changed'] = np.bool_(res2.norm_params_changed)
    finally:
        for mod, nm, o in origs:
            setattr(mod, nm, o)
    out['meta'] = np.array(json.dumps(meta))
    # compact integer arrays
    for k in list(out.keys()):
        v = out[k]
        if isinstance(v, np.ndarray) and v.dtype == np.int64 and v.size > 16 and \
                np.abs(v).max() < 2 ** 31:
            out[k] = v.astype(np.int32)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    sz = os.path.getsize(os.path.join(HERE, name + '.npz'))
    print('%-28s err=%-60r %6.1f KB  %s' % (
        name, err, sz / 1024., '' if res is None else
        'score=%.4f changed=%s start=%d bnd_err=%.1f' % (
            res.sig_match_score, res.norm_params_changed, res.read_start_rel_to_raw,
            meta['median_abs_boundary_err'])))


CASES = [
    dict(name='dna_b600_w300', samp_name='DNA', n_bases=600, seed=11, full=True,
         second_iter=True),
    dict(name='dna_b2000_w100_s0', samp_name='DNA', n_bases=2000, seed=0, bandwidth=100,
         band_bound_thresh=10),
    dict(name='dna_b2000_w100_s1', samp_name='DNA', n_bases=2000, seed=1, bandwidth=100,
         band_bound_thresh=10),
    dict(name='dna_b2000_w300', samp_name='DNA', n_bases=2000, seed=2, second_iter=True),
    dict(name='dna_b2000_w500', samp_name='DNA', n_bases=2000, seed=3, bandwidth=500),
    dict(name='dna_b10000_w500', samp_name='DNA', n_bases=10000, seed=100, bandwidth=500),
    dict(name='dna_b150_static', samp_name='DNA', n_bases=150, seed=4, full=True),
    dict(name='dna_b300_static', samp_name='DNA', n_bases=300, seed=5, full=True),
    dict(name='dna_leader5000_retry', samp_name='DNA', n_bases=1500, seed=6,
         synth_kw=dict(lead=5000)),
    dict(name='dna_noise_body', samp_name='DNA', n_bases=1000, seed=7, noise_body=True),
    dict(name='dna_dwell400_bandfail', samp_name='DNA', n_bases=400, seed=8,
         synth_kw=dict(mean_dwell=400)),
    dict(name='dna_b20_too_much_signal', samp_name='DNA', n_bases=20, seed=9,
         synth_kw=dict(lead=60000)),
    dict(name='dna_b2000_w100_bbt40_fail', samp_name='DNA', n_bases=2000, seed=0, bandwidth=100),
    dict(name='dna_skip_seq_scaling', samp_name='DNA', n_bases=800, seed=12,
         skip_seq_scaling=True),
    dict(name='dna_const_scale', samp_name='DNA', n_bases=800, seed=13, const_scale=12.0),
    dict(name='dna_no_outlier', samp_name='DNA', n_bases=800, seed=14, outlier_thresh=None),
    dict(name='rna_b600_w500', samp_name='RNA', n_bases=600, seed=20, second_iter=True),
    dict(name='rna_b3000_w500', samp_name='RNA', n_bases=3000, seed=21),
    # ---- off the defaults (round 3): --segmentation-parameters / --signal-align-parameters ----
    # (running_stat_width, min_obs_per_base, raw_min_obs_per_base, mean_obs_per_event)
    dict(name='p_dna_seg_7_4_2_6', samp_name='DNA', n_bases=1500, seed=30, seg_params=(7, 4, 2, 6),
         second_iter=True),                                  # generic k_peaks radius, raw m = 2 on DNA
    dict(name='p_dna_seg_w33', samp_name='DNA', n_bases=1200, seed=31, seg_params=(33, 3, 1, 5)),  # 2w > 64
    dict(name='p_dna_seg_3_2_1_4', samp_name='DNA', n_bases=1200, seed=32, seg_params=(3, 2, 1, 4)),
    dict(name='p_rna_seg_10_5_3_12', samp_name='RNA', n_bases=700, seed=33, seg_params=(10, 5, 3, 12)),
    # (match_evalue, skip_pen, bandwidth, save_bandwidth, max_half_z_score, band_bound_thresh,
    #  start_bw, start_save_bw, start_n_bases)
    dict(name='p_dna_aln_me35_sp5', samp_name='DNA', n_bases=1500, seed=34,
         sig_aln_params=(3.5, 5.0, 300, 1500, 20.0, 40, 750, 2500, 250)),
    dict(name='p_dna_aln_bw700_z10_bbt30', samp_name='DNA', n_bases=2500, seed=35,
         sig_aln_params=(4.2, 4.2, 700, 2000, 10.0, 30, 750, 2500, 250)),
    dict(name='p_dna_aln_start500_150', samp_name='DNA', n_bases=1500, seed=36,
         sig_aln_params=(4.2, 4.2, 300, 1500, 20.0, 40, 500, 1800, 150)),
    dict(name='p_dna_aln_start_retry', samp_name='DNA', n_bases=1500, seed=37, synth_kw=dict(lead=3500),
         sig_aln_params=(4.2, 4.2, 300, 1500, 20.0, 40, 500, 1800, 150)),
    dict(name='p_dna_aln_z3', samp_name='DNA', n_bases=1200, seed=38,
         sig_aln_params=(4.2, 4.2, 300, 1500, 3.0, 40, 750, 2500, 250)),
    dict(name='p_rna_aln_me5_sp3_bw300', samp_name='RNA', n_bases=700, seed=39,
         sig_aln_params=(5.0, 3.0, 300, 1500, 15.0, 40, 800, 2500, 200)),
    dict(name='p_dna_outlier3', samp_name='DNA', n_bases=1200, seed=40, outlier_thresh=3.0, second_iter=True),
    dict(name='p_rna_outlier3', samp_name='RNA', n_bases=700, seed=41, outlier_thresh=3.0),
    # (RNA with outlier_thresh=None is a TypeError in the reference -- get_scale_values_from_events,
    # tombo_stats.py:228 -- i.e. an "unexpected error": TBA_INTERNAL here, tests/test_gpu_parity.py)
    dict(name='p_dna_max_raw_cpts_4', samp_name='DNA', n_bases=1500, seed=43, max_raw_cpts=4),
    # ---- sequence and signal that disagree; the remaining error reachable through resquiggle_read ----
    dict(name='e_dna_dwell2_fewer_cpts', samp_name='DNA', n_bases=1500, seed=51,
         synth_kw=dict(mean_dwell=2, min_dwell=1)),          # int(1.1 B) events asked of ~3 B samples
    dict(name='e_dna_seg_fewer_cpts', samp_name='DNA', n_bases=1200, seed=52, seg_params=(5, 6, 2, 4)),
    dict(name='e_rna_seg_fewer_cpts', samp_name='RNA', n_bases=500, seed=62, seg_params=(12, 9, 3, 8)),
    dict(name='e_dna_cut30', samp_name='DNA', n_bases=1500, seed=54, edit=dict(kind='cut', n=30)),
    dict(name='e_dna_cut60_bandfail', samp_name='DNA', n_bases=1500, seed=54, edit=dict(kind='cut', n=60)),
    dict(name='e_dna_insert40', samp_name='DNA', n_bases=1500, seed=55, edit=dict(kind='insert', n=40, seed=5)),
    dict(name='e_dna_trunc95', samp_name='DNA', n_bases=2000, seed=50, edit=dict(kind='truncate', frac=0.95)),
    dict(name='e_dna_trunc85_bandfail', samp_name='DNA', n_bases=2000, seed=50,
         edit=dict(kind='truncate', frac=0.85)),
    dict(name='e_dna_lead12000_static', samp_name='DNA', n_bases=400, seed=59, synth_kw=dict(lead=12000)),
    dict(name='e_rna_dwell8', samp_name='RNA', n_bases=800, seed=61, synth_kw=dict(mean_dwell=8, min_dwell=2)),
    dict(name='e_rna_trunc70_bandfail', samp_name='RNA', n_bases=800, seed=60,
         edit=dict(kind='truncate', frac=0.7)),
]

if __name__ == '__main__':
    only = sys.argv[1:]
    for c in CASES:
        if only and c['name'] not in only:
            continue
        run_case(**c)
