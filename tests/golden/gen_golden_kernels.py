"""Golden vectors for the stand-alone raw-signal DP / helper kernels, from the REFERENCE's
compiled Cython functions (build container only):

    python tests/golden/gen_golden_kernels.py      # writes tests/golden/kernels_tail.npz

c_reg_z_scores, c_base_forward_pass, c_base_traceback (_c_dynamic_programming.pyx:34-182),
c_compute_slopes, c_new_mean_stds (_c_helper.pyx:362-377, 38-57) and the three log-likelihood
ratio kernels (_c_helper.pyx:277-358) are called on seeded inputs;
inputs and outputs are stored as data (ragged lists as concatenation + offsets).
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import ref_oracle  # noqa: E402

rq, ts, th = ref_oracle.load()
out = {}


def ragged(key, arrs, dtype):
    out[key] = np.concatenate([np.asarray(a, dtype) for a in arrs]) if arrs else np.zeros(0, dtype)
    out[key + '_off'] = np.concatenate([[0], np.cumsum([len(a) for a in arrs])]).astype(np.int64)


# (name, n_bases, region, max_base_shift, min_obs_per_base, max_half_z_score, seed)
REG_CASES = [
    ('dna_window', 7, (0, 7), 7, 1, None, 1),
    ('dna_window_winsor', 9, (0, 9), 9, 1, 1.5, 2),
    ('rna_window', 6, (0, 6), 6, 2, 10.0, 3),
    ('inner_region_shift2', 12, (3, 10), 2, 3, 20.0, 4),
    ('two_bases', 2, (0, 2), 2, 1, None, 5),
]
names = []
for name, n, (rs, re_), mbs, m, mh, seed in REG_CASES:
    rng = np.random.default_rng(seed)
    dw = rng.integers(max(m, 1) * 2, 14, size=n)
    starts = np.concatenate([[0], np.cumsum(dw)]).astype(np.int64)
    if name.endswith('window') or name.startswith('dna_window') or name == 'two_bases':
        # what resolve_skipped_bases_with_raw passes: pseudo starts by linspace
        starts = np.linspace(0, starts[-1], n + 1).astype(np.int64)
    means, sds = rng.normal(0, 1, n), rng.uniform(0.1, 0.4, n)
    sig = rng.normal(0, 1, int(starts[-1]))
    res = rq.c_reg_z_scores(sig, means, sds, starts, rs, re_, mbs, m, max_half_z_score=mh)
    names.append(name)
    p = 'rz_%s_' % name
    out[p + 'sig'], out[p + 'means'], out[p + 'sds'], out[p + 'starts'] = sig, means, sds, starts
    out[p + 'args'] = np.array([rs, re_, mbs, m], np.int64)
    out[p + 'mh'] = np.array([np.nan if mh is None else mh])
    ragged(p + 'z', [r[0] for r in res], np.float64)
    out[p + 'bounds'] = np.array([r[1] for r in res], np.int64)

    # forward pass + traceback over the region, recording every kernel call
    fcalls, tcalls = [], []
    ofp, otb = rq.c_base_forward_pass, rq.c_base_traceback

    def rec_fp(*a):
        r = ofp(*a)
        fcalls.append((a, r))
        return r

    def rec_tb(*a):
        r = otb(*a)
        tcalls.append((a, r))
        return r
    rq.c_base_forward_pass, rq.c_base_traceback = rec_fp, rec_tb
    try:
        fwd = rq.raw_forward_pass(res, m)
        segs = rq.raw_traceback(fwd, m)
    finally:
        rq.c_base_forward_pass, rq.c_base_traceback = ofp, otb
    out[p + 'new_segs'] = np.asarray(segs, np.int64)
    ragged(p + 'fp_fwd', [r[0] for _, r in fcalls], np.float64)
    ragged(p + 'fp_last_diag', [r[1] for _, r in fcalls], np.int64)
    # scalar args of each call: forward (b_start, b_end, prev_b_start, prev_b_end, m);
    # traceback (curr_start, next_start, next_end, sig_start, m, result)
    out[p + 'fp_args'] = np.array([[a[1], a[2], a[4], a[5], a[8]] for a, _ in fcalls], np.int64)
    out[p + 'fp_first_fwd'] = np.asarray(fwd[0][0], np.float64)
    out[p + 'fp_first_last_diag'] = np.asarray(fwd[0][1], np.int64)
    out[p + 'tb_args'] = np.array([[a[1], a[3], a[4], a[5], a[6], -1 if r is None else r]
                                   for a, r in tcalls], np.int64).reshape(-1, 6)
out['rz_names'] = np.array(names)

# c_compute_slopes: ties in the event means take max_slope
rng = np.random.default_rng(10)
ev, md = rng.normal(0, 1, 41), rng.normal(0, 1, 41)
ev[7] = ev[3]
ev[40] = ev[0]
out['sl_ev'], out['sl_md'] = ev, md
out['sl_out'] = ts.c_compute_slopes(ev, md)
out['sl_out_max5'] = ts.c_compute_slopes(ev, md, 5.0)

# c_new_mean_stds
sig = rng.normal(0, 1, 5000)
segs = np.concatenate([[0], np.sort(rng.choice(np.arange(1, 5000), 300, replace=False)), [5000]])
segs = segs.astype(np.int64)
m, s = th.c_new_mean_stds(sig, segs)
out['ms_sig'], out['ms_segs'], out['ms_means'], out['ms_stds'] = sig, segs, m, s

# c_calc_llh_ratio, c_calc_llh_ratio_const_var, c_calc_scaled_llh_ratio_const_var
# (_c_helper.pyx:277-358): per-position log-likelihood ratio tests of the model-comparison
# statistics (tombo_stats.py:4059-4074), regions of one k-mer width; OCLLHR_* as in
# _default_parameters.py:132-134
rng = np.random.default_rng(31)
n_reg, kw = 400, 6
L = n_reg + kw - 1
lm, lr, la = rng.normal(0, 1, L), rng.normal(0, 1, L), rng.normal(0, 1, L)
la[50:60] = lr[50:60]                      # equal reference / alternative levels are skipped
lrv, lav = rng.uniform(0.01, 0.3, L), rng.uniform(0.01, 0.3, L)
out['llh_means'], out['llh_ref_means'], out['llh_alt_means'] = lm, lr, la
out['llh_ref_vars'], out['llh_alt_vars'] = lrv, lav
out['llh_kw'] = np.array([kw], np.int64)
out['llh_scaled_params'] = np.array([4.0, 1.0, 0.2])
out['llh_var'] = np.array([ts.c_calc_llh_ratio(lm[i:i + kw], lr[i:i + kw], la[i:i + kw],
                                               lrv[i:i + kw], lav[i:i + kw]) for i in range(n_reg)])
out['llh_const'] = np.array([ts.c_calc_llh_ratio_const_var(lm[i:i + kw], lr[i:i + kw], la[i:i + kw],
                                                           lrv[i]) for i in range(n_reg)])
out['llh_scaled'] = np.array([ts.c_calc_scaled_llh_ratio_const_var(
    lm[i:i + kw], lr[i:i + kw], la[i:i + kw], lrv[i], 4.0, 1.0, 0.2) for i in range(n_reg)])

np.savez_compressed(os.path.join(HERE, 'kernels_tail.npz'), **out)
print('wrote kernels_tail.npz: %d arrays, %.1f KB' % (
    len(out), os.path.getsize(os.path.join(HERE, 'kernels_tail.npz')) / 1024.))
