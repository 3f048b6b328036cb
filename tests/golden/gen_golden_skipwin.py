"""Golden vectors for rq.resolve_skipped_bases_with_raw (tombo/resquiggle.py:402-540) OFF its default
window constants, from the live REFERENCE (build container only):

    python tests/golden/gen_golden_skipwin.py      # writes tests/golden/kernels_skipwin.npz

The function's keyword arguments del_fix_window / max_del_fix_window / extra_sig_factor
(_default_parameters.py:67,72,73) were compile-time constants of the engine and of the oracle until
round 4; these vectors pin both once they became parameters.  Inputs are synthetic dpResults (base
boundaries with skipped bases -- single, in runs, close together, at both ends --, expected levels,
a normalised signal); per input a list of settings, and for each the function's result or the
message of the TomboError it raised.  Stored as data.
"""
import os
import sys
import json
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import ref_oracle  # noqa: E402

rq, ts, th = ref_oracle.load()

# (del_fix_window, max_del_fix_window, extra_sig_factor, max_raw_cpts)
SETTINGS = [(2, 10, 1.1, 200), (1, 4, 1.1, 200), (3, 6, 1.5, 200), (0, 3, 2.0, 200), (2, 2, 1.1, 200),
            (4, 12, 1.0, 200), (2, 10, 3.0, 200), (1, 2, 4.0, 200), (5, 20, 1.1, 12), (2, 10, 1.1, None)]
# (name, sample type, bases, seed, skipped positions)
CASES = [
    ('dna_single', 'DNA', 80, 1, [30]),
    ('dna_runs', 'DNA', 120, 2, [10, 11, 12, 40, 44, 47, 90, 91]),
    ('dna_ends', 'DNA', 60, 3, [0, 1, 57, 58]),
    ('dna_tight', 'DNA', 90, 4, [20, 22, 24, 26, 28, 30, 60]),
    ('dna_short_signal', 'DNA', 70, 5, [15, 16, 17, 18, 35]),
    ('rna_runs', 'RNA', 100, 6, [5, 30, 31, 32, 33, 70]),
    ('rna_ends', 'RNA', 50, 7, [0, 48]),
    ('dna_none', 'DNA', 40, 8, []),
]
out = {'names': np.array([c[0] for c in CASES]), 'settings': json.dumps(SETTINGS)}
for name, samp_name, nb, seed, skipped in CASES:
    rng = np.random.default_rng(seed)
    samp = th.seqSampleType(samp_name, samp_name == 'RNA')
    params = ts.load_resquiggle_parameters(samp)
    lo = 2 if name == 'dna_short_signal' else (3 if samp_name == 'DNA' else 6)
    dwell = rng.integers(lo, lo + (2 if name == 'dna_short_signal' else 12), size=nb)
    dwell[skipped] = 0
    segs = np.concatenate([[0], np.cumsum(dwell)]).astype(np.int64)
    means, sds = rng.normal(0, 1, nb), rng.uniform(0.1, 0.4, nb)
    norm = np.repeat(means, dwell) + rng.normal(0, 0.25, int(segs[-1]))
    dp = th.dpResults(read_start_rel_to_raw=0, segs=segs, ref_means=means, ref_sds=sds, genome_seq='A' * nb)
    p = 'sw_%s_' % name
    out[p + 'segs'], out[p + 'means'], out[p + 'sds'], out[p + 'norm'] = segs, means, sds, norm
    out[p + 'samp'] = np.array(samp_name)
    for k, (dfw, mdfw, esf, mrc) in enumerate(SETTINGS):
        try:
            res = rq.resolve_skipped_bases_with_raw(dp, norm, params, mrc, dfw, mdfw, esf)
            out[p + 'res%d' % k], out[p + 'err%d' % k] = np.asarray(res, np.int64), np.array('')
        except th.TomboError as e:
            out[p + 'res%d' % k], out[p + 'err%d' % k] = np.zeros(0, np.int64), np.array(str(e))
        print(name, (dfw, mdfw, esf, mrc), str(out[p + 'err%d' % k]) or 'ok, %d boundaries moved' % int(
            (out[p + 'res%d' % k] != segs).sum()))
np.savez_compressed(os.path.join(HERE, 'kernels_skipwin.npz'), **out)
