"""What the REFERENCE does with a read whose MAD is exactly 0 (build container only).

    python tests/golden/gen_golden_degenerate.py     # writes tests/golden/degenerate_cases.json

`ts.normalize_raw_signal` (tombo_stats.py:482-573) divides by the scale under
`np.seterr(all='raise')` (tombo_stats.py:19, resquiggle.py:29): a flat or saturated signal -- more
than half of the samples equal -- has a median absolute deviation of 0 and the division raises
FloatingPointError.  That is not a TomboError: the read dies with an "unexpected error", which the
oracle and the engine report as status 100 (ORC_INTERNAL / TBA_INTERNAL) whatever the batch size.
Only DATA is written: per case the generator arguments and the exception class + message the live
reference raised (or the scale it returned).
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


def make(case):
    """the signal of a case from its arguments (restated in tests/test_oracle_golden.py)"""
    rng = np.random.default_rng(case['seed'])
    x = rng.normal(90.0, 12.0, case['n'])
    if case['kind'] == 'flat':
        x[:] = case['value']
    elif case['kind'] == 'saturated':      # more than half of the samples sit on one value
        x[rng.permutation(case['n'])[:case['n'] // 2 + case['extra']]] = case['value']
    elif case['kind'] == 'half':           # exactly half: the MAD is the smallest other deviation / 2
        x[rng.permutation(case['n'])[:case['n'] // 2]] = case['value']
    if case['dtype'] == 'int16':
        x = np.round(x).astype(np.int16)
    return x


CASES = [dict(kind='flat', n=5000, value=431.0, seed=1, dtype='float64', extra=0),
         dict(kind='flat', n=5001, value=-3.0, seed=2, dtype='int16', extra=0),
         dict(kind='saturated', n=6000, value=120.0, seed=3, dtype='float64', extra=1),
         dict(kind='saturated', n=6001, value=120.0, seed=4, dtype='int16', extra=300),
         dict(kind='half', n=6000, value=120.0, seed=5, dtype='float64', extra=0),
         dict(kind='noise', n=4000, value=0.0, seed=6, dtype='float64', extra=0)]

out = []
for c in CASES:
    x = make(c)
    rec = dict(c)
    try:
        _, sv = ts.normalize_raw_signal(x, outlier_thresh=5.0)
        rec.update(raised=None, shift=float(sv.shift), scale=float(sv.scale))
    except Exception as e:
        rec.update(raised=type(e).__name__, message=str(e), is_tombo_error=isinstance(e, th.TomboError))
    out.append(rec)
    print(rec)
with open(os.path.join(HERE, 'degenerate_cases.json'), 'w') as fp:
    json.dump(out, fp, indent=1)
