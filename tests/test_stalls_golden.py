"""`ts.identify_stalls` (tombo/tombo_stats.py:269-368) off its default parameters, against
intervals recorded from the live reference (tests/golden/gen_golden_stalls.py ->
stalls_params.npz): eight parameter sets (window counts 2-16, full / no / negative widening, runs
of any length, nothing below the threshold) x seven signals (down to one sample below the window)
x {float64, int16 DAC}.  On CPU the numpy restatement in oracle/ is checked, on the GPU the device
kernels behind `tombo_amd.tombo_stats.identify_stalls` (`tba_identify_stalls`)."""
import os
import sys
import json
import hashlib

import numpy as np
import pytest

from conftest import GOLDEN_DIR


def _cases():
    from tombo_amd import tombo_helper as th
    g = np.load(os.path.join(GOLDEN_DIR, 'stalls_params.npz'))
    m = json.loads(str(g['meta']))
    from tombo_amd import synth
    for i, (n, n_st, scale) in enumerate(m['signals']):
        raw = synth.stalled_signal(np.random.default_rng(m['seed'] + i), n, n_st, scale)
        if i in (0, 2) and n > 2000:
            rng = np.random.default_rng(m['seed'] + 100 + i)
            raw[-900:] = raw[-900] + rng.normal(0, 2.0, 900)
            raw[:700] = raw[0] + rng.normal(0, 2.0, 700)
        assert hashlib.sha256(raw.tobytes()).hexdigest() == str(g['raw%d__sha' % i])
        dac = np.round(raw).astype(np.int16)
        for j, (ws, thr, mco, eb, nw, mw) in enumerate(m['params']):
            sp = th.stallParams(window_size=ws, threshold=thr, min_consecutive_obs=mco,
                                edge_buffer=eb, n_windows=nw, mini_window_size=mw)
            yield (i, j), sp, raw, dac, g['f64_s%d_p%d' % (i, j)], g['dac_s%d_p%d' % (i, j)]


def _ints(x):
    return np.array([[int(a), int(b)] for a, b in x], dtype=np.int64).reshape(-1, 2)


def test_restatement_reproduces_the_reference_intervals():
    import oracle
    n = n_iv = 0
    for key, sp, raw, dac, want_f, want_d in _cases():
        assert np.array_equal(_ints(oracle.identify_stalls(raw, sp)), want_f), key
        assert np.array_equal(_ints(oracle.identify_stalls(dac.astype(np.float64), sp)), want_d), key
        n += 2
        n_iv += want_f.shape[0] + want_d.shape[0]
    assert n == 112 and n_iv > 1000


@pytest.mark.gpu
def test_device_reproduces_the_reference_intervals():
    from tombo_amd import tombo_stats as ts
    for key, sp, raw, dac, want_f, want_d in _cases():
        assert np.array_equal(_ints(ts.identify_stalls(raw, sp)), want_f), key
        assert np.array_equal(_ints(ts.identify_stalls(dac, sp)), want_d), key   # int16 boundary
        assert np.array_equal(_ints(ts.identify_stalls(raw.astype(np.float32), sp)),
                              _ints(ts.identify_stalls(raw.astype(np.float32).astype(np.float64), sp))), key
