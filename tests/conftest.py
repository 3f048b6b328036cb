import os
import sys
import json
import glob
import hashlib

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def golden_names(oracle_only=False):
    # read-level cases; kernels_*.npz hold stand-alone kernel vectors (test_oracle_kernels_golden),
    # dacq_*.npz the reference's runs on DAC-quantised reads (test_dac_quantised), stalls_*.npz
    # identify_stalls off its defaults (test_stalls_golden), loop_*.npz the worker loop.
    # o_*.npz (gen_golden_box.py) pin the oracle over the parameter box of the GPU fuzz and are
    # only listed with oracle_only=True: the engine meets that box through hypothesis.
    skip = ('kernels_', 'dacq_', 'stats_', 'loop_', 'stalls_') + (() if oracle_only else ('o_',))
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN_DIR, '*.npz'))
                  if not os.path.basename(f).startswith(skip))


class GoldenCase(object):
    """A golden fixture + the regenerated inputs it was produced from."""

    def __init__(self, name):
        from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
        self.name = name
        self.g = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        self.meta = json.loads(str(self.g['meta']))
        m = self.meta
        self.samp = th.seqSampleType(m['samp'], False)
        self.model = ts.TomboModel(seq_samp_type=self.samp)
        # (round-3 fixtures carry --signal-align-parameters / --segmentation-parameters overrides)
        self.params = ts.load_resquiggle_parameters(
            self.samp, m.get('sig_aln_params'), m.get('seg_params'))._replace(
            bandwidth=m['bandwidth'], band_bound_thresh=m['band_bound_thresh'])
        self.max_raw_cpts = m.get('max_raw_cpts') or 200
        seq, raw, starts = synth.synth_read(self.model, m['n_bases'], m['seed'], **m['synth_kw'])
        seq, raw = synth.edit_read(seq, raw, starts, m.get('edit'))
        if m['noise_body']:
            rng = np.random.default_rng(m['seed'] + 12345)
            raw = rng.normal(0.0, 1.0, size=raw.shape[0]) * m['synth_kw']['scale'] + \
                m['synth_kw']['offset']
        assert sha(raw) == str(self.g['raw__sha']), 'synthetic generator drifted'
        self.seq, self.raw = seq, raw
        self.stall_ints = None
        if m['samp'] == 'RNA':
            import oracle
            self.stall_ints = oracle.identify_stalls(raw)
            want = self.g['stall_ints'] if 'stall_ints' in self.g else np.zeros((0, 2))
            got = np.array([[int(a), int(b)] for a, b in self.stall_ints]).reshape(-1, 2)
            assert np.array_equal(got, want), 'identify_stalls differs from the reference'
        self.error = str(self.g['error'])

    def samp_ind(self, n_bases=None):
        """np.random.choice(B, 1000, replace=False) under the recorded seed."""
        # (an edited read's sequence may be longer than the synthetic one it was made from)
        n = len(self.seq) - self.model.kmer_width + 1 if n_bases is None else n_bases
        if n <= 1000:
            return None
        st = np.random.get_state()
        np.random.seed(self.meta['np_seed'])
        idx = np.random.choice(n, 1000, replace=False)
        np.random.set_state(st)
        return idx.astype(np.int64)

    def check_float(self, key, arr, exact=True, tol=0.0):
        """compare arr against the fixture entry (full array, or sha + samples)"""
        g = self.g
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        assert arr.shape[0] == int(g[key + '__len']), (key, arr.shape, int(g[key + '__len']))
        if exact:
            if key in g:
                np.testing.assert_array_equal(arr, g[key], err_msg=key)
            assert sha(arr) == str(g[key + '__sha']), key + ' sha mismatch'
        else:
            if key in g:
                np.testing.assert_allclose(arr, g[key], rtol=0, atol=tol, err_msg=key)
            else:
                np.testing.assert_allclose(arr[:64], g[key + '__head'], rtol=0, atol=tol)
                np.testing.assert_allclose(arr[-64:], g[key + '__tail'], rtol=0, atol=tol)
                np.testing.assert_allclose(arr[::max(1, arr.size // 512)], g[key + '__stride'],
                                           rtol=0, atol=tol)


@pytest.fixture(scope='session')
def golden_case():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = GoldenCase(name)
        return cache[name]
    return get
