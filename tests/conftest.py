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


# ---- the two dispatch forms of event detection and traceback -------------------------------------
# DNA event detection and the main traceback each have a latency form (small batches) and a
# throughput form (what a 10 000-read batch -- the benchmark -- runs); the engine picks by read
# count (tba_engine_set_dispatch).  Parity tests run through BOTH: 'latency' forces the small-batch
# kernels for any batch, 'throughput' the large-batch ones (k_detect + k_pick, k_normalize without
# its last pass, k_main_tb_par<16>), and `check_forms` asserts through TBA_GET_ED_FORM /
# TBA_GET_TB_FORM that those kernels really produced every read's result.
FORMS = ('latency', 'throughput')
_FORM_THRESHOLDS = {'latency': (1 << 40, 1 << 40), 'throughput': (0, 0), 'default': (1024, 1024)}


class engine_dispatch(object):
    """context manager: the process-wide default engine in one dispatch form"""

    def __init__(self, form):
        self.form = form

    def __enter__(self):
        from tombo_amd import resquiggle as rq
        self.eng = rq.get_engine(0)
        self.eng.set_dispatch(*_FORM_THRESHOLDS[self.form])
        return self.eng

    def __exit__(self, *exc):
        self.eng.set_dispatch(*_FORM_THRESHOLDS['default'])
        return False


@pytest.fixture(params=FORMS)
def dispatch_form(request):
    with engine_dispatch(request.param):
        yield request.param


def check_forms(eng, form, params, require_all_fused=False):
    """every read's change points and main traceback came from the kernels of `form`.
    Returns (ed_form, tb_form) arrays."""
    from tombo_amd import _native as N
    ed, tb = eng.get(N.GET_ED_FORM), eng.get(N.GET_TB_FORM)
    path = eng.get(N.GET_PATH)[:, 0]
    rna = bool(params.use_t_test_seg)
    w, m = params.running_stat_width, params.min_obs_per_base
    ran = ed != N.ED_FORM_NONE
    if rna:
        fused = m == 6 and w <= 64     # TT_MAXW (k_segment.h)
        allowed = {N.ED_FORM_TTEST_PEAKS} | ({N.ED_FORM_DETECT_TT_PICK} if fused else set())
        want = N.ED_FORM_DETECT_TT_PICK if fused else N.ED_FORM_TTEST_PEAKS
    elif form == 'latency':
        # (2w > 64: no fused scan at all, k_cumsum + k_scores_dna in every form)
        want = N.ED_FORM_WG_SCAN_PEAKS if 2 * w <= 64 else N.ED_FORM_SCORES_PEAKS
        allowed = {want}
    else:
        fused = 2 * w <= 32 and m == 3  # DT_W2MAX (k_detect.h)
        want = N.ED_FORM_DETECT_PICK if fused else N.ED_FORM_SCORES_PEAKS
        # flagged reads fall back to the kernels that keep the scores; long reads are scanned by k_long.h
        allowed = {want, N.ED_FORM_SCORES_PEAKS, N.ED_FORM_WG_SCAN_PEAKS}
    assert set(ed[ran].tolist()) <= allowed, (form, sorted(set(ed[ran].tolist())), sorted(allowed))
    if require_all_fused:
        assert np.all(ed[ran] == want), (form, np.flatnonzero(ran & (ed != want)).tolist())
    # traceback: adaptive reads by the chunk-parallel walk of the form (64 lanes per read also for the
    # long reads of any batch), static whole-read bands and broken chains by the lane-per-read walk
    walked = tb != N.TB_FORM_NONE
    want_tb = N.TB_FORM_PAR64 if form == 'latency' else N.TB_FORM_PAR16
    assert set(tb[walked].tolist()) <= {want_tb, N.TB_FORM_PAR64, N.TB_FORM_LANE, N.TB_FORM_LONG}, (form, set(tb[walked].tolist()))
    if form == 'latency':
        assert N.TB_FORM_PAR16 not in set(tb.tolist())
    assert np.all(tb[walked & (path == 2)] == N.TB_FORM_LANE)
    # the verifier behind the chunk-parallel traceback (k_tb_par_verify) agreed with every row it looked at
    vf = eng.get(N.GET_TB_VERIFY_FAIL)
    assert not vf.any(), ('the traceback verifier disagreed', np.flatnonzero(vf).tolist()[:20], vf[vf != 0][:20].tolist())
    return ed, tb
