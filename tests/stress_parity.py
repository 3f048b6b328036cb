"""One-off stress run of the engine against the oracle beyond the default test budget: a larger
hypothesis sweep of the parameter box, start-retry-heavy reads (long leaders, every k_dp_wg class),
stall detection at awkward lengths for every boundary type.  GPU box:  python tests/stress_parity.py  (not collected by pytest: no test_ prefix)"""
import os
import sys
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))  # (this directory)
import oracle  # noqa: E402  (checker)
from tombo_amd import synth, tombo_stats as ts, tombo_helper as th  # noqa: E402
from test_gpu_parity import run_batch, compare_batch  # noqa: E402


def retry_heavy():
    n_bad = n_retry = 0
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    rng = np.random.default_rng(1)
    for save_bw, start_bw, nbases in ((2500, 750, 250), (900, 300, 120), (1800, 500, 200), (3000, 1000, 250), (640, 200, 60)):
        aln = (4.2, 4.2, 300, 1500, 20.0, 40, start_bw, save_bw, nbases)
        params = ts.load_resquiggle_parameters(samp, aln)
        reads = []
        for k in range(24):
            nb = int(rng.integers(nbases + 50, 2500))
            lead = int(rng.integers(200, 9 * save_bw))     # leaders around and beyond both start bands
            kw = dict(synth.DNA_SYNTH, lead=lead)
            seq, raw, _ = synth.synth_read(model, nb, 4000 + 31 * k + save_bw, **kw)
            rs = np.random.RandomState(k)
            reads.append((raw, seq, None, rs.choice(nb, 1000, replace=False).astype(np.int64) if nb > 1000 else None))
        eng, out, oracles = run_batch(model, params, 'DNA', reads)
        bad = compare_batch(eng, oracles, out, 'retry%d' % save_bw)
        n_bad += len(bad)
        n_retry += sum(o['dbg']['n_start_calls'] == 2 for o in oracles)
        for b in bad[:5]:
            print(b)
    print('retry-heavy: %d mismatches, %d reads took the retry' % (n_bad, n_retry))
    return n_bad


def stalls_awkward():
    n_bad = 0
    rng = np.random.default_rng(2)
    for n in (349, 350, 351, 2047, 2048, 2049, 2048 + 175, 4096 - 175, 4096, 8191, 65536, 65537, 1856 * 3, 262144, 262145, 300001):
        for rep in range(2):
            x = np.repeat(rng.normal(0, 1, n // 30 + 2), 30)[:n] * 90 + 500 + rng.normal(0, 20, n)
            for _ in range(3):
                a = int(rng.integers(0, max(1, n - 2000)))
                ln = int(rng.integers(150, 3000))
                x[a:a + ln] = x[min(a, n - 1)] + rng.normal(0, 3.0, x[a:a + ln].shape[0])
            for arr in (x, x.astype(np.float32), np.round(x).astype(np.int16)):
                want = np.array([[int(a), int(b)] for a, b in oracle.identify_stalls(arr.astype(np.float64))]).reshape(-1, 2)
                got = np.array([[int(a), int(b)] for a, b in ts.identify_stalls(arr)]).reshape(-1, 2)
                if not np.array_equal(want, got):
                    n_bad += 1
                    print('stalls differ: n=%d dtype=%s' % (n, arr.dtype), want[:3], got[:3])
    print('stall lengths: %d mismatches' % n_bad)
    return n_bad


def big_fuzz(n_examples, form):
    import test_gpu_param_fuzz as f
    from hypothesis import settings, HealthCheck, given

    @settings(max_examples=n_examples, deadline=None, derandomize=False, database=None,
              suppress_health_check=list(HealthCheck))
    @given(f.param_points())
    def t(pt):
        f._box_point(form, pt)
    t()
    print('fuzz: %d examples ok' % n_examples)
    return 0


if __name__ == '__main__':
    from conftest import FORMS, engine_dispatch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    bad = 0
    for form in FORMS:      # the latency and the throughput kernels of event detection / traceback
        print('--- dispatch form: %s' % form)
        with engine_dispatch(form):
            bad += retry_heavy() + big_fuzz(n, form)
    bad += stalls_awkward()
    print('STRESS', 'OK' if bad == 0 else 'FAILED (%d)' % bad)
