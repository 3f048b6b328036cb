"""Does a batch of a few hundred LONG reads (one wavefront per read, ~0.15 s serial chains) slow a
batch of ordinary reads running beside it on another engine / stream?  Prints, for the short batch,
its completion time alone and next to the long batch, with per-stage GPU times.

    python tools/overlap_probe2.py [long_reads] [long_bases] [short_reads] [short_bases]
"""
import os
import sys
import time
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tombo_amd import _native as N, tombo_stats as ts, tombo_helper as th  # noqa: E402
from tombo_amd._default_parameters import SIG_MATCH_THRESH  # noqa: E402


def main(nl=300, bl=100000, nsh=10000, bs=5000):
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
    sl, rl, _ = bench.make_reads(np.full(nl, bl), 1, 32)
    ss, rs_, _ = bench.make_reads(np.full(nsh, bs), 100000, 32)
    p = N.make_params(params)
    o = N.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA'], subsample_seed=1)
    A, B = N.Engine(0), N.Engine(0)
    for e, seqs, raws in ((A, sl, rl), (B, ss, rs_)):
        e.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
        e.upload(p, o, raws, [ts.encode_seq(s) for s in seqs])
        e.run()

    def show(tag, e, t):
        ms = e.get(N.GET_KERNEL_MS)
        print('%-28s done after %6.1f ms; stages: %s' % (tag, t * 1e3, ' '.join(
            '%s=%.1f' % (k, v) for k, v in zip(N.STAGE_NAMES, ms[:16]) if v > 0.5)))
    for name, first in (('alone', None), ('beside the long batch', A)):
        t0 = time.perf_counter()
        if first is not None:
            first.enqueue()
            time.sleep(0.02)
        t1 = time.perf_counter()
        B.enqueue()
        B.sync()
        tb = time.perf_counter() - t1
        if first is not None:
            first.sync()
            show('long batch', A, time.perf_counter() - t0)
        show('short batch ' + name, B, tb)


if __name__ == '__main__':
    main(*[int(x) for x in sys.argv[1:]])
