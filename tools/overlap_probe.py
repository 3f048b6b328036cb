"""Probe: does running sub-batches on several engines (one HIP stream each) overlap the
VALU-bound DP of one sub-batch with the memory-bound stages of another?  GPU box only.

    python tools/overlap_probe.py [reads] [bases] [bandwidth]
"""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from tombo_amd import _native, tombo_stats as ts, tombo_helper as th  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
bases = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
bw = int(sys.argv[3]) if len(sys.argv) > 3 else 500
samp = th.seqSampleType('DNA', False)
model = ts.TomboModel(seq_samp_type=samp)
params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=bw)
seqs, raws = bench.make_reads(reads, bases, 1000003, 32)
rng = np.random.RandomState(12345)
si = np.stack([rng.choice(bases, 1000, replace=False) for _ in range(reads)])
p = _native.make_params(params)
o = _native.make_opts(outlier_thresh=5.0, sig_match_thresh=1.1)

for n_eng, stagger in [(1, 0), (2, 0), (2, 1), (4, 0), (4, 1), (8, 1)]:
    engs = []
    per = reads // n_eng
    for k in range(n_eng):
        e = _native.Engine(0)
        e.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
        e.upload(p, o, raws[k * per:(k + 1) * per], seqs[k * per:(k + 1) * per],
                 samp_ind=si[k * per:(k + 1) * per])
        engs.append(e)
    for e in engs:
        e.run()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        if stagger and n_eng > 1:
            # start the first half, let it get into its DP, then start the rest
            for e in engs[::2]:
                e.enqueue()
            time.sleep(0.012 * per / 2500.0)
            for e in engs[1::2]:
                e.enqueue()
        else:
            for e in engs:
                e.enqueue()
        for e in engs:
            e.sync()
        best = min(best, time.perf_counter() - t0)
    print('engines=%d stagger=%d  %.1f ms  %.0f reads/s' % (n_eng, stagger, best * 1e3,
                                                           per * n_eng / best), flush=True)
    for e in engs:
        e.close()
