"""Stage times of a batch of one (resquiggle_read, 10 kb DNA, W = 500) and of small batches:
python tools/latency_stages.py"""
import os
import sys
import time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from tombo_amd import _native, resquiggle as rq, synth, tombo_stats as ts, tombo_helper as th  # noqa: E402

samp = th.seqSampleType('DNA', False)
model = ts.TomboModel(seq_samp_type=samp)
params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
mrs = [synth.synth_map_res(model, 10000, 300 + k, **synth.DNA_SYNTH) for k in range(64)]
eng = rq.get_engine(0)
mrs += [synth.synth_map_res(model, 10000, 400 + k, **synth.DNA_SYNTH) for k in range(64, 2048 if len(sys.argv) > 1 else 64)]
for nb in (1, 8, 64) + ((256, 384, 512, 1024, 2048) if len(sys.argv) > 1 else ()):
    for _ in range(3):
        rq.resquiggle_batch(mrs[:nb], model, params, 5.0, seq_samp_type=samp)
    t0 = time.perf_counter()
    rq.resquiggle_batch(mrs[:nb], model, params, 5.0, seq_samp_type=samp)
    dt = (time.perf_counter() - t0) * 1e3
    ms = dict(zip(_native.STAGE_NAMES, [round(float(x), 3) for x in eng.get(_native.GET_KERNEL_MS)]))
    print('batch of %d: call %.2f ms, stages %s' % (nb, dt, {k: v for k, v in ms.items() if v > 0.02}))
