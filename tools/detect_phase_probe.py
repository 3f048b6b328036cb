"""Cycles per role of k_detect (TBA_LIB_PATH = a -DTBA_PHASE_DEBUG=7 build): python tools/detect_phase_probe.py [n_reads]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tombo_amd import _native, synth, tombo_stats as ts, tombo_helper as th  # noqa: E402
from tombo_amd._default_parameters import SIG_MATCH_THRESH  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
samp = th.seqSampleType('DNA', False)
model = ts.TomboModel(seq_samp_type=samp)
params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
raws, seqs = [], []
for i in range(n):
    seq, raw, _ = synth.synth_read(model, 10000, 5000 + i, **synth.DNA_SYNTH)
    raws.append(raw)
    seqs.append(ts.encode_seq(seq))
eng = _native.Engine(0)
eng.ensure_model(model)
eng.upload(_native.make_params(params), _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA'], subsample_seed=1), raws, seqs)
for _ in range(2):
    eng.run_stages(_native.STAGE_SEGMENT, _native.STAGE_SEGMENT)
    eng.sync()
d = eng.get(_native.GET_DEBUG_COUNTERS)[::20].astype(np.float64)
steps = d[:, 7]
ms = eng.get(_native.GET_KERNEL_MS)
print('reads', n, 'cumsum stage ms %.3f' % ms[1], 'steps', steps.mean())
for name, k in (('scan', 0), ('loader', 1), ('greedy total', 2), ('  masks', 3), ('  rounds', 4), ('  emission', 5), ('  tail', 6)):
    print('%-14s %8.0f cycles per step' % (name, (d[:, k] / steps).mean()))
