"""Cycle stamps of k_pick (the cap of the fused event detection) per read: build the library with
-DTBA_PHASE_DEBUG=8 and point TBA_LIB_PATH at it."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tombo_amd import _native, synth, tombo_stats as ts, tombo_helper as th
from tombo_amd._default_parameters import SIG_MATCH_THRESH
n=3000
samp = th.seqSampleType('DNA', False); model = ts.TomboModel(seq_samp_type=samp)
params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
raws, seqs = [], []
for i in range(n):
    seq, raw, _ = synth.synth_read(model, 10000, 5000 + i, **synth.DNA_SYNTH); raws.append(raw); seqs.append(ts.encode_seq(seq))
eng = _native.Engine(0); eng.ensure_model(model)
eng.upload(_native.make_params(params), _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA'], subsample_seed=1), raws, seqs)
for _ in range(2):
    eng.run_stages(_native.STAGE_SEGMENT, _native.STAGE_SEGMENT); eng.sync()
d = eng.get(_native.GET_DEBUG_COUNTERS).astype(np.float64)
m = d.mean(axis=0)
print('cycles to: threshold + counts %.0f  (checks %.0f)  emit %.0f | 100 MHz ticks %.0f' % (m[0], m[1], m[2], m[7]))
