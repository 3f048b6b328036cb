"""Event detection of one synthetic batch under the library TBA_LIB_PATH names: change points,
fused-path flags and stage times into an .npz (compare two builds: fused / -DTBA_NO_FUSED_DETECT).
python tools/detect_probe.py out.npz [n_reads] [n_bases] [dac]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tombo_amd import _native, synth, tombo_stats as ts, tombo_helper as th  # noqa: E402
from tombo_amd._default_parameters import SIG_MATCH_THRESH  # noqa: E402

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
dac = len(sys.argv) > 4 and sys.argv[4] == 'dac'
samp = th.seqSampleType('DNA', False)
model = ts.TomboModel(seq_samp_type=samp)
params = ts.load_resquiggle_parameters(samp)
raws, seqs = [], []
for i in range(n):
    seq, raw, _ = synth.synth_read(model, nb + 37 * (i % 11), 9000 + i, **synth.DNA_SYNTH)
    raws.append(np.round(raw * 4).astype(np.int16) if dac else raw)
    seqs.append(ts.encode_seq(seq))
eng = _native.Engine(0)
eng.ensure_model(model)
eng.upload(_native.make_params(params),
           _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA'], subsample_seed=1),
           raws, seqs)
eng.run()
cpts, ncp = eng.get(_native.GET_VALID_CPTS), eng.get(_native.GET_N_CPTS)
try:
    fused = eng.get(_native.GET_ED_FUSED)
except Exception:
    fused = np.zeros(n, np.int32)
ms = eng.get(_native.GET_KERNEL_MS)
np.savez(out, cpts=cpts, ncp=ncp, fused=fused, ev_off=eng.ev_off, status=eng.get(_native.GET_STATUS))
print(os.environ.get('TBA_LIB_PATH', 'tree'), 'fused', int(fused.sum()), 'of', n, 'status ok', int((eng.get(_native.GET_STATUS) == 0).sum()),
      'cumsum+peaks ms', round(float(ms[1] + ms[2] + ms[3]), 3))
