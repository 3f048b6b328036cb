"""How many reads of a cfg2-like batch the chunk-parallel traceback (k_tb_par.h) finishes, and the
stage time of the traceback: python tools/tb_par_probe.py [n_reads] [n_bases] [bandwidth]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tombo_amd import _native, synth, tombo_stats as ts, tombo_helper as th  # noqa: E402
from tombo_amd._default_parameters import SIG_MATCH_THRESH  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
bw = int(sys.argv[3]) if len(sys.argv) > 3 else 500
samp = th.seqSampleType('DNA', False)
model = ts.TomboModel(seq_samp_type=samp)
params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=bw)
raws, seqs = [], []
for i in range(n):
    seq, raw, _ = synth.synth_read(model, nb, 5000 + i, **synth.DNA_SYNTH)
    raws.append(raw)
    seqs.append(ts.encode_seq(seq))
eng = _native.Engine(0)
eng.ensure_model(model)
eng.upload(_native.make_params(params),
           _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA'], subsample_seed=1),
           raws, seqs)
for _ in range(2):
    eng.run()
done, path, status = eng.get(_native.GET_TB_PARALLEL), eng.get(_native.GET_PATH)[:, 0], eng.get(_native.GET_STATUS)
ms = eng.get(_native.GET_KERNEL_MS)
print('reads', n, 'ok', int((status == 0).sum()), 'adaptive', int((path == 1).sum()),
      'walked chunk-parallel', int(done.sum()), 'left to the serial walk', np.flatnonzero((path == 1) & (done != 1))[:20].tolist())
print('event detection without the score array finished', int(eng.get(_native.GET_ED_FUSED).sum()), 'of', n)
print('stage ms', dict(zip(_native.STAGE_NAMES, [round(float(x), 3) for x in ms[:len(_native.STAGE_NAMES)]])))
