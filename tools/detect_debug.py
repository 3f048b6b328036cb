"""The taken list of k_detect against a numpy greedy over the engine's own normalised signal:
first disagreement per read.  python tools/detect_debug.py [n_reads] [n_bases]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tombo_amd import _native, synth, tombo_stats as ts, tombo_helper as th  # noqa: E402
from tombo_amd._default_parameters import SIG_MATCH_THRESH  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 600
samp = th.seqSampleType('DNA', False)
model = ts.TomboModel(seq_samp_type=samp)
params = ts.load_resquiggle_parameters(samp)
raws, seqs = [], []
for i in range(n):
    seq, raw, _ = synth.synth_read(model, nb, 9000 + i, **synth.DNA_SYNTH)
    raws.append(raw)
    seqs.append(ts.encode_seq(seq))
eng = _native.Engine(0)
eng.ensure_model(model)
eng.upload(_native.make_params(params),
           _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA'], subsample_seed=1),
           raws, seqs)
eng.run_stages(_native.STAGE_SEGMENT, _native.STAGE_SEGMENT)
eng.sync()
norm = eng.get(_native.GET_SEG_NORM)
pos = eng.get(_native.GET_ED_TAKEN_POS)
nt = eng.get(_native.GET_ED_N_TAKEN)
w, R = 5, 2
for i in range(n):
    a, b = int(eng.raw_off[i]), int(eng.raw_off[i + 1])
    x = norm[a:b]
    c = np.concatenate([[0.0], np.cumsum(x)])
    ns = len(x) + 1 - 2 * w
    s = np.abs(((2 * c[w:w + ns]) - c[:ns]) - c[2 * w:2 * w + ns])
    order = np.lexsort((np.arange(ns), s))[::-1]
    taken, blocked = np.zeros(ns, bool), np.zeros(ns, bool)
    for p in order:
        if blocked[p]:
            continue
        taken[p] = True
        blocked[max(0, p - R):p + R + 1] = True
    want = np.flatnonzero(taken)
    got = pos[2 * a:2 * a + int(nt[i])]
    m = min(len(want), len(got))
    k = int(np.flatnonzero(want[:m] != got[:m])[0]) if (want[:m] != got[:m]).any() else -1
    print('read', i, 'n_taken', len(want), int(nt[i]), 'first diff idx', k,
          'want', want[max(k - 2, 0):k + 4] if k >= 0 else '', 'got', got[max(k - 2, 0):k + 4] if k >= 0 else '',
          'slot of first diff', (int(want[k]) + 2 * w - 1, int(got[k]) + 2 * w - 1) if k >= 0 else '')
    if k >= 0:
        p0 = int(min(want[k], got[k]))
        print('   scores around', p0, s[max(p0 - 3, 0):p0 + 4])
