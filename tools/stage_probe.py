"""Stage times of a sub-range of the pipeline on device-made reads (no host synthesis):
python tools/stage_probe.py <DNA|RNA> <n reads> <first stage> <last stage> [repeats]
stages: 0 segment, 1 event_means, 2 ref_levels, 3 start, 4 assign, 5 skip, 6 rescale.
TBA_LIB_PATH picks the library (A/B of experiment builds)."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tombo_amd import _native, synth, tombo_stats as ts, tombo_helper as th
from tombo_amd._default_parameters import SIG_MATCH_THRESH, STALL_PARAMS
samp_name, n, s0, s1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rep = int(sys.argv[5]) if len(sys.argv) > 5 else 3
rna = samp_name == 'RNA'
samp = th.seqSampleType(samp_name, rna); model = ts.TomboModel(seq_samp_type=samp)
params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
g = _native.Synth(model, 0)
sp = _native.make_synth_params(**(synth.RNA_SYNTH if rna else synth.DNA_SYNTH))
raw, raw_off, seq, seq_off = g.generate(sp, 1, np.full(n, 3000 if rna else 10000), raw_dtype=np.float64)
eng = _native.Engine(0); eng.ensure_model(model)
o = _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH[samp_name], subsample_seed=1,
                      stall_params=th.stallParams(**STALL_PARAMS) if rna else None)
eng.upload_packed(_native.make_params(params), o, raw, raw_off, seq, seq_off, wait=True)
tot = np.zeros(32)
for k in range(rep + 1):
    eng.run_stages(0, s1); eng.sync()
    if k:
        tot += eng.get(_native.GET_KERNEL_MS)
print(os.path.basename(os.environ.get('TBA_LIB_PATH', 'tree')), samp_name, n, ' '.join(
    '%s=%.3f' % (a, b / rep) for a, b in zip(_native.STAGE_NAMES, tot[:16]) if b > 0))
