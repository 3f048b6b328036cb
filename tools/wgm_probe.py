"""Per-row counters of the workgroup-per-read forward pass (k_dp_wgm.h) on one 10 kb read: build with
-DTBA_WGM_STATS and point TBA_LIB_PATH at the library."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tombo_amd import _native, resquiggle as rq, synth, tombo_stats as ts, tombo_helper as th
samp = th.seqSampleType('DNA', False); model = ts.TomboModel(seq_samp_type=samp)
params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=int(sys.argv[1]) if len(sys.argv) > 1 else 500)
mrs = [synth.synth_map_res(model, 10000, 300 + k, **synth.DNA_SYNTH) for k in range(4)]
eng = rq.get_engine(0)
for _ in range(3):
    rq.resquiggle_batch(mrs, model, params, 5.0, seq_samp_type=samp)
d = eng.get(_native.GET_DEBUG_COUNTERS).astype(np.float64)
ms = dict(zip(_native.STAGE_NAMES, eng.get(_native.GET_KERNEL_MS)))
rows = d[:, 4]
print('main_dp %.2f ms; per row: sweeps of wavefronts 0..3 %s, rounds %.2f; cycles of wavefront 0 in the sweep phase %.0f, of which waiting at its barriers %.0f'
      % (ms['main_dp'], ' '.join('%.2f' % (d[:, w] / rows).mean() for w in range(4)), (d[:, 5] / rows).mean(),
         (d[:, 6] / rows).mean(), (d[:, 7] / rows).mean()))
