import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
from conftest import GoldenCase
from test_gpu_parity import run_batch
from tombo_amd import _native as N
c = GoldenCase('dna_noise_body')
for rep in range(1):
    reads = [(c.raw, c.seq, None, None)] * (1 if rep < 2 else 3)
    eng, out, oracles = run_batch(c.model, c.params, 'DNA', reads)
    lr = eng.get(N.GET_LAST_ROW)
    path = eng.get(N.GET_PATH)
    for i, o in enumerate(oracles):
        d = o['dbg']
        w = len(d['fwd_last_row'])
        diff = np.flatnonzero(lr[i, :w] != d['fwd_last_row'])
        bst = eng.get(N.GET_BAND_STARTS)[eng.ref_off[i]:eng.ref_off[i+1]]
        print('rep', rep, 'read', i, 'status', out['status'][i], 'path', path[i], 'W', w,
              'n_diff', diff.size, 'first/last', (diff[:3], diff[-3:]) if diff.size else None,
              'bst_equal', np.array_equal(bst, d['band_event_starts']))
        if diff.size:
            print('   gpu', lr[i, diff[:4]], 'orc', d['fwd_last_row'][diff[:4]])
