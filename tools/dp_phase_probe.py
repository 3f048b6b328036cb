"""Where the cycles of a k_dp row go, for a lone read and under load: run with a
-DTBA_PHASE_DEBUG=5 build (TBA_LIB_PATH=alt_builds/dpphase.so).  Prints shader cycles per row and
part (stamps cost ~50 cycles each, five per row)."""
import os
import sys
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from tombo_amd import _native, resquiggle as rq, synth, tombo_stats as ts, tombo_helper as th  # noqa: E402

PARTS = ['z-scores', 'candidates', 'scan+sweeps', 'cells+stores', 'arg-max', 'placement']


def main():
    bw = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=bw)
    n_max = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    mrs = [synth.synth_map_res(model, 10000, 300 + k, **synth.DNA_SYNTH) for k in range(n_max)]
    eng = rq.get_engine(0)
    for n in (1024, 2048, 4096, n_max):
        rq.resquiggle_batch(mrs[:n], model, params, 5.0, seq_samp_type=samp, subsample_seed=1, return_signal=False)
        rq.resquiggle_batch(mrs[:n], model, params, 5.0, seq_samp_type=samp, subsample_seed=1, return_signal=False)
        d = eng.get(_native.GET_DEBUG_COUNTERS).astype(np.float64)
        rows = d[:, 6].sum()
        per = d[:, :6].sum(axis=0) / rows
        ms = dict(zip(_native.STAGE_NAMES, eng.get(_native.GET_KERNEL_MS)))['main_dp']
        print('W=%d reads=%5d  main_dp %8.3f ms  cycles/row %7.0f : %s' % (
            bw, n, ms, per.sum(), '  '.join('%s %.0f' % (p, v) for p, v in zip(PARTS, per))))
        tot = d[:, :6].sum(axis=1) / 2.39e6
        hw = eng.get(_native.GET_DEBUG_COUNTERS)[:, 7]
        simd = ((hw >> 32) & 15) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 15) * 4 + ((hw >> 4) & 3)
        ids, cnt = np.unique(simd, return_counts=True)
        per_simd = dict(zip(ids, cnt))
        mine = np.array([per_simd[x] for x in simd])
        print('      row-loop ms min/p50/max %.2f %.2f %.2f | SIMDs used %d, waves per SIMD histogram %s | ms by waves-per-SIMD: %s' % (
            tot.min(), np.median(tot), tot.max(), len(ids), dict(zip(*np.unique(cnt, return_counts=True))),
            {int(k): round(float(tot[mine == k].mean()), 2) for k in np.unique(mine)}))

if __name__ == '__main__':
    main()
