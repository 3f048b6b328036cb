"""Completion times of 1 long + 3 ordinary batches submitted together on 4 engines: resident
(enqueue only) vs streamed (pinned upload + enqueue + download), to find what serialises them."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
import bench
from tombo_amd import _native as N, streaming, tombo_stats as ts, tombo_helper as th
from tombo_amd._default_parameters import SIG_MATCH_THRESH


def main():
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
    sets = [bench.make_reads(np.full(300, 100000), 1, 32, 'DNA', True)] + \
           [bench.make_reads(np.full(5000, 8000), 1000 * (k + 1), 32, 'DNA', True) for k in range(3)]
    p = N.make_params(params)
    o = N.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['DNA'], subsample_seed=1, skip_norm_out=True)
    engs = [N.Engine(0) for _ in sets]
    stages = [N.PinnedStage() for _ in sets]
    packed = []
    for e, st, (seqs, raws, dacs) in zip(engs, stages, sets):
        e.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
        packed.append(N.pack_reads(dacs, seqs, stage=st))
        raw, raw_off, seq, seq_off, _ = packed[-1]
        e.upload_packed(p, o, raw, raw_off, seq, seq_off, wait=True)
        e.run()
    outs = [(N.PinnedArray(e.n, N.RESULT_DTYPE), N.PinnedArray(int(e.seg_off[-1]), np.int32)) for e in engs]

    def wait_all(t0, label):
        done = [None] * len(engs)
        while any(d is None for d in done):
            for k, e in enumerate(engs):
                if done[k] is None and not e.query():
                    done[k] = time.perf_counter() - t0
            time.sleep(0.0002)
        print('%-34s' % label, ' '.join('%7.1f' % (d * 1e3) for d in done), 'ms  (long, short x3)')
    for rep in range(2):
        t0 = time.perf_counter()
        for e in engs:
            e.enqueue()
        wait_all(t0, 'resident, enqueue only')
        t0 = time.perf_counter()
        for e, pk in zip(engs, packed):
            raw, raw_off, seq, seq_off, _ = pk
            e.upload_packed(p, o, raw, raw_off, seq, seq_off)
            e.enqueue()
        wait_all(t0, 'upload + enqueue')
        t0 = time.perf_counter()
        for e, pk, (ores, osegs) in zip(engs, packed, outs):
            raw, raw_off, seq, seq_off, _ = pk
            e.upload_packed(p, o, raw, raw_off, seq, seq_off)
            e.enqueue()
            e.download_async(results=ores.a, segs32=osegs.a)
        wait_all(t0, 'upload + enqueue + download')


if __name__ == '__main__':
    main()
