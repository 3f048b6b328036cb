"""Per-stage GPU time of one resident batch (development aid).

    python tools/stage_times.py [--reads N] [--bases B] [--bandwidth W] [--dac] [--dtype i16|f32|f64]
                                [--rna] [--repeat K]

--dac quantises the synthetic pA to int16 DAC values first (tie-heavy change-point scores);
--dtype picks the sample type handed to the engine."""
import os
import sys
import argparse
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=4096)
    ap.add_argument('--bases', type=int, default=10000)
    ap.add_argument('--bandwidth', type=int, default=500)
    ap.add_argument('--dac', action='store_true')
    ap.add_argument('--dtype', default='f64')
    ap.add_argument('--rna', action='store_true')
    ap.add_argument('--repeat', type=int, default=3)
    ap.add_argument('--skip-norm-out', action='store_true')
    ap.add_argument('--libs', default='', help='comma-separated builds of libtombo_amd.so to compare on the same reads')
    a = ap.parse_args()
    from tombo_amd import _native, tombo_stats as ts, tombo_helper as th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    sn = 'RNA' if a.rna else 'DNA'
    samp = th.seqSampleType(sn, a.rna)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=a.bandwidth)
    if a.bandwidth <= 100:
        params = params._replace(band_bound_thresh=10)
    bases = np.full(a.reads, a.bases, np.int64)
    seqs, raws, dacs = bench.make_reads(bases, 1000003, min(32, os.cpu_count() or 8), sn, a.dac)
    # (RNA: the stalls are found on the device since round 3 -- stall_params below; the int16 arrays of
    # make_reads are in acquisition order, hence reverse_raw)
    src = dacs if a.dac else raws
    from tombo_amd._default_parameters import STALL_PARAMS
    dt = {'i16': np.int16, 'f32': np.float32, 'f64': np.float64}[a.dtype]
    src = [r.astype(dt) for r in src]
    rng = np.random.RandomState(1)
    si = np.stack([rng.choice(a.bases, 1000, replace=False) for _ in range(a.reads)]) if a.bases > 1000 else None
    import hashlib
    digests = []
    eng = None
    for lib_path in (a.libs.split(',') if a.libs else [None]):
        if lib_path is not None:   # another build of the library, same process, same reads
            if eng is not None:
                eng.close()
            _native.LIB_PATH, _native._lib = os.path.abspath(lib_path), None
            print('--- %s' % lib_path)
        eng = _native.Engine(0)
        eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
        eng.upload(_native.make_params(params),
                   _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH[sn],
                                     skip_norm_out=a.skip_norm_out, reverse_raw=a.rna and a.dac,
                                     stall_params=th.stallParams(**STALL_PARAMS) if a.rna else None),
                   src, [ts.encode_seq(q) for q in seqs], samp_ind=si)
        eng.run()
        acc = np.zeros(32)
        for _ in range(a.repeat):
            eng.run()
            acc += eng.get(_native.GET_KERNEL_MS)
        acc /= a.repeat
        out = eng.download(want_norm=not a.skip_norm_out)
        print('reads %d bases %d W %d dac %s dtype %s: ok %d' % (a.reads, a.bases, a.bandwidth, a.dac,
                                                               a.dtype, int((out['status'] == 0).sum())))
        print('  '.join('%s %.2f' % (k, v) for k, v in zip(_native.STAGE_NAMES, acc[:16]) if v > 0.004))
        h = hashlib.sha256()   # (large arrays enter through a 64-bit xor / wrapping sum)
        for k in sorted(out):
            if isinstance(out[k], np.ndarray):
                b = np.ascontiguousarray(out[k]).reshape(-1).view(np.uint8)
                b = b[:b.size // 8 * 8].view(np.uint64)
                w = np.arange(1, 1 + min(b.size, 1 << 16), dtype=np.uint64)
                h.update(np.bitwise_xor.reduce(b).tobytes() + b.sum(dtype=np.uint64).tobytes() +
                         (b[:w.size] * w).sum(dtype=np.uint64).tobytes())
        digests.append(h.hexdigest())
        if os.environ.get('TBA_DUMP_READS'):
            # per-read fingerprints of this build's run (to find WHICH reads differ between two runs)
            import zlib
            n = a.reads
            seg_h = np.array([zlib.crc32(out['segs'][eng.seg_off[i]:eng.seg_off[i + 1]].tobytes()) for i in range(n)], np.uint32)
            nrm_h = np.array([zlib.crc32(out['norm'][eng.raw_off[i]:eng.raw_off[i] + int(out['norm_len'][i])].tobytes()) for i in range(n)], np.uint32) \
                if 'norm' in out else np.zeros(n, np.uint32)
            np.savez(os.environ['TBA_DUMP_READS'] + '_%d.npz' % (len(digests) - 1), seg=seg_h, norm=nrm_h,
                     status=out['status'], sv=out['sv'], score=out['score'], read_start=out['read_start'],
                     ts=eng.get(_native.GET_THEIL_SEN), stalls=eng.get(_native.GET_N_STALL) if hasattr(_native, 'GET_N_STALL') else np.zeros(n))
        print('result digest %s' % digests[-1][:16])
        if os.environ.get('TBA_DBG_PHASES'):
            d = eng.get(_native.GET_DEBUG_COUNTERS)
            print('dbg mean', ' '.join('%.0f' % x for x in d.mean(axis=0)))
            print('dbg sum ', ' '.join('%.0f' % x for x in d.sum(axis=0)))
            print('dbg max ', ' '.join('%.0f' % x for x in d.max(axis=0)))

    if len(digests) > 1:
        print('all builds give identical results: %s' % (len(set(digests)) == 1))
    path = eng.get(_native.GET_PATH)
    print('start calls: %s' % np.bincount(path[:, 3], minlength=3).tolist())


if __name__ == '__main__':
    main()
