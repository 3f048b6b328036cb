"""Run one resident batch K times and report, stage by stage, which reads came out differently from
the first run (development aid: a pipeline whose result depends on timing has a race).

    python tools/determinism_probe.py [--reads N] [--bases B] [--rna] [--runs K] [--bandwidth W]"""
import os
import sys
import argparse
import zlib
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=10000)
    ap.add_argument('--bases', type=int, default=3000)
    ap.add_argument('--bandwidth', type=int, default=500)
    ap.add_argument('--rna', action='store_true')
    ap.add_argument('--runs', type=int, default=8)
    a = ap.parse_args()
    from tombo_amd import _native as N, tombo_stats as ts, tombo_helper as th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH, STALL_PARAMS
    sn = 'RNA' if a.rna else 'DNA'
    samp = th.seqSampleType(sn, a.rna)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=a.bandwidth)
    bases = np.full(a.reads, a.bases, np.int64)
    seqs, raws, _ = bench.make_reads(bases, 1000003, min(32, os.cpu_count() or 8), sn, False)
    rng = np.random.RandomState(1)
    si = np.stack([rng.choice(a.bases, 1000, replace=False) for _ in range(a.reads)]) if a.bases > 1000 else None
    eng = N.Engine(0)
    eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    eng.upload(N.make_params(params),
               N.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH[sn],
                           stall_params=th.stallParams(**STALL_PARAMS) if a.rna else None),
               raws, [ts.encode_seq(q) for q in seqs], samp_ind=si)
    n = a.reads

    def per_read(flat, off):
        return np.array([zlib.crc32(flat[off[i]:off[i + 1]].tobytes()) for i in range(n)], np.uint32)

    def snapshot():
        out = eng.download()
        s = {}
        s['n_cpts'] = eng.get(N.GET_N_CPTS)
        s['valid_cpts'] = per_read(eng.get(N.GET_VALID_CPTS), eng.ev_off)
        s['event_means'] = per_read(eng.get(N.GET_EVENT_MEANS), eng.ev_off)
        s['n_stall'] = eng.get(N.GET_N_STALL)
        s['seg_sv'] = eng.get(N.GET_SEG_SV)
        s['start'] = eng.get(N.GET_START)
        s['band_starts'] = per_read(eng.get(N.GET_BAND_STARTS), eng.ref_off)
        s['last_row'] = np.array([zlib.crc32(r.tobytes()) for r in eng.get(N.GET_LAST_ROW)], np.uint32)
        s['read_tb'] = per_read(eng.get(N.GET_READ_TB), eng.seg_off)
        s['dp_segs'] = per_read(eng.get(N.GET_DP_SEGS), eng.seg_off)
        s['segs'] = per_read(eng.get(N.GET_SEGS), eng.seg_off)
        s['theil_sen'] = eng.get(N.GET_THEIL_SEN)
        s['tb_form'] = eng.get(N.GET_TB_FORM)
        if os.environ.get('TBA_DBG_PHASES'):
            s['dbg'] = eng.get(N.GET_DEBUG_COUNTERS)
        s['ed_form'] = eng.get(N.GET_ED_FORM)
        s['status'] = out['status']
        s['score'] = out['score']
        s['norm'] = np.array([zlib.crc32(out['norm'][eng.raw_off[i]:eng.raw_off[i] + int(out['norm_len'][i])].tobytes())
                              for i in range(n)], np.uint32)
        return s

    first = None
    keep = {}
    for k in range(a.runs):
        eng.run()
        s = snapshot()
        s_tb = eng.get(N.GET_READ_TB)
        if first is None:
            first = s
            first_tb = s_tb
            keep['band_starts'] = eng.get(N.GET_BAND_STARTS)
            keep['path'] = eng.get(N.GET_PATH)
            keep['seg_off'], keep['ref_off'] = eng.seg_off, eng.ref_off
            print('run 0: ok %d, side stream %s' % (int((s['status'] == 0).sum()), eng.last_side_stream()))
            continue
        rep = []
        for key in first:
            x, y = first[key], s[key]
            ne = x != y
            if x.dtype.kind == 'f':
                ne &= ~(np.isnan(x) & np.isnan(y))
            idx = np.flatnonzero(ne.reshape(n, -1).any(axis=1))
            if idx.size:
                rep.append('%s: %d reads %s' % (key, idx.size, idx[:12].tolist()))
        print('run %d: %s' % (k, 'same as run 0' if not rep else ' | '.join(rep)))
        if rep and os.environ.get('TBA_PROBE_SAVE'):
            bad = np.flatnonzero(first['read_tb'] != s['read_tb'])
            np.savez(os.environ['TBA_PROBE_SAVE'] + '_run%d.npz' % k, bad=bad,
                     **{'tb0_%d' % i: first_tb[eng.seg_off[i]:eng.seg_off[i + 1]] for i in bad},
                     **{'tb1_%d' % i: s_tb[eng.seg_off[i]:eng.seg_off[i + 1]] for i in bad},
                     **{'bst_%d' % i: keep['band_starts'][eng.ref_off[i]:eng.ref_off[i + 1]] for i in bad},
                     **{'path_%d' % i: keep['path'][i] for i in bad})
        if rep and os.environ.get('TBA_PROBE_DETAIL'):
            bad = np.flatnonzero(first['segs'] != s['segs'])[:4]
            for i in bad:
                if 'dbg' in s:
                    print('   read %d dbg run0 %s | now %s' % (i, first['dbg'][i].tolist(), s['dbg'][i].tolist()))
                print('   read %d raw_off %d n_raw %d tb_form %d/%d status %d/%d' % (
                    i, eng.raw_off[i], eng.raw_off[i + 1] - eng.raw_off[i], first['tb_form'][i], s['tb_form'][i],
                    first['status'][i], s['status'][i]))


if __name__ == '__main__':
    main()
