"""The chunk-parallel traceback under repetition: one resident batch run K times; read_tb, status, tb_form and the
verifier's count of every run against run 0 (round-6 fault hunt; profiles/r06_traceback_rootcause.txt).

    TBA_LIB_PATH=<build> python tools/tb_hunt.py [--rna] [--reads N] [--bases B] [--runs K] [--aux b2]

One line per run that differs, and a summary
    HUNT <tag> <DNA|RNA> runs K DISTINCT_RESULTS D (minority runs M) bad_reads .. verify_fail_rows_per_run [..]
(D = 1: every run gave the same bytes).  --aux b2: a -DTBA_TB_B2 experiment build -- phase B's stores live in a
second array and the state every lane entered phase B with in a third; those are what is compared (read_tb holds
phase A's values only in such a build), and the boundaries whose first row differs are printed with their entry
state, the band starts and both versions of what phase B stored."""
import os
import sys
import argparse
import collections
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
    ap.add_argument('--runs', type=int, default=20)
    ap.add_argument('--aux', default='', help="'b2': read the experiment build's second / third array; 'times': a -DTBA_TB_TIMES build -- "
                    "for the wavefronts of a run that differs, who shared their SIMD and when")
    ap.add_argument('--tag', default=os.path.basename(os.environ.get('TBA_LIB_PATH', 'tree')))
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
    off = np.asarray(eng.seg_off)

    def read_of(pos):
        return np.unique(np.searchsorted(off, pos, side='right') - 1)

    digests, vfail, bad_reads = [], [], set()
    for k in range(a.runs):
        eng.run()
        tb = eng.get(N.GET_READ_TB)
        st = eng.get(N.GET_STATUS)
        fm = eng.get(N.GET_TB_FORM)
        vf = eng.get(N.GET_TB_VERIFY_FAIL)
        vfail.append(int(vf.sum()))
        if vf.any():
            w = np.flatnonzero(vf)
            print('run %d: the verifier disagreed on %d rows of reads %s (wavefronts %s); their tb_form now %s' % (
                k, vfail[-1], w[:12].tolist(), sorted(set((w // 4).tolist()))[:6], fm[w[:12]].tolist()), flush=True)
        if a.aux == 'b2':
            b23 = eng.get(97)
            b2, b3 = b23[:tb.size], b23[tb.size:2 * tb.size]
            digests.append(zlib.crc32(b2.tobytes()))
        else:
            digests.append(zlib.crc32(tb.tobytes()) ^ zlib.crc32(st.tobytes()))
        if k == 0:
            tb0, st0, fm0 = tb, st, fm
            if a.aux == 'b2':
                b20, b30 = b2, b3
                print('b2: %d entries stored by phase B in run 0' % int((b2 != -1).sum()))
            continue
        d = np.flatnonzero(tb != tb0)
        ds = np.flatnonzero(st != st0)
        df = np.flatnonzero(fm != fm0)
        if a.aux == 'times' and d.size:
            dbg = eng.get(N.GET_DEBUG_COUNTERS)[::4]          # one record per wavefront (its first read)
            t0, t1, hw, b0, b1 = dbg[:, 0], dbg[:, 1], dbg[:, 2], dbg[:, 4], dbg[:, 5]
            place = (hw >> 32 & 0xf) << 16 | (hw & 0xffff & ~0xf)     # XCC, SE, SH, CU, pipe, SIMD (wave slot masked out)
            for wv in sorted(set((read_of(d) // 4).tolist()))[:4]:
                same = np.flatnonzero(place == place[wv])
                ov = [int(j) for j in same if j != wv and t0[j] < t1[wv] and t1[j] > t0[wv]]
                print('   wavefront %d: xcc %d hw_id %04x (slot %d), ran %d..%d (%.1f us); on its SIMD %d wavefronts of the kernel in all, '
                      'overlapping it in time: %s' % (
                          wv, hw[wv] >> 32 & 0xf, hw[wv] & 0xffff, hw[wv] & 0xf, t0[wv] - t0.min(), t1[wv] - t0.min(), (t1[wv] - t0[wv]) / 100.0,
                          same.size, [(j, int(hw[j] & 0xf), int(t0[j] - t0.min()), int(t1[j] - t0.min())) for j in ov]), flush=True)
                # the largest number of wavefronts alive at once on that SIMD
                ev = sorted([(int(t0[j]), 1) for j in same] + [(int(t1[j]), -1) for j in same])
                cur = mx = 0
                for _, dlt in ev:
                    cur += dlt
                    mx = max(mx, cur)
                print('      most wavefronts of this kernel alive at once on that SIMD: %d; its phase B ran %d..%d; SIMD neighbours ended at %s, '
                      'started at %s (ticks of 10 ns from the kernel\'s first wavefront)' % (
                          mx, b0[wv] - t0.min(), b1[wv] - t0.min(), [int(t1[j] - t0.min()) for j in same if j != wv],
                          [int(t0[j] - t0.min()) for j in same if j != wv]), flush=True)
            # how common is "a SIMD neighbour ends (or starts) inside my phase B" among ALL wavefronts of this run?
            order = np.argsort(place, kind='stable')
            n_end = n_start = 0
            grp = collections.defaultdict(list)
            for j in range(place.size):
                grp[int(place[j])].append(j)
            for js in grp.values():
                for j in js:
                    for q in js:
                        if q != j:
                            n_end += int(b0[j] <= t1[q] <= b1[j])
                            n_start += int(b0[j] <= t0[q] <= b1[j])
            print('      of the %d wavefronts of this run: %d have a SIMD neighbour ENDING inside their phase B, %d one STARTING' % (place.size, n_end, n_start), flush=True)
        if d.size or ds.size or df.size:
            rd = read_of(d)
            bad_reads.update(rd.tolist())
            print('run %d: read_tb differs from run 0 at %d entries of reads %s (wavefronts %s); status differs %s; tb_form differs %s' % (
                k, d.size, rd[:16].tolist(), sorted(set((rd // 4).tolist()))[:8], ds[:8].tolist(),
                [(int(i), int(fm0[i]), int(fm[i])) for i in df[:8]]), flush=True)
        if a.aux == 'b2':
            db = np.flatnonzero(b2 != b20)
            d3 = np.flatnonzero(b3 != b30)
            if d3.size:
                print('run %d: phase B ENTRY STATE differs at %d boundaries of reads %s' % (k, d3.size, read_of(d3)[:8].tolist()), flush=True)
            if db.size:
                bad_reads.update(read_of(db).tolist())
                print('run %d: what phase B stored differs at %d entries of reads %s' % (k, db.size, read_of(db)[:8].tolist()), flush=True)
                if k <= 3:
                    bst = eng.get(N.GET_BAND_STARTS)
                    roff = np.asarray(eng.ref_off)
                    ent = np.flatnonzero(b3 != -1)                 # index lo of every lane that extended
                    shown = 0
                    for p in db[::-1]:                             # (top-down inside a boundary: highest index first)
                        j = np.searchsorted(ent, p, side='right')
                        if j >= ent.size or ent[j] - p != 1:
                            continue                               # only the first row under a chunk top
                        lo_i = ent[j]
                        i = int(np.searchsorted(off, p, side='right') - 1)
                        lo = int(lo_i - off[i])
                        rows = range(lo - 1, lo - 7, -1)
                        print('   read %d boundary lo=%d: entry state cur %d band cell %d (run 0: %d %d) | band starts %s | phase A wrote %s | '
                              'phase B stored, run 0: %s | now: %s' % (
                                  i, lo, int(b3[lo_i]) & (2**40 - 1), int(b3[lo_i]) >> 40, int(b30[lo_i]) & (2**40 - 1), int(b30[lo_i]) >> 40,
                                  [int(bst[roff[i] + r]) for r in rows], [int(tb[off[i] + r]) for r in rows],
                                  [int(b20[off[i] + r]) for r in rows], [int(b2[off[i] + r]) for r in rows]), flush=True)
                        shown += 1
                        if shown >= 6:
                            break
    cnt = collections.Counter(digests)
    print('HUNT %s %s runs %d DISTINCT_RESULTS %d (minority runs %d) bad_reads %d %s forms %s ok %d verify_fail_rows_per_run %s' % (
        a.tag, sn, a.runs, len(cnt), a.runs - max(cnt.values()), len(bad_reads), sorted(bad_reads)[:12],
        dict(collections.Counter(fm0.tolist())), int((st0 == 0).sum()), vfail), flush=True)


if __name__ == '__main__':
    main()
