"""Golden vectors of the WORKER LOOP (build container only): the per-read sequence of
`_resquiggle_worker` (tombo/resquiggle.py:1492-1504: run_rsqgl_iters; :1578-1589: the
save-parameter retry) driven over a list of reads under ONE numpy seed, through the live
reference's resquiggle_read.  The loop itself is a nested function of the worker process and
cannot be imported, so its dozen lines are restated here around the reference's own
resquiggle_read; what is recorded is data: final boundaries, scale values, scores and the
number of passes per read.  Writes tests/golden/loop_dna.npz and loop_rna.npz.

RNA: the reads are handed over as `_io_and_map_read` leaves them -- float64 signal in ACQUISITION
order (3'->5'), no stall intervals -- and `adjust_map_res` (resquiggle.py:1506-1530: the flip and
`ts.identify_stalls(..., DEFAULT_STALL_PARAMS)`; TRIM_RNA_ADAPTER is off) runs on them before the
loop, as in the worker.  The stall intervals the reference found are recorded too.
"""
import os
import sys
import json
import hashlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import ref_oracle  # noqa: E402
from tombo_amd import synth, tombo_stats as my_ts, tombo_helper as my_th  # noqa: E402
from gen_golden import ref_model  # noqa: E402

rq, ts, th = ref_oracle.load()

# narrow main band: some reads need the save-bandwidth retry; several reads above 1000 bases
# draw Theil-Sen subsamples, so the order of the draws matters
ALN = (4.2, 4.2, 100, 1500, 20.0, 40, 750, 2500, 250)
SPECS = [(1400, 1, {}), (700, 2, {}), (1800, 3, dict(lead=4000)), (1200, 4, dict(scale=14.5, offset=70.0)),
         (300, 5, {}), (2200, 6, {}), (1100, 7, dict(mean_dwell=14)), (1600, 8, dict(scale=9.0))]
SEED, MAX_ITERS, OUTLIER = 20260927, 3, 5.0
# RNA: a band of 160 events (default 500) sends some reads to the save-bandwidth retry; reads above
# 1000 bases draw subsamples; other offsets / ranges move the event-based scaling
RNA_ALN = (2.0, 2.0, 160, 3000, 14.0, 40, 1000, 3000, 250)
RNA_SPECS = [(700, 1, {}), (1300, 2, {}), (450, 3, dict(scale=95.0, offset=420.0)), (1100, 4, dict(mean_dwell=60)),
             (900, 5, dict(lead=3000)), (1500, 6, dict(scale=70.0))]


def main(name='DNA'):
    samp = th.seqSampleType(name, False)
    rna = name == 'RNA'
    aln, specs = (RNA_ALN, RNA_SPECS) if rna else (ALN, SPECS)
    my_model = my_ts.TomboModel(seq_samp_type=my_th.seqSampleType(name, False))
    std_ref = ref_model(my_model, samp)
    params = ts.load_resquiggle_parameters(samp, aln)
    save_params = ts.load_resquiggle_parameters(samp, aln, use_save_bandwidth=True)
    out = {}
    np.random.seed(SEED)
    n_pass, errs, saved = [], [], []
    for k, (nb, seed, kw) in enumerate(specs):
        skw = dict(synth.RNA_SYNTH if rna else synth.DNA_SYNTH)
        skw.update(kw)
        seq, raw, _ = synth.synth_read(my_model, nb, 7000 + seed, **skw)
        if rna:
            raw = np.ascontiguousarray(raw[::-1])   # what the FAST5 holds: acquisition order
        map_res = th.resquiggleResults(
            align_info=th.alignInfo('r%d' % k, 'BaseCalled_template', 0, 0, 0, 0, nb, 0),
            genome_loc=th.genomeLocation(0, '+', 'synth'), genome_seq=seq, mean_q_score=10.0,
            raw_signal=raw)
        if rna:                       # adjust_map_res, resquiggle.py:1506-1530
            map_res = map_res._replace(raw_signal=map_res.raw_signal[::-1])
            map_res = map_res._replace(stall_ints=ts.identify_stalls(
                map_res.raw_signal, rq.DEFAULT_STALL_PARAMS))
            out['stall_ints%d' % k] = np.array(
                [[int(a), int(b)] for a, b in map_res.stall_ints], dtype=np.int64).reshape(-1, 2)
        passes = [0]

        def run_rsqgl_iters(mr, p):   # resquiggle.py:1492-1504
            passes[0] += 1
            res = rq.resquiggle_read(mr, std_ref, p, OUTLIER, seq_samp_type=samp)
            n_iters = 1
            while n_iters < MAX_ITERS and res.norm_params_changed:
                passes[0] += 1
                res = rq.resquiggle_read(mr._replace(scale_values=res.scale_values), std_ref, p,
                                         OUTLIER, all_raw_signal=mr.raw_signal, seq_samp_type=samp)
                n_iters += 1
            return res
        err = ''
        try:
            try:
                res = run_rsqgl_iters(map_res, params)
            except Exception:         # :1584-1587 "if the resquiggle read fails for any reason"
                saved.append(k)
                res = run_rsqgl_iters(map_res, save_params)
        except th.TomboError as e:
            res, err = None, str(e)
        n_pass.append(passes[0])
        errs.append(err)
        out['raw%d__sha' % k] = np.array(hashlib.sha256(np.ascontiguousarray(raw).tobytes()).hexdigest())
        if res is not None:
            out['segs%d' % k] = res.segs.astype(np.int64)
            out['read_start%d' % k] = np.int64(res.read_start_rel_to_raw)
            sv = res.scale_values
            out['sv%d' % k] = np.array([sv.shift, sv.scale, sv.lower_lim, sv.upper_lim], np.float64)
            out['score%d' % k] = np.float64(res.sig_match_score)
            out['norm%d__sha' % k] = np.array(hashlib.sha256(
                np.ascontiguousarray(res.raw_signal, dtype=np.float64).tobytes()).hexdigest())
            out['changed%d' % k] = np.bool_(res.norm_params_changed)
    out['n_passes'] = np.array(n_pass, np.int64)
    out['used_save_params'] = np.array(saved, np.int64)
    out['meta'] = np.array(json.dumps(dict(aln=aln, specs=specs, seed=SEED, max_iters=MAX_ITERS,
                                           outlier_thresh=OUTLIER, errors=errs, seed_base=7000,
                                           samp=name)))
    np.savez_compressed(os.path.join(HERE, 'loop_%s.npz' % name.lower()), **out)
    print(name, 'passes', n_pass, 'save-parameter retries', saved, 'errors', errs,
          'stalls', [out['stall_ints%d' % k].shape[0] for k in range(len(specs))] if rna else '')


if __name__ == '__main__':
    main('DNA')
    main('RNA')
