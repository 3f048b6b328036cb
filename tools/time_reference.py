"""Time the REFERENCE's Cython resquiggle path next to the CPU port (build container only).

    python tools/time_reference.py [--reads 100] [--bases 10000] [--bandwidth 500]

`tombo.resquiggle.resquiggle_read` of the live reference (tools/ref_oracle.py builds its two
Cython modules from /root/reference) and `oracle.resquiggle_read` (the C restatement bench.py
times on the GPU box as `cpu_baseline`, kind "port") on the same synthetic reads, one process, one
thread each.  Prints per-read median and p10-p90 and the ratio t_port / t_cython that converts the
port's reads/s on another host into an estimate of the reference's.  BASELINE.md quotes the
output of this script; bench.py does not hard-code it.
"""
import os
import sys
import json
import time
import argparse
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def cpu_model():
    with open('/proc/cpuinfo') as fp:
        for line in fp:
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    return 'unknown'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=100)
    ap.add_argument('--bases', type=int, default=10000)
    ap.add_argument('--bandwidth', type=int, default=500)
    ap.add_argument('--rna', action='store_true')
    a = ap.parse_args()
    import ref_oracle
    import oracle
    from tombo_amd import synth, tombo_stats as my_ts, tombo_helper as my_th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    rq, ts, th = ref_oracle.load()
    sn = 'RNA' if a.rna else 'DNA'
    samp = th.seqSampleType(sn, False)
    my_model = my_ts.TomboModel(seq_samp_type=my_th.seqSampleType(sn, False))
    kmers = sorted(my_model.means.keys())
    std_ref = ts.TomboModel(kmer_ref=[(k, my_model.means[k], my_model.sds[k]) for k in kmers],
                            central_pos=my_model.central_pos, seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=a.bandwidth)
    my_params = my_ts.load_resquiggle_parameters(my_th.seqSampleType(sn, False))._replace(
        bandwidth=a.bandwidth)
    if a.bandwidth <= 100:
        params = params._replace(band_bound_thresh=10)
        my_params = my_params._replace(band_bound_thresh=10)
    kw = dict(synth.RNA_SYNTH if a.rna else synth.DNA_SYNTH)
    p = oracle.make_params(my_params)
    o = oracle.make_opts(my_model.kmer_width, my_model.central_pos, outlier_thresh=5.0,
                         sig_match_thresh=SIG_MATCH_THRESH[sn])
    t_ref, t_port, same = [], [], 0
    for i in range(a.reads + 1):
        seq, raw, _ = synth.synth_read(my_model, a.bases, 500000 + i, **kw)
        stalls = ts.identify_stalls(raw, rq.DEFAULT_STALL_PARAMS) if a.rna else None
        mr = th.resquiggleResults(
            align_info=th.alignInfo('r', 'BaseCalled_template', 0, 0, 0, 0, a.bases, 0),
            genome_loc=th.genomeLocation(0, '+', 'synth'), genome_seq=seq, mean_q_score=10.0,
            raw_signal=raw, stall_ints=stalls)
        np.random.seed(i)
        t0 = time.perf_counter()
        try:
            res = rq.resquiggle_read(mr, std_ref, params, 5.0, seq_samp_type=samp)
        except th.TomboError:
            res = None
        t1 = time.perf_counter()
        np.random.seed(i)
        si = np.random.choice(a.bases, 1000, replace=False) if a.bases > 1000 else None
        codes = my_ts.encode_seq(seq)
        t2 = time.perf_counter()
        r = oracle.resquiggle_read(raw, codes, my_model.level_means, my_model.level_sds, p, o,
                                   stall_ints=stalls, samp_ind=si)
        t3 = time.perf_counter()
        if i == 0:
            continue   # page-in
        t_ref.append(t1 - t0)
        t_port.append(t3 - t2)
        same += int(res is not None and r['status'] == 0 and np.array_equal(res.segs, r['segs']))
    q = lambda v: [float(np.percentile(v, x)) for x in (10, 50, 90)]
    out = dict(cpu=cpu_model(), reads=a.reads, bases=a.bases, bandwidth=a.bandwidth, sample=sn,
               reference_cython_s_per_read=dict(zip(('p10', 'median', 'p90'), q(t_ref))),
               port_s_per_read=dict(zip(('p10', 'median', 'p90'), q(t_port))),
               reference_reads_per_s=1.0 / float(np.median(t_ref)),
               port_reads_per_s=1.0 / float(np.median(t_port)),
               t_port_over_t_cython=float(np.median(t_port) / np.median(t_ref)),
               identical_segs=same)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
