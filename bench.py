#!/usr/bin/env python
"""bench.py -- resquiggle reads/s of the HIP batch engine (BASELINE.json metric).

A "step" is one pass of the whole hot path (normalise -> event detection -> start discovery ->
adaptive banded DP -> traceback -> skipped-base raw DP -> Theil-Sen rescale -> score) over one
batch of synthetic reads that is already resident in HBM when the timed region starts.
Workload at N=1 is BASELINE.json configs[1]: 10k synthetic 10 kb DNA reads, bandwidth 500.
With N>1 every rank (one process per GPU, no data-path collective: reads are independent)
gets its own batch of the same size (weak scaling); value = reads of all ranks / max-over-ranks
time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--bases B] [--bandwidth W]

Prints ONE JSON line on rank 0.
"""
import os
import sys
import json
import time
import argparse

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec peak (MI355X_MICROARCH.md); ~6300 achievable


def _gen(args):
    n_bases, seed, samp_name = args
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    models = _gen.models if hasattr(_gen, 'models') else {}
    if samp_name not in models:
        models[samp_name] = ts.TomboModel(seq_samp_type=th.seqSampleType(samp_name, samp_name == 'RNA'))
        _gen.models = models
    model = models[samp_name]
    if samp_name == 'RNA':
        # generated in 5'->3' order = what the worker passes after [::-1]; stalls as the worker
        # finds them
        seq, raw, _ = synth.synth_read(model, n_bases, seed, **synth.RNA_SYNTH)
        return ts.encode_seq(seq).copy(), raw, ts.identify_stalls(raw)
    seq, raw, _ = synth.synth_read(model, n_bases, seed, **synth.DNA_SYNTH)
    return ts.encode_seq(seq).copy(), raw, None


def _under_profiler():
    keys = ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB', 'ROCPROFILER_REGISTER_LIBRARY')
    return any('rocprof' in os.environ.get(k, '').lower() for k in keys)


def make_reads(n_reads, n_bases, base_seed, workers, samp_name='DNA'):
    """Synthetic reads (read i: seed base_seed + i).  Worker processes are forked before any
    HIP state exists; under rocprofv3 forked workers deadlock in the tool's signal handler, so
    threads are used there."""
    jobs = [(n_bases, base_seed + i, samp_name) for i in range(n_reads)]
    _gen(jobs[0])
    if workers > 1 and n_reads >= 64 and not _under_profiler():
        import multiprocessing as mp
        with mp.get_context('fork').Pool(workers) as pool:
            res = pool.map(_gen, jobs, chunksize=max(1, n_reads // (workers * 8)))
    elif workers > 1 and n_reads >= 64:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as ex:
            res = list(ex.map(_gen, jobs, chunksize=max(1, n_reads // (workers * 8))))
    else:
        res = [_gen(j) for j in jobs]
    return [r[0] for r in res], [r[1] for r in res], [r[2] for r in res]


def cpu_baseline(seqs, raws, params, model, n_sample, n_bases, samp_name='DNA', stalls=None):
    """the CPU restatement (oracle/, kind "port") timed single-threaded on a bounded sample of
    the same workload -- reported baseline only; the oracle is never on the measured path"""
    import oracle
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    p = oracle.make_params(params)
    o = oracle.make_opts(model.kmer_width, model.central_pos, outlier_thresh=5.0,
                         sig_match_thresh=SIG_MATCH_THRESH[samp_name])
    rng = np.random.RandomState(7)
    n_sample = min(n_sample, len(raws))
    si = [rng.choice(n_bases, 1000, replace=False) if n_bases > 1000 else None
          for _ in range(n_sample)]
    st = stalls if stalls is not None else [None] * len(raws)
    oracle.resquiggle_read(raws[0], seqs[0], model.level_means, model.level_sds, p, o,
                           stall_ints=st[0], samp_ind=si[0])  # page in
    t0 = time.perf_counter()
    ok = 0
    for i in range(n_sample):
        r = oracle.resquiggle_read(raws[i], seqs[i], model.level_means, model.level_sds, p, o,
                                   stall_ints=st[i], samp_ind=si[i])
        ok += r['status'] == 0
    dt = time.perf_counter() - t0
    return n_sample / dt, n_sample, ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--reads', type=int, default=10000, help='reads per GPU per step')
    ap.add_argument('--bases', type=int, default=10000)
    ap.add_argument('--bandwidth', type=int, default=500)
    ap.add_argument('--cpu-sample', type=int, default=150)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--preset', choices=['cfg2', 'cfg3', 'cfg1', 'cfg4'], default=None,
                    help='BASELINE.json configs: cfg2 10kb/W=500 (default), cfg3 10kb/W=300, '
                         'cfg1 2kb/W=100, cfg4 RNA 3kb/W=500')
    a = ap.parse_args()
    samp_name = 'DNA'
    if a.preset == 'cfg3':
        a.bandwidth = 300
    elif a.preset == 'cfg1':
        a.bases, a.bandwidth = 2000, 100
    elif a.preset == 'cfg4':
        samp_name, a.bases, a.bandwidth = 'RNA', 3000, 500

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # synthetic input first: worker processes must be forked before HIP is initialised
    from tombo_amd import _native, tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType(samp_name, samp_name == 'RNA')
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=a.bandwidth)
    if a.bandwidth <= 100:
        params = params._replace(band_bound_thresh=10)  # the default 40 fails every read at W=100
    workers = max(1, min(32, (os.cpu_count() or 8) // max(world, 1)))
    seqs, raws, stalls = make_reads(a.reads, a.bases, 1000003 * (rank + 1), workers, samp_name)
    if samp_name != 'RNA':
        stalls = None
    rng = np.random.RandomState(12345 + rank)
    si = None
    if a.bases > 1000:
        si = np.stack([rng.choice(a.bases, 1000, replace=False) for _ in range(a.reads)])

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    dev = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)

    eng = _native.Engine(dev)
    eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    p = _native.make_params(params)
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    o = _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH[samp_name])
    t_up = time.perf_counter()
    eng.upload(p, o, raws, seqs, samp_ind=si, stall_ints=stalls)   # host -> HBM, outside the timed region
    t_up = time.perf_counter() - t_up
    algo_bytes, dp_cells = eng.stats()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        eng.run()
    barrier()
    t0 = time.perf_counter()
    stage = np.zeros(32)
    for _ in range(a.steps):
        eng.enqueue()
        eng.sync()
        stage += eng.get(_native.GET_KERNEL_MS)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    out = eng.download(want_norm=False)
    if os.environ.get('TBA_DBG_PHASES'):
        # profiling aid: per-read debug counters of a -DTBA_PHASE_DEBUG / -DTBA_SWEEP_STATS build
        # (TBA_EXTRA_HIPCC_FLAGS, see tombo_amd/_native.py), to stderr
        d = eng.get(99)
        print('dbg mean', ' '.join('%.1f' % x for x in d.mean(axis=0)), file=sys.stderr)
        print('dbg median', ' '.join(str(int(x)) for x in np.median(d, axis=0)), file=sys.stderr)
    n_ok = int((out['status'] == 0).sum())
    stage /= max(a.steps, 1)

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = a.reads * world * a.steps / dt
        # dominant kernel: the main banded DP launch (k_dp<CPL>, DP_MAIN), timed with HIP events
        # on the engine's own stream
        i_dp = _native.STAGE_NAMES.index('main_dp')
        dp_ms = float(stage[i_dp])
        achieved = algo_bytes / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0
        # HBM traffic of that kernel: FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes
        # (profiles/r01h_pmc_hbm_traffic.json, bytes per read), scaled to this launch
        traffic = None
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01h_pmc_hbm_traffic.json')) as fp:
                pmc = json.load(fp)
            if a.bases == 10000 and a.bandwidth == 500:
                traffic = float(pmc['k_dp_bytes_per_read']) * a.reads
        except (IOError, KeyError, ValueError):
            pass
        res = {
            'metric': 'resquiggle reads/s (%s, bw=%d)' % (
                '10 kb DNA' if (samp_name, a.bases) == ('DNA', 10000) else
                '%g kb %s' % (a.bases / 1000.0, samp_name), a.bandwidth), 'value': round(value, 2),
            'unit': 'reads/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%d synthetic %d-base %s reads per GPU per step, '
                                   'bandwidth=%d, full resquiggle_read path, inputs resident '
                                   'in HBM' % (a.reads, a.bases, samp_name, a.bandwidth),
                       'reads_per_gpu': a.reads, 'bases': a.bases, 'bandwidth': a.bandwidth,
                       'success_rate': round(n_ok / float(a.reads), 4),
                       'parallelism': 'reads sharded over %d process(es), no collective' % world,
                       'h2d_upload_s': round(t_up, 3),
                       'stage_ms': {k: round(float(v), 3) for k, v in
                                    zip(_native.STAGE_NAMES, stage[:16]) if v > 0}},
            'roofline': {'bound': 'hbm', 'kernel': 'k_dp (main adaptive banded forward pass)',
                         'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic,
                         'algorithmic_bytes_per_launch': algo_bytes,
                         'kernel_ms': round(dp_ms, 3),
                         'dp_cell_updates_per_s': round(dp_cells / (dp_ms * 1e-3), 1)
                         if dp_ms > 0 else None},
        }
        if world == 1 and not a.no_cpu_baseline:
            v, ns, ok = cpu_baseline(seqs, raws, params, model, a.cpu_sample, a.bases, samp_name,
                                     stalls)
            res['cpu_baseline'] = {
                'value': round(v, 3), 'unit': 'reads/s', 'cores': 1, 'kind': 'port',
                'sample': '%d of the same reads through oracle/ (C restatement, 1 thread; '
                          'measured 1.25x faster than the reference Cython in the build '
                          'container)' % ns}
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
