#!/usr/bin/env python
"""bench.py -- resquiggle reads/s of the HIP batch engine (BASELINE.json metric).

A "step" is one pass of the whole hot path (normalise -> event detection -> start discovery ->
adaptive banded DP -> traceback -> skipped-base raw DP -> Theil-Sen rescale -> score) over one
batch of synthetic reads.  Workload at N=1 is BASELINE.json configs[1]: 10k synthetic 10 kb DNA
reads, bandwidth 500.

Two measurements per run, both over the host work queue (tombo_amd/sharding.py: a shared counter
every rank draws batch indices from; reads are independent, no data-path collective):

  value        RESIDENT: every rank's batch already sits in HBM when the timed region starts
               (float64 pA, the reference's in-memory type); K * N passes are drawn from the queue.
  end_to_end   HOST BUFFERS IN -> HOST BUFFERS OUT through the streaming pipeline
               (tombo_amd/streaming.py: n_slots engines per GPU, upload N+1 || compute N ||
               download N-1): int16 DAC samples in page-locked host memory in, 64-byte record +
               int32 boundaries per read out ("compact"), or float64 in / float64 normalised
               signal + int64 boundaries out (--e2e full).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--preset cfg1..cfg4|longtail] ...

With --gpus N > 1 and no torchrun environment the script launches its own N ranks (one process
per GPU, RCCL only for the barrier and the max-over-ranks time).  Prints ONE JSON line on rank 0.
"""
import os
import sys
import json
import time
import socket
import argparse
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec peak (MI355X_MICROARCH.md); ~6300 achievable
DAC_PER_PA, DAC_OFFSET = 1.0 / 0.1709, 10.0  # MinION-like digitisation of the synthetic pA

# stage (engine event bracket) -> the kernel that dominates it
STAGE_KERNEL = {
    'normalize': 'k_normalize', 'cumsum': 'k_cumsum_scores', 'scores': 'k_scores_ttest',
    'peaks': 'k_peaks (+ RNA: stall removal, event scaling, normalisation)',
    'event_means': 'k_event_means', 'ref_levels': 'k_ref_levels', 'start_dp': 'k_dp (start discovery)',
    'start_tb': 'k_dp (start retry) + k_start_tb', 'prep': 'k_prep',
    'main_dp': 'k_dp (main adaptive banded forward pass)', 'main_tb': 'k_main_tb',
    'skip_resolve': 'k_skip_dp', 'theil_sen': 'k_theil_sen', 'rescale_score': 'k_rescale_absz'}


def _gen(args):
    n_bases, seed, samp_name, want_dac = args
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    models = _gen.models if hasattr(_gen, 'models') else {}
    if samp_name not in models:
        models[samp_name] = ts.TomboModel(seq_samp_type=th.seqSampleType(samp_name, samp_name == 'RNA'))
        _gen.models = models
    model = models[samp_name]
    kw = synth.RNA_SYNTH if samp_name == 'RNA' else synth.DNA_SYNTH
    # RNA: generated in 5'->3' order = what the worker passes after [::-1]; stalls as the worker
    # finds them
    seq, raw, _ = synth.synth_read(model, n_bases, seed, **kw)
    dac = np.round(raw * DAC_PER_PA + DAC_OFFSET).astype(np.int16) if want_dac else None
    stalls = stalls_dac = None
    if samp_name == 'RNA':
        stalls = ts.identify_stalls(raw)
        if want_dac:
            stalls_dac = ts.identify_stalls(dac.astype(np.float64))
    return ts.encode_seq(seq).copy(), raw, stalls, dac, stalls_dac


def _under_profiler():
    keys = ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB', 'ROCPROFILER_REGISTER_LIBRARY')
    return any('rocprof' in os.environ.get(k, '').lower() for k in keys)


def make_reads(bases, base_seed, workers, samp_name='DNA', want_dac=False):
    """Synthetic reads (read i: bases[i] bases, seed base_seed + i).  Worker processes are forked
    before any HIP state exists; under rocprofv3 forked workers deadlock in the tool's signal
    handler, so threads are used there."""
    n_reads = len(bases)
    jobs = [(int(bases[i]), base_seed + i, samp_name, want_dac) for i in range(n_reads)]
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
    return [[r[k] for r in res] for k in range(5)]


def longtail_bases(n_reads, seed):
    """long-tailed read lengths: log-normal (median 8 kb, sigma 0.9) clipped to 1-100 kb"""
    rng = np.random.default_rng(seed)
    return np.clip(np.exp(rng.normal(np.log(8000.0), 0.9, n_reads)), 1000, 100000).astype(np.int64)


def cpu_model():
    try:
        with open('/proc/cpuinfo') as fp:
            for line in fp:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except IOError:
        pass
    return 'unknown'


_CPU_CTX = {}


def _cpu_one(i):
    """one read through the oracle (forked worker processes see _CPU_CTX copy-on-write)"""
    import oracle
    c = _CPU_CTX
    r = oracle.resquiggle_read(c['raws'][i], c['seqs'][i], c['means'], c['sds'], c['p'], c['o'],
                               stall_ints=c['st'][i], samp_ind=c['sis'].get(i))
    return r['status'] == 0


def cpu_baseline(seqs, raws, params, model, n_bases, samp_name, stalls, n_single, n_per_core):
    """The CPU restatement (oracle/, kind "port") on bounded samples of the same workload: one
    process, then one process per host core (forked before any HIP state exists; threads under a
    profiler, where forking deadlocks).  Reported baseline only; the oracle is never on the
    measured path."""
    import oracle
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    n1 = min(n_single, len(raws))
    cores = os.cpu_count() or 1
    nall = min(max(n_per_core * cores, cores), len(raws)) if cores > 1 else 0
    rng = np.random.RandomState(7)
    sis = {i: rng.choice(int(n_bases[i]), 1000, replace=False)
           for i in range(max(n1, nall)) if n_bases[i] > 1000}
    _CPU_CTX.update(raws=raws, seqs=seqs, means=model.level_means, sds=model.level_sds,
                    p=oracle.make_params(params),
                    o=oracle.make_opts(model.kmer_width, model.central_pos, outlier_thresh=5.0,
                                       sig_match_thresh=SIG_MATCH_THRESH[samp_name]),
                    st=stalls if stalls is not None else [None] * len(raws), sis=sis)
    _cpu_one(0)  # page in
    t0 = time.perf_counter()
    ok1 = sum(_cpu_one(i) for i in range(n1))
    dt1 = time.perf_counter() - t0
    legs = [dict(value=round(n1 / dt1, 3), unit='reads/s', cores=1, kind='port',
                 sample='%d of the same reads through oracle/ (C restatement, 1 thread), %d ok' % (n1, ok1))]
    if cores > 1:
        if _under_profiler():
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(cores) as ex:
                t0 = time.perf_counter()
                okn = sum(ex.map(_cpu_one, range(nall)))
                dtn = time.perf_counter() - t0
        else:
            import multiprocessing as mp
            with mp.get_context('fork').Pool(cores) as pool:
                pool.map(_cpu_one, range(cores))      # workers up, library paged in
                t0 = time.perf_counter()
                okn = sum(pool.map(_cpu_one, range(nall), chunksize=1))
                dtn = time.perf_counter() - t0
        legs.append(dict(value=round(nall / dtn, 3), unit='reads/s', cores=cores, kind='port',
                         sample='%d of the same reads through oracle/, %d worker processes (one per '
                                'host core incl. SMT), %d ok' % (nall, cores, okn)))
    _CPU_CTX.clear()
    return legs


# ---- PMC traffic (whole pipeline) of the configuration being run ---------------------------
def pmc_child(path):
    """child of measure_pmc_traffic: load the reads the parent saved, one upload + one pass"""
    from tombo_amd import _native, tombo_stats as ts, tombo_helper as th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    d = np.load(path, allow_pickle=False)
    meta = json.loads(str(d['meta']))
    samp = th.seqSampleType(meta['samp'], meta['samp'] == 'RNA')
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=meta['bandwidth'])
    if meta['bandwidth'] <= 100:
        params = params._replace(band_bound_thresh=10)
    eng = _native.Engine(0)
    eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    eng.upload_packed(_native.make_params(params),
                      _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH[meta['samp']]),
                      d['raw'], d['raw_off'], d['seq'], d['seq_off'],
                      samp_ind=d['samp_ind'] if 'samp_ind' in d.files else None,
                      stall_ints=d['stall_ints'] if 'stall_ints' in d.files else None,
                      stall_off=d['stall_off'] if 'stall_off' in d.files else None, wait=True)
    eng.run()
    print('pmc child ok', int((eng.download(want_norm=False)['status'] == 0).sum()))


def measure_pmc_traffic(packed, meta, timeout=150):
    """FETCH_SIZE + WRITE_SIZE of every kernel of one pass over `packed` (a sub-batch of the reads
    being benchmarked), from two separate `rocprofv3 --pmc` passes of a child process -- the two
    counters do not fit one pass on gfx950 (MI355X_MICROARCH.md, HBM section; KiB units; FETCH
    raw and doubled).  Returns (bytes per read raw, bytes per read with FETCH doubled, per-kernel
    dict) or raises."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rocprof):
        raise RuntimeError('rocprofv3 not found')
    tmp = tempfile.mkdtemp(prefix='tba_pmc_', dir='/tmp')
    path = os.path.join(tmp, 'reads.npz')
    np.savez(path, meta=np.array(json.dumps(meta)), **{k: v for k, v in packed.items() if v is not None})
    n_reads = packed['raw_off'].shape[0] - 1
    tot = {}
    env = dict(os.environ, TMPDIR='/tmp')
    try:
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, ctr)
            subprocess.run([rocprof, '--kernel-trace', '--pmc', ctr, '-d', out, '--',
                            sys.executable, os.path.abspath(__file__), '--pmc-child', path],
                           cwd='/tmp', env=env, check=True, timeout=timeout,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dbs = glob.glob(os.path.join(out, '**', '*.db'), recursive=True)
            if not dbs:
                raise RuntimeError('rocprofv3 wrote no database')
            c = sqlite3.connect(dbs[0])
            for name, v in c.execute('select kernel_name, sum(value) from counters_collection '
                                     'where counter_name = ? group by kernel_name', (ctr,)):
                k = name.split('(')[0].replace('void ', '')
                tot.setdefault(k, {})[ctr] = float(v)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    raw = up = 0.0
    kern = {}
    for k, v in tot.items():
        if k.startswith('__amd'):
            continue
        f, w = v.get('FETCH_SIZE', 0.0) * 1024.0, v.get('WRITE_SIZE', 0.0) * 1024.0
        raw += f + w
        up += 2 * f + w
        kern[k] = round((f + w) / n_reads, 1)
    return raw / n_reads, up / n_reads, kern


def pack_lists(raws, seqs, samp_ind, stalls):
    raw_off = np.zeros(len(raws) + 1, np.int64)
    np.cumsum([len(r) for r in raws], out=raw_off[1:])
    seq_off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=seq_off[1:])
    d = dict(raw=np.concatenate(raws), raw_off=raw_off, seq=np.concatenate(seqs), seq_off=seq_off,
             samp_ind=samp_ind, stall_ints=None, stall_off=None)
    if stalls is not None and any(s is not None and len(s) for s in stalls):
        so = np.zeros(len(raws) + 1, np.int64)
        np.cumsum([0 if s is None else len(s) for s in stalls], out=so[1:])
        d['stall_off'] = so
        d['stall_ints'] = np.array([[int(a), int(b)] for s in stalls if s is not None for a, b in s],
                                   dtype=np.int64).reshape(-1, 2)
    return d


# ---- self-launch of N ranks ------------------------------------------------------------------
def self_launch(n):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--reads', type=int, default=10000, help='reads per GPU per step')
    ap.add_argument('--bases', type=int, default=10000)
    ap.add_argument('--bandwidth', type=int, default=500)
    ap.add_argument('--cpu-sample', type=int, default=150, help='reads of the 1-thread CPU leg')
    ap.add_argument('--cpu-per-core', type=int, default=6, help='reads per core of the all-core CPU leg')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--preset', choices=['cfg2', 'cfg3', 'cfg1', 'cfg4', 'longtail'], default=None,
                    help='BASELINE.json configs: cfg2 10kb/W=500 (default), cfg3 10kb/W=300, '
                         'cfg1 2kb/W=100, cfg4 RNA 3kb/W=500; longtail: log-normal 1-100 kb DNA '
                         'reads (median 8 kb), W=500, batches cut by the planner')
    ap.add_argument('--e2e', choices=['compact', 'full', 'none'], default='compact',
                    help='end-to-end (host in -> host out) measurement: int16 in / records + '
                         'int32 boundaries out, float64 in / float64 signal + int64 boundaries '
                         'out, or skipped')
    ap.add_argument('--stream-batch', type=int, default=None,
                    help='reads per streamed batch (default: 10000 compact, 5000 full: its float64 outputs are page-locked per slot)')
    ap.add_argument('--slots', type=int, default=3, help='engine slots per GPU of the streaming pipeline')
    ap.add_argument('--resident-split', type=int, default=1,
                    help='resident phase: cut the batch into this many sub-batches, each on its own '
                         'engine / stream (kernels of different sub-batches overlap)')
    ap.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 counter passes behind roofline.traffic')
    ap.add_argument('--pmc-reads', type=int, default=1024, help='reads of the counter passes')
    ap.add_argument('--pmc-child', default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.pmc_child:
        return pmc_child(a.pmc_child)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(a.gpus)
    samp_name = 'DNA'
    if a.preset == 'cfg3':
        a.bandwidth = 300
    elif a.preset == 'cfg1':
        a.bases, a.bandwidth = 2000, 100
    elif a.preset == 'cfg4':
        samp_name, a.bases, a.bandwidth = 'RNA', 3000, 500
    longtail = a.preset == 'longtail'
    if longtail and a.reads == 10000:
        a.reads = 16000   # enough work per pass to hide the serial time of a 100 kb read

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # stdout carries exactly one line, the JSON of rank 0: library chatter written to fd 1 (gloo /
    # RCCL banners, HIP warnings) is sent to stderr for the rest of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # synthetic input first: worker processes must be forked before HIP is initialised
    from tombo_amd import _native, planner, sharding, streaming, tombo_stats as ts, tombo_helper as th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    samp = th.seqSampleType(samp_name, samp_name == 'RNA')
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=a.bandwidth)
    if a.bandwidth <= 100:
        params = params._replace(band_bound_thresh=10)  # the default 40 fails every read at W=100
    workers = max(1, min(32, (os.cpu_count() or 8) // max(world, 1)))
    seed0 = 1000003 * (rank + 1)
    bases = longtail_bases(a.reads, seed0) if longtail else np.full(a.reads, a.bases, np.int64)
    want_dac = a.e2e == 'compact'
    t_gen = time.perf_counter()
    seqs, raws, stalls, dacs, stalls_dac = make_reads(bases, seed0, workers, samp_name, want_dac)
    t_gen = time.perf_counter() - t_gen
    if samp_name != 'RNA':
        stalls = stalls_dac = None
    rng = np.random.RandomState(12345 + rank)
    si = np.zeros((a.reads, 1000), np.int64)
    for i in range(a.reads):
        if bases[i] > 1000:
            si[i] = rng.choice(int(bases[i]), 1000, replace=False)
    if not (bases > 1000).any():
        si = None
    n_raw = np.array([len(r) for r in raws], np.int64)
    seq_len = np.array([len(s) for s in seqs], np.int64)

    cpu_legs = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu_legs = cpu_baseline(seqs, raws, params, model, bases, samp_name, stalls, a.cpu_sample,
                                a.cpu_per_core)

    import torch
    dist = None
    ndev = max(torch.cuda.device_count(), 1)
    dev = local_rank % ndev
    torch.cuda.set_device(dev)
    red_dev = 'cuda'
    if world > 1:
        import torch.distributed as dist
        if ndev >= world:
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev))
        else:
            # fewer GPUs than ranks (a development box): the ranks share devices, which RCCL
            # refuses; the barrier / max-over-ranks then go over gloo
            dist.init_process_group('gloo')
            red_dev = 'cpu'

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce(x, op):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=op)
        return float(t.item())

    def max_over_ranks(x):
        return reduce(x, dist.ReduceOp.MAX) if dist is not None else x

    def sum_over_ranks(x):
        return reduce(x, dist.ReduceOp.SUM) if dist is not None else x

    p = _native.make_params(params)
    o = _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH[samp_name])

    # ---- phase 1: resident ---------------------------------------------------------------
    # one engine per planned batch (the uniform presets are a single batch); a pass = every
    # batch's kernel sequence, each on its own stream
    probe = _native.Engine(dev)
    probe.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    free_b, total_b = probe.device_mem()
    if longtail:
        # several batches resident at once, each on its own stream: the one-wave-per-read kernels
        # of the batch holding the 100 kb reads run for ~0.2 s, the other batches fill the machine
        tot = planner.exact_bytes(n_raw, seq_len, p, o, model.kmer_width)
        plan = planner.plan_batches(n_raw, seq_len, p, o, model.kmer_width,
                                    min(0.2 * free_b, max(tot / 6.0, 2e9)))
    else:
        plan = [x for x in np.array_split(np.arange(a.reads), max(1, a.resident_split)) if len(x)]
    engines = [probe] + [_native.Engine(dev) for _ in plan[1:]]
    t_up = time.perf_counter()
    up_bytes = 0
    for eng, idx in zip(engines, plan):
        eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
        eng.upload(p, o, [raws[i] for i in idx], [seqs[i] for i in idx],
                   samp_ind=None if si is None else si[idx],
                   stall_ints=None if stalls is None else [stalls[i] for i in idx])
        up_bytes += int(n_raw[idx].sum()) * 8
    t_up = time.perf_counter() - t_up
    algo_bytes = dp_cells = 0.0
    for eng in engines:
        ab, dc = eng.stats()
        algo_bytes += ab
        dp_cells += dc

    def one_pass():
        for eng in engines:
            eng.enqueue()
        for eng in engines:
            eng.sync()

    for _ in range(a.warmup):
        one_pass()
    queue = sharding.BatchQueue(a.steps * world)   # constructed by every rank, in the same order
    barrier()
    t0 = time.perf_counter()
    stage = np.zeros(32)
    my_steps = 0
    for _ in queue:
        one_pass()
        for eng in engines:
            stage += eng.get(_native.GET_KERNEL_MS)
        my_steps += 1
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    steps_done = int(round(sum_over_ranks(float(my_steps))))
    n_ok = 0
    for eng in engines:
        n_ok += int((eng.download(want_norm=False)['status'] == 0).sum())
    if os.environ.get('TBA_DBG_PHASES'):
        # profiling aid: per-read debug counters of a -DTBA_PHASE_DEBUG / -DTBA_SWEEP_STATS build
        # (TBA_EXTRA_HIPCC_FLAGS, see tombo_amd/_native.py), to stderr
        d = engines[0].get(_native.GET_DEBUG_COUNTERS)
        print('dbg mean', ' '.join('%.1f' % x for x in d.mean(axis=0)), file=sys.stderr)
        print('dbg median', ' '.join(str(int(x)) for x in np.median(d, axis=0)), file=sys.stderr)
    stage /= max(my_steps, 1)
    for eng in engines:
        eng.close()
    del engines, probe

    # ---- phase 2: end to end through the streaming pipeline ---------------------------------
    e2e = None
    if a.e2e != 'none':
        compact = a.e2e == 'compact'
        if a.stream_batch is None:   # (ranks sharing one device -- a rehearsal -- share its memory too)
            a.stream_batch = 10000 if compact and ndev >= world else 5000
        src = dacs if compact else raws
        src_stalls = stalls_dac if compact else stalls
        if longtail:
            o_s = _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH[samp_name],
                                    skip_norm_out=compact)
            splan = planner.plan_batches(n_raw, seq_len, p, o_s, model.kmer_width, 0.2 * free_b,
                                         np.int16 if compact else np.float64, max_reads=a.stream_batch)
        else:
            splan = [np.arange(s, min(s + a.stream_batch, a.reads)) for s in range(0, a.reads, a.stream_batch)]
        t_pin = time.perf_counter()
        pool = [streaming.ReadBatch.from_lists(
            [src[i] for i in idx], [seqs[i] for i in idx],
            samp_inds=None if si is None else [si[i] for i in idx],
            stalls=None if src_stalls is None else [src_stalls[i] for i in idx], tag=k, pinned=True)
            for k, idx in enumerate(splan)]
        t_pin = time.perf_counter() - t_pin
        pipe = streaming.StreamPipeline(model, params, n_slots=a.slots, device=dev, outlier_thresh=5.0,
                                        seq_samp_type=samp, want_norm=not compact,
                                        segs_dtype=np.int32 if compact else np.int64)
        in_bytes = [b.raw.nbytes + b.seq.nbytes + (0 if b.samp_ind is None else b.samp_ind.nbytes) for b in pool]
        # warm-up: every slot sees the largest batch once (buffers sized, code paged in); the
        # transfer rate of one isolated upload is taken on the way
        big = max(range(len(pool)), key=lambda k: in_bytes[k])
        for _ in range(2 * a.slots):   # both output sets of every slot get their pinned arrays
            pipe.submit(pool[big])
        pipe.flush()
        eng0 = pipe.slots[0].eng
        torch.cuda.synchronize()
        th2d = time.perf_counter()
        eng0.upload_packed(pipe.params, pipe.opts, pool[big].raw, pool[big].raw_off, pool[big].seq,
                           pool[big].seq_off, samp_ind=pool[big].samp_ind, stall_ints=pool[big].stall_ints,
                           stall_off=pool[big].stall_off, wait=True)
        th2d = time.perf_counter() - th2d
        # batches of the whole job: K passes over the pool, but never so few that filling and
        # draining the slots is most of the measurement
        n_stream = max(len(pool) * a.steps, 2 * a.slots + 2) * world
        queue = sharding.BatchQueue(n_stream)
        barrier()
        t0 = time.perf_counter()
        cnt = dict(reads=0, ok=0, out=0, moved=0, submit_s=0.0, nb=0)
        est = np.zeros(32)

        def consume(res):
            cnt['reads'] += res.n
            cnt['nb'] += 1
            est[:] += res.stage_ms
            cnt['ok'] += int((res.results['status'] == 0).sum())
            cnt['out'] += res.results.nbytes + res.segs.nbytes + (0 if res.norm is None else res.norm.nbytes)
        for b in queue:
            k = b % len(pool)
            cnt['moved'] += in_bytes[k]
            ts0 = time.perf_counter()
            done = pipe.submit(pool[k])
            cnt['submit_s'] += time.perf_counter() - ts0
            if done is not None:
                consume(done)
        for done in pipe.flush():
            consume(done)
        barrier()
        dt_e = max_over_ranks(time.perf_counter() - t0)
        tot_reads = sum_over_ranks(float(cnt['reads']))
        e2e = {
            'value': round(tot_reads / dt_e, 2), 'unit': 'reads/s', 'mode': a.e2e,
            'what': ('int16 DAC samples + sequence codes + Theil-Sen subsamples in page-locked host '
                     'memory -> 64-byte record + int32 base boundaries per read in page-locked host '
                     'memory' if compact else
                     'float64 samples + sequence codes + subsamples in page-locked host memory -> '
                     'record + int64 boundaries + float64 normalised signal in page-locked host memory'),
            'slots_per_gpu': a.slots, 'reads_per_batch': int(np.mean([b.n for b in pool])),
            'batches': n_stream, 'reads': int(tot_reads), 'seconds': round(dt_e, 4),
            'success_rate': round(sum_over_ranks(float(cnt['ok'])) / max(tot_reads, 1), 4),
            'in_GB_per_10k_reads': round(cnt['moved'] / max(cnt['reads'], 1) * 1e4 / 1e9, 3),
            'out_GB_per_10k_reads': round(cnt['out'] / max(cnt['reads'], 1) * 1e4 / 1e9, 3),
            'h2d_GBps': round(in_bytes[big] / th2d / 1e9, 2),
            'host_s_in_submit': round(cnt['submit_s'], 4),
            'stage_ms_per_batch': {k: round(float(v) / max(cnt['nb'], 1), 3) for k, v in
                                   zip(_native.STAGE_NAMES, est[:16]) if v > 0},
            'pinned_pool_build_s': round(t_pin, 2)}
        pipe.close()

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = a.reads * steps_done / dt
        tot_bases = float(bases.sum())
        # dominant kernel of THIS configuration: the longest stage of the engine's event brackets
        # (HIP events on the engine's own stream)
        names = _native.STAGE_NAMES
        i_dom = int(np.argmax(stage[:14]))
        dom_ms = float(stage[i_dom])
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        i_dp = names.index('main_dp')
        dp_ms = float(stage[i_dp])
        traffic = traffic_up = None
        traffic_note = 'not measured (--no-pmc, N > 1, or already under a profiler)'
        traffic_kernels = None
        if world == 1 and not a.no_pmc and not _under_profiler():
            try:
                sub = np.arange(min(a.pmc_reads, a.reads))
                packed = pack_lists([raws[i] for i in sub], [seqs[i] for i in sub],
                                    None if si is None else si[sub],
                                    None if stalls is None else [stalls[i] for i in sub])
                per_raw, per_up, traffic_kernels = measure_pmc_traffic(
                    packed, dict(samp=samp_name, bandwidth=a.bandwidth))
                # the counter passes run a sub-batch of the same reads; traffic scales with the
                # samples / bases processed
                scale = float(n_raw.sum()) / float(n_raw[sub].sum())
                traffic = per_raw * len(sub) * scale
                traffic_up = per_up * len(sub) * scale
                traffic_note = ('whole pipeline, FETCH_SIZE + WRITE_SIZE (KiB) from two rocprofv3 --pmc '
                                'passes of a %d-read sub-batch of this run, scaled by samples to the '
                                'launch; traffic_fetch_doubled applies the gfx950 wide-read correction '
                                'to every read (upper bound)' % len(sub))
            except Exception as e:  # counters are evidence, not the metric: never fail the bench
                traffic_note = 'counter passes failed: %s' % (str(e)[:200],)
        res = {
            'metric': 'resquiggle reads/s (%s, bw=%d)' % (
                '10 kb DNA' if (samp_name, a.bases, longtail) == ('DNA', 10000, False) else
                'long-tailed 1-100 kb DNA' if longtail else
                '%g kb %s' % (a.bases / 1000.0, samp_name), a.bandwidth), 'value': round(value, 2),
            'unit': 'reads/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%d synthetic %s %s reads per GPU per step, bandwidth=%d, full '
                                   'resquiggle_read path, float64 inputs resident in HBM; %d passes '
                                   'drawn from the host work queue by %d rank(s)' % (
                                       a.reads, 'long-tailed (1-100 kb)' if longtail else '%d-base' % a.bases,
                                       samp_name, a.bandwidth, a.steps * world, world),
                       'reads_per_gpu': a.reads, 'bases': int(a.bases) if not longtail else None,
                       'mean_bases': round(tot_bases / a.reads, 1), 'bandwidth': a.bandwidth,
                       'resident_batches_per_gpu': len(plan),
                       'bases_per_s': round(tot_bases * steps_done / dt, 1),
                       'success_rate': round(n_ok / float(a.reads), 4),
                       'parallelism': 'reads sharded over %d process(es) through a shared batch '
                                      'counter, no collective' % world,
                       'h2d_upload_s': round(t_up, 3),
                       'h2d_upload_GBps_pageable': round(up_bytes / t_up / 1e9, 2),
                       'synth_generation_s': round(t_gen, 1),
                       'stage_ms': {k: round(float(v), 3) for k, v in zip(names, stage[:16]) if v > 0}},
            'end_to_end': e2e,
            'roofline': {'bound': 'hbm', 'kernel': STAGE_KERNEL.get(names[i_dom], names[i_dom]),
                         'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 5),
                         'traffic': traffic, 'traffic_fetch_doubled': traffic_up,
                         'traffic_scope': traffic_note,
                         'traffic_over_algorithmic': None if traffic is None else round(traffic / algo_bytes, 3),
                         'traffic_bytes_per_read_by_kernel': traffic_kernels,
                         'algorithmic_bytes_per_launch': algo_bytes,
                         'kernel_ms': round(dom_ms, 3),
                         'pipeline_hbm_frac': round(algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         'dp_cell_updates_per_s': round(dp_cells / (dp_ms * 1e-3), 1) if dp_ms > 0 else None},
        }
        # the banded DP is bound by f64 VALU issue, not HBM: instructions per launch come from the
        # committed SQ counter profile of this configuration (profiles/), time is measured here
        try:
            with open(os.path.join(ROOT, 'profiles', 'valu_counts.json')) as fp:
                vc = json.load(fp)
            key = '%s_b%d_w%d' % (samp_name, a.bases, a.bandwidth)
            if key in vc and not longtail and dp_ms > 0:
                insts = float(vc[key]['k_dp_valu_wave_insts_per_read']) * a.reads
                peak = 1024 * float(vc['clock_ghz']) * 1e9 / 4.0  # 1024 SIMDs, one f64 VALU op / 4 cycles
                res['roofline_valu'] = {
                    'bound': 'valu_f64_issue', 'kernel': 'k_dp (main adaptive banded forward pass)',
                    'achieved': round(insts / (dp_ms * 1e-3) / 1e9, 2), 'peak': round(peak / 1e9, 2),
                    'unit': 'G wave-instr/s', 'frac': round(insts / (dp_ms * 1e-3) / peak, 4),
                    'valu_insts_per_dp_row': vc[key].get('valu_insts_per_row'),
                    'kernel_ms': round(dp_ms, 3), 'source': vc[key].get('source')}
        except (IOError, KeyError, ValueError):
            pass
        if cpu_legs:
            res['cpu_baseline'] = dict(cpu_legs[0], cpu=cpu_model(),
                                       all_cores=cpu_legs[1] if len(cpu_legs) > 1 else None,
                                       reference_cython='see BASELINE.md (tools/time_reference.py: the '
                                                        'reference Cython path vs this port on the build host)')
        os.write(json_fd, (json.dumps(res) + '\n').encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
