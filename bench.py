#!/usr/bin/env python
"""bench.py -- resquiggle reads/s of the HIP batch engine (BASELINE.json metric).

A "step" is one pass of the whole hot path (stall detection for RNA -> normalise -> event
detection -> start discovery -> adaptive banded DP -> traceback -> skipped-base raw DP ->
Theil-Sen rescale -> score) over one batch of synthetic reads.  Workload at N=1 is BASELINE.json
configs[1]: 10k synthetic 10 kb DNA reads, bandwidth 500.

Measurements of one run, all drawn through the host work queue (tombo_amd/sharding.py: a shared
counter every rank draws batch indices from; reads are independent, no data-path collective, and
the control plane -- barrier, two scalar reductions, the per-rank report -- is gloo on every
path: no RCCL anywhere, as BASELINE.json's north_star states):

  value        RESIDENT: every rank's batch already sits in HBM when the timed region starts
               (float64 pA, the reference's in-memory type); K * N passes are drawn from the queue.
  end_to_end   READS IN -> RESULTS OUT: per-read int16 DAC arrays (as the FAST5 `Signal` dataset
               holds them; RNA in acquisition order) and sequence strings are packed into
               page-locked CSR staging by native threads (tba_pack_reads), uploaded, run (RNA: flip
               + stall detection on the device; Theil-Sen subsample drawn on the device) and come
               back as a 64-byte record + int32 boundaries per read, through the streaming
               pipeline (tombo_amd/streaming.py: n_slots engines per GPU, upload N+1 || compute
               N || download N-1).  --e2e full: float64 in / float64 signal + int64 boundaries out.
  api          (N = 1) the drop-in Python API itself: resquiggle_batch(list of map_res) reads/s
               and the latency of a batch-of-one resquiggle_read.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--preset cfg1..cfg4|longtail] ...

With --gpus N > 1 and no torchrun environment the script launches its own N ranks (one process
per GPU).  Prints ONE JSON line on rank 0.
"""
import os
import zlib
import gc
import sys
import json
import time
import argparse
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec peak (MI355X_MICROARCH.md); ~6300 achievable
DAC_PER_PA, DAC_OFFSET = 1.0 / 0.1709, 10.0  # MinION-like digitisation of the synthetic pA

# roofline: the dominant stage GROUP of the configuration (engine event brackets) and the kernels
# inside it
# stage brackets (HIP events on the engine's stream) whose time is ONE kernel's, or nearly: the dominant one is the
# line's `roofline.kernel` (rocprof's per-kernel averages of the same runs: profiles/r06_kernel_stats_*.txt)
STAGE_GROUPS = [
    ('main_dp', ['main_dp'], 'k_dp (main adaptive banded forward pass)'),
    ('cumsum', ['cumsum'], 'k_detect<2> + k_pick (DNA event detection; k_detect is 82 % of the bracket)'),
    ('scores', ['scores'], 'k_detect_tt<5, 12> (RNA t-test event detection: the bracket is this kernel)'),
    ('peaks', ['peaks'], 'k_pick + k_remove_stalls + event scaling (RNA)'),
    ('normalize', ['normalize'], 'k_normalize'), ('stalls', ['stalls'], 'k_cumsum_scores<raw> + k_stall_metric'),
    ('event_means', ['event_means'], 'k_event_means'), ('start', ['start_dp', 'start_tb'], 'k_dp (start discovery) + k_start_tb'),
    ('main_tb', ['main_tb'], 'k_main_tb_par + k_tb_par_verify + k_tb_gather'), ('skip_resolve', ['skip_resolve'], 'k_skip_plan + k_skip_dp'),
    ('theil_sen', ['theil_sen'], 'k_theil_sen'), ('rescale_score', ['rescale_score'], 'k_rescale_absz')]


def _gen(args):
    n_bases, seed, samp_name, want_dac = args
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    models = _gen.models if hasattr(_gen, 'models') else {}
    if samp_name not in models:
        models[samp_name] = ts.TomboModel(seq_samp_type=th.seqSampleType(samp_name, samp_name == 'RNA'))
        _gen.models = models
    model = models[samp_name]
    kw = synth.RNA_SYNTH if samp_name == 'RNA' else synth.DNA_SYNTH
    # RNA: generated in 5'->3' order = what the worker holds after [::-1]; the
    # acquisition-order array (the file's) is its flip
    seq, raw, _ = synth.synth_read(model, n_bases, seed, **kw)
    dac = None
    if want_dac:
        dac = np.round(raw * DAC_PER_PA + DAC_OFFSET).astype(np.int16)
        if samp_name == 'RNA':
            dac = np.ascontiguousarray(dac[::-1])
    return seq, raw, dac


def _under_profiler():
    keys = ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB', 'ROCPROFILER_REGISTER_LIBRARY')
    return any('rocprof' in os.environ.get(k, '').lower() for k in keys)


def _gen_chunk(args):
    """a worker's share of make_reads: the reads of one chunk, their sample arrays concatenated into
    files of a memory-backed directory (the parent maps them: pickling 9 GB of samples through the
    pool's pipes -- one parent thread unpickles -- was most of the 11.5 s the set-up spent here)"""
    jobs, k, shm_dir = args
    res = [_gen(j) for j in jobs]
    out = dict(k=k, seqs=[r[0] for r in res], n=[len(r[1]) for r in res])
    for key, col in (('raw', 1), ('dac', 2)):
        if res[0][col] is None:
            out[key] = None
            continue
        path = os.path.join(shm_dir, 'chunk%d_%s.npy' % (k, key))
        np.save(path, np.concatenate([r[col] for r in res]))
        out[key] = path
    return out


def make_reads(bases, base_seed, workers, samp_name='DNA', want_dac=False):
    """Synthetic reads (read i: bases[i] bases, seed base_seed + i) -> (seq strings, float64 pA in
    5'->3' order, int16 DAC in acquisition order or None).  Worker processes are forked before any
    HIP state exists and hand their samples over through files in /dev/shm (pipes when that is
    missing or small); under rocprofv3 forked workers deadlock in the tool's signal handler, so
    threads are used there."""
    n_reads = len(bases)
    jobs = [(int(bases[i]), base_seed + i, samp_name, want_dac) for i in range(n_reads)]
    _gen(jobs[0])
    if workers > 1 and n_reads >= 64 and not _under_profiler():
        import multiprocessing as mp
        import shutil
        import tempfile
        need = float(np.sum(bases)) * 9.5 * (8 + 2) * 1.3     # samples x bytes, with margin
        shm_dir = None
        try:
            st = os.statvfs('/dev/shm')
            if st.f_bavail * st.f_frsize > need:
                shm_dir = tempfile.mkdtemp(prefix='tba_bench_', dir='/dev/shm')
        except OSError:
            shm_dir = None
        with mp.get_context('fork').Pool(workers) as pool:
            if shm_dir is None:
                res = pool.map(_gen, jobs, chunksize=max(1, n_reads // (workers * 8)))
            else:
                try:
                    step = max(1, n_reads // (workers * 4))
                    chunks = [(jobs[a:a + step], a, shm_dir) for a in range(0, n_reads, step)]
                    res = []
                    for c in sorted(pool.imap_unordered(_gen_chunk, chunks), key=lambda c: c['k']):
                        off = np.concatenate([[0], np.cumsum(c['n'])])
                        raw = np.load(c['raw'])
                        dac = None if c['dac'] is None else np.load(c['dac'])
                        for i, sq in enumerate(c['seqs']):
                            sl = slice(off[i], off[i + 1])
                            res.append((sq, raw[sl], None if dac is None else dac[sl]))
                finally:
                    shutil.rmtree(shm_dir, ignore_errors=True)
    elif workers > 1 and n_reads >= 64:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as ex:
            res = list(ex.map(_gen, jobs, chunksize=max(1, n_reads // (workers * 8))))
    else:
        res = [_gen(j) for j in jobs]
    return [[r[k] for r in res] for k in range(3)]


def longtail_bases(n_reads, seed, max_bases=200000):
    """long-tailed read lengths: log-normal (median 8 kb, sigma 0.9) clipped to 1-200 kb"""
    rng = np.random.default_rng(seed)
    return np.clip(np.exp(rng.normal(np.log(8000.0), 0.9, n_reads)), 1000, max_bases).astype(np.int64)


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


def cpu_baseline(seqs, raws, params, model, n_bases, samp_name, n_single, n_per_core):
    """The CPU restatement (oracle/, kind "port") on bounded samples of the same workload: one
    process, then one process per host core (forked before any HIP state exists; threads under a
    profiler, where forking deadlocks).  Reported baseline only; the oracle is never on the
    measured path.  (RNA: the stall intervals the worker would pass in are computed by the
    restatement outside the clock -- the reference's resquiggle_read does not include them.)"""
    import oracle
    from tombo_amd import tombo_stats as ts
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    n1 = min(n_single, len(raws))
    cores = os.cpu_count() or 1
    nall = min(max(n_per_core * cores, cores), len(raws)) if cores > 1 and n_per_core > 0 else 0
    nmax = max(n1, nall)
    rng = np.random.RandomState(7)
    sis = {i: rng.choice(int(n_bases[i]), 1000, replace=False)
           for i in range(nmax) if n_bases[i] > 1000}
    st = [oracle.identify_stalls(raws[i]) if samp_name == 'RNA' else None for i in range(nmax)]
    _CPU_CTX.update(raws=raws, seqs=[ts.encode_seq(s) for s in seqs[:nmax]], means=model.level_means,
                    sds=model.level_sds, p=oracle.make_params(params),
                    o=oracle.make_opts(model.kmer_width, model.central_pos, outlier_thresh=5.0,
                                       sig_match_thresh=SIG_MATCH_THRESH[samp_name]), st=st, sis=sis)
    _cpu_one(0)  # page in
    t0 = time.perf_counter()
    ok1 = sum(_cpu_one(i) for i in range(n1))
    dt1 = time.perf_counter() - t0
    legs = [dict(value=round(n1 / dt1, 3), unit='reads/s', cores=1, kind='port',
                 sample='%d of the same reads through oracle/ (C restatement, 1 thread), %d ok' % (n1, ok1))]
    if nall > 0:
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


def cpu_child(job):
    """The CPU legs in a process of their own: it regenerates the reads it needs (same seeds, so the
    same reads as the parent's first ones) instead of inheriting the parent's gigabytes -- forking
    one worker per host core out of a process that holds the whole synthetic batch cost more than
    the legs themselves (56 s of set-up for 14 s of measurement on the 256-thread host)."""
    j = json.loads(job)
    from tombo_amd import tombo_stats as ts, tombo_helper as th
    samp = th.seqSampleType(j['samp'], j['samp'] == 'RNA')
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=j['bandwidth'])
    if j['bandwidth'] <= 100:
        params = params._replace(band_bound_thresh=10)
    bases = np.array(j['bases'], np.int64)
    workers = max(1, min(32, os.cpu_count() or 8))
    seqs, raws, _ = make_reads(bases, j['seed0'], workers, j['samp'], False)
    legs = cpu_baseline(seqs, raws, params, model, bases, j['samp'], j['n_single'], j['n_per_core'])
    print(json.dumps(legs))


# ---- PMC traffic (whole pipeline) of the configuration being run ---------------------------
def pmc_child(path):
    """child of measure_pmc_traffic: load the reads the parent saved, one upload + one pass"""
    from tombo_amd import _native, tombo_stats as ts, tombo_helper as th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH, STALL_PARAMS
    d = np.load(path, allow_pickle=False)
    meta = json.loads(str(d['meta']))
    samp = th.seqSampleType(meta['samp'], meta['samp'] == 'RNA')
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=meta['bandwidth'])
    if meta['bandwidth'] <= 100:
        params = params._replace(band_bound_thresh=10)
    eng = _native.Engine(0)
    # the sub-batch takes the dispatch forms of the TIMED batch (event detection and traceback switch
    # kernels on the read count: a 2 048-read sub-batch of a 100-read run must not count k_detect's bytes,
    # nor a 1 024-read sub-batch of a 10 000-read run those of k_peaks)
    sb, tw = eng.get_dispatch()
    eng.set_dispatch(0 if meta['timed_reads'] > sb else 1 << 40, 0 if meta['timed_reads'] > tw else 1 << 40)
    eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    eng.upload_packed(_native.make_params(params),
                      _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH[meta['samp']],
                                        stall_params=th.stallParams(**STALL_PARAMS) if meta['samp'] == 'RNA' else None),
                      d['raw'], d['raw_off'], d['seq'], d['seq_off'],
                      samp_ind=d['samp_ind'] if 'samp_ind' in d.files else None, wait=True)
    eng.run()
    ok = eng.download(want_norm=False)['status'] == 0
    ed, tb = eng.get(_native.GET_ED_FORM)[ok], eng.get(_native.GET_TB_FORM)[ok]
    with open(path + '.forms.json', 'w') as fp:   # which kernels the counted pass really ran, per read
        json.dump({'ed_form': {str(k): int(v) for k, v in zip(*np.unique(ed, return_counts=True))},
                   'tb_form': {str(k): int(v) for k, v in zip(*np.unique(tb, return_counts=True))}}, fp)
    print('pmc child ok', int(ok.sum()))


def measure_pmc_traffic(packed, meta, device=0, timeout=150):
    """FETCH_SIZE + WRITE_SIZE of every kernel of one pass over `packed` (a sub-batch of the reads
    being benchmarked), from two separate `rocprofv3 --pmc` passes of a child process -- the two
    counters do not fit one pass on gfx950 (MI355X_MICROARCH.md, HBM section; KiB units; FETCH
    raw and doubled).  The child sees only `device`.  Returns (bytes per read raw, bytes per read
    with FETCH doubled, per-kernel dict, the dispatch forms the counted pass took) or raises."""
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
    vis = os.environ.get('HIP_VISIBLE_DEVICES')
    phys = vis.split(',')[device] if vis else str(device)
    env = dict(os.environ, TMPDIR='/tmp', HIP_VISIBLE_DEVICES=phys)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'TBA_STORE_PORT'):
        env.pop(k, None)
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
        with open(path + '.forms.json') as fp:
            forms = json.load(fp)
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
    return raw / n_reads, up / n_reads, kern, forms


def pack_lists(raws, seqs, samp_ind):
    from tombo_amd import tombo_stats as ts
    raw_off = np.zeros(len(raws) + 1, np.int64)
    np.cumsum([len(r) for r in raws], out=raw_off[1:])
    codes = [ts.encode_seq(s) for s in seqs]
    seq_off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(s) for s in codes], out=seq_off[1:])
    return dict(raw=np.concatenate(raws), raw_off=raw_off, seq=np.concatenate(codes), seq_off=seq_off,
                samp_ind=samp_ind)


# ---- self-launch of N ranks ------------------------------------------------------------------
def self_launch(n):
    """N ranks of this script, one per GPU.  The rendezvous store is created HERE on a port the
    kernel picks (no bind / close / rebind race) and stays up until the ranks are done; they join
    it as clients (TBA_STORE_PORT)."""
    from datetime import timedelta
    import torch.distributed as dist
    store = dist.TCPStore('127.0.0.1', 0, n, is_master=True, timeout=timedelta(seconds=1800),
                          wait_for_workers=False)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', TBA_STORE_PORT=str(store.port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    del store
    sys.exit(rc)


def early_store(world):
    """A client of the launcher's rendezvous store (torchrun's agent store, or self_launch's), before
    the process group exists: used for ONE flag -- rank 0 raises 'tba_cpu_legs_done' when its CPU
    baseline legs are over, the other ranks wait for it before they start anything that loads the
    host (read synthesis workers, generators, pack threads).  None when there is no such store yet
    (a bare env:// launch whose rank 0 creates the store inside init_process_group): no wait then."""
    try:
        import torch.distributed as dist
        from datetime import timedelta
        if 'TBA_STORE_PORT' in os.environ:
            addr, port = '127.0.0.1', int(os.environ['TBA_STORE_PORT'])
        elif os.environ.get('TORCHELASTIC_USE_AGENT_STORE', '').lower() in ('1', 'true'):
            addr, port = os.environ.get('MASTER_ADDR', '127.0.0.1'), int(os.environ['MASTER_PORT'])
        else:
            return None
        return dist.TCPStore(addr, port, world, is_master=False, timeout=timedelta(seconds=1800),
                             wait_for_workers=False)
    except Exception:
        return None


def init_control_plane(rank, world):
    """gloo process group for the barrier / reductions / per-rank report (never RCCL: ranks of a
    resquiggle job exchange no data).  Under torchrun the env:// rendezvous of the launcher is
    used; ranks launched by self_launch join its store."""
    import torch.distributed as dist
    from datetime import timedelta
    if 'TBA_STORE_PORT' in os.environ:
        store = dist.TCPStore('127.0.0.1', int(os.environ['TBA_STORE_PORT']), world, is_master=False,
                              timeout=timedelta(seconds=1800))
        dist.init_process_group('gloo', store=store, rank=rank, world_size=world,
                                timeout=timedelta(seconds=1800))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=timedelta(seconds=1800))
    return dist


def run_job(ctx, job_reads, batch_reads, n_bases, job_seed):
    """One resquiggle job of `job_reads` DISTINCT synthetic reads (BASELINE.json cfg5): the job is
    cut into batches of `batch_reads`, the ranks draw batch indices from the shared counter
    (sharding.BatchQueue) and every drawn batch is synthesised on the drawing rank's device
    (_native.Synth: read r of the job is a function of (job_seed, r) alone, whichever rank makes it,
    whatever the batch size), copied device to device into a pipeline slot, resquiggled (Theil-Sen
    subsample drawn on the device under a key of the batch index), and its compact results (64-byte
    record + int32 boundaries per read) land in page-locked host memory.  Nothing is synthesised on
    the host and no rank touches a batch it did not draw.  Collective over the ranks."""
    _native, streaming, sharding = ctx['_native'], ctx['streaming'], ctx['sharding']
    from tombo_amd import synth as synth_mod
    model, dev, world, slots = ctx['model'], ctx['dev'], ctx['world'], ctx['slots']
    n_batches = (job_reads + batch_reads - 1) // batch_reads
    pipe = streaming.StreamPipeline(model, ctx['params'], n_slots=slots, device=dev, outlier_thresh=5.0,
                                    seq_samp_type=ctx['samp'], want_norm=False, segs_dtype=np.int32,
                                    subsample_seed=job_seed, in_order=True)
    # a generator's batch must outlive the slot's copy of it: two generators per slot
    gens = [_native.Synth(model, dev) for _ in range(2 * slots + 2)]   # (+ the one being made, + one spare)
    sp = _native.make_synth_params(dac_per_pa=DAC_PER_PA, dac_offset=DAC_OFFSET, **synth_mod.DNA_SYNTH)
    cnt = dict(reads=0, ok=0, nb=0, chk=0, gen_s=0.0, submit_s=0.0, wait_s=0.0, out=0)
    est = np.zeros(32)
    drawn = []
    state = dict(i=0)

    def consume(res, count=True):
        if res is None or not count:
            return
        r = res.results
        ok = r['status'] == 0
        cnt['reads'] += res.n
        cnt['ok'] += int(ok.sum())
        cnt['nb'] += 1
        cnt['out'] += r.nbytes + res.segs.nbytes
        est[:] += res.stage_ms
        # a checksum of the batch that does not depend on who computed it: the records of the reads
        # that succeeded + their last base boundary
        last = np.asarray(res.segs)[np.asarray(res.seg_off[1:]) - 1].astype(np.int64)
        cnt['chk'] += int(((r['read_start_rel_to_raw'] + r['norm_len'] + last) * ok).sum()) % (1 << 40)

    def make(k, first_read, n_k):
        """batch k of the job in device memory (blocks until the generator's kernels have run)"""
        g = gens[state['i'] % len(gens)]
        state['i'] += 1
        t0 = time.perf_counter()
        raw, raw_off, seq, seq_off = g.generate(sp, job_seed, np.full(n_k, n_bases, np.int64),
                                                raw_dtype=np.int16, first_read=first_read)
        b = streaming.ReadBatch(raw, raw_off, seq, seq_off, tag=k)
        b.subsample_seed = (job_seed * 0x9E3779B97F4A7C15 + first_read) & 0xffffffffffffffff
        return b, time.perf_counter() - t0

    def submit(b, dt_gen, count=True):
        t1 = time.perf_counter()
        done = pipe.submit(b)
        if count:
            cnt['gen_s'] += dt_gen
            cnt['submit_s'] += time.perf_counter() - t1
        consume(done, count)

    # warm-up: reads outside the job (behind its last read), every slot and generator sees a full batch
    for w in range(len(gens) + 2):
        submit(*make(-1 - w, job_reads + w * batch_reads, batch_reads), count=False)
    for done in pipe.flush():
        pass
    algo_bytes, dp_cells = pipe.slots[0].eng.stats()
    queue = sharding.BatchQueue(n_batches, key='distinct_read_job')
    # The generator's kernels wait for SIMD slots behind the wavefronts of the batches in flight (a k_dp
    # wavefront lives for tens of milliseconds), so a batch is made one draw ahead on a helper thread
    # while the previous one is submitted: the draw of batch k + 1 happens before batch k is submitted.
    from concurrent.futures import ThreadPoolExecutor
    helper = ThreadPoolExecutor(1)

    def draw(it):
        k = next(it, None)
        if k is None:
            return None
        first = k * batch_reads
        drawn.append((int(k), int(first)))
        return helper.submit(make, k, first, min(batch_reads, job_reads - first))

    ctx['barrier']()
    t0 = time.perf_counter()
    it = iter(queue)
    pending = draw(it)
    while pending is not None:
        tw = time.perf_counter()
        b, dt_gen = pending.result()
        cnt['wait_s'] += time.perf_counter() - tw
        pending = draw(it)
        submit(b, dt_gen)
    for done in pipe.flush():
        consume(done)
    ctx['dev_sync']()
    my_dt = time.perf_counter() - t0
    ctx['barrier']()
    dt = ctx['max_over_ranks'](time.perf_counter() - t0)
    helper.shutdown()
    tot = {k: ctx['sum_over_ranks'](float(cnt[k])) for k in ('reads', 'ok', 'nb', 'chk', 'out')}
    pipe.close()
    for g in gens:
        g.close()
    names = _native.STAGE_NAMES
    nb = max(cnt['nb'], 1)
    report = {
        'what': 'one job of %d distinct synthetic 10 kb DNA reads (read r = f(job seed, r)), %d batches of <= %d '
                'drawn from the host work queue by %d rank(s); each batch synthesised as int16 DAC samples on the '
                'device of the rank that drew it, resquiggled, 64-byte record + int32 boundaries per read '
                'downloaded to page-locked host memory' % (job_reads, n_batches, batch_reads, world),
        'value': round(tot['reads'] / dt, 2), 'unit': 'reads/s', 'job_reads': int(job_reads),
        'reads_done': int(tot['reads']), 'batches': int(tot['nb']), 'seconds': round(dt, 4),
        'success_rate': round(tot['ok'] / max(tot['reads'], 1), 4),
        'checksum_mod_2_40_summed': int(tot['chk']), 'job_seed': int(job_seed),
        'out_GB': round(tot['out'] / 1e9, 3),
        'helper_s_in_generate_rank0': round(cnt['gen_s'], 4), 'host_s_waiting_for_generator_rank0': round(cnt['wait_s'], 4),
        'host_s_in_submit_rank0': round(cnt['submit_s'], 4),
        'stage_ms_per_batch_rank0': {k: round(float(v) / nb, 3) for k, v in zip(names, est[:16]) if v > 0}}
    return dict(report=report, dt=dt, my_dt=my_dt, my_batches=cnt['nb'], my_reads=cnt['reads'], drawn=drawn,
                stage=est / nb, algo_bytes=algo_bytes, dp_cells=dp_cells, n_batches=n_batches, tot=tot)


def finish_cfg5(a, job, ctx, dist, json_fd, cpu_legs, dev_name, ndev, t_start, t_cpu, scaling):
    """the JSON line of --preset cfg5 (rank 0 prints)"""
    _native = ctx['_native']
    rank, world = ctx['rank'], ctx['world']
    rank_rec = dict(rank=rank, device=ctx['dev'], device_name=dev_name, steps=job['my_batches'],
                    job_batches=job['my_batches'], job_reads=job['my_reads'],
                    first_reads_drawn=[f for _, f in job['drawn']],
                    reads_per_s=round(job['my_reads'] / job['my_dt'], 2) if job['my_dt'] > 0 else 0.0,
                    busy_s=round(job['my_dt'], 4), host_threads=ctx['workers'],
                    waited_for_cpu_legs_s=round(ctx['t_wait_cpu'], 1))
    per_rank = [rank_rec]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, rank_rec)
    if rank == 0:
        rep = job['report']
        sms = dict(zip(_native.STAGE_NAMES, [float(x) for x in job['stage'][:16]]))
        grp = max(STAGE_GROUPS, key=lambda g: sum(sms.get(k, 0.0) for k in g[1]))
        dom_ms = sum(sms.get(k, 0.0) for k in grp[1])
        achieved = job['algo_bytes'] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        rates = [r['reads_per_s'] for r in per_rank if r['job_batches']]
        res = {
            'metric': 'resquiggle reads/s (10 kb DNA, bw=500)', 'value': rep['value'], 'unit': 'reads/s',
            'n_gpus': world, 'steps': int(job['n_batches']), 'warmup': 2 * a.slots + 2,
            'ms_per_step': round(job['dt'] / max(job['n_batches'], 1) * 1e3, 3), 'higher_is_better': True,
            'scaling': scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': rep['what'], 'job_reads': rep['job_reads'], 'reads_per_batch': a.reads,
                       'bases': int(a.bases), 'bandwidth': a.bandwidth, 'success_rate': rep['success_rate'],
                       'parallelism': 'batches of one job sharded over %d process(es) through a shared batch '
                                      'counter; no collective on the data path, gloo control plane' % world,
                       'devices': [r['device'] for r in per_rank], 'visible_devices': ndev,
                       'ranks_share_devices': ndev < world,
                       'setup_s': {'synthesis': 0.0, 'cpu_legs': round(t_cpu, 1),
                                   'total_wall': round(time.perf_counter() - t_start, 1)},
                       'stage_ms': {k: round(v, 3) for k, v in sms.items() if v > 0}},
            'per_rank': per_rank,
            'per_rank_reads_per_s': {'min': min(rates) if rates else None, 'max': max(rates) if rates else None},
            'distinct_read_job': rep,
            'roofline': {'bound': 'hbm', 'kernel': grp[2], 'stage_group': grp[0],
                         'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': None,
                         'traffic_scope': 'not measured in the job form (the cfg2 line carries the counter passes)',
                         'algorithmic_bytes_per_launch': job['algo_bytes'], 'kernel_ms': round(dom_ms, 3)},
        }
        if cpu_legs:
            res['cpu_baseline'] = dict(cpu_legs[0], cpu=cpu_model(),
                                       all_cores=cpu_legs[1] if len(cpu_legs) > 1 else None)
        os.write(json_fd, (json.dumps(res) + '\n').encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--reads', type=int, default=10000, help='reads per GPU per step')
    ap.add_argument('--bases', type=int, default=10000)
    ap.add_argument('--bandwidth', type=int, default=500)
    ap.add_argument('--cpu-sample', type=int, default=150, help='reads of the 1-thread CPU leg')
    ap.add_argument('--cpu-per-core', type=int, default=4, help='reads per core of the all-core CPU leg')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--preset', choices=['cfg2', 'cfg3', 'cfg1', 'cfg4', 'cfg5', 'longtail'], default=None,
                    help='BASELINE.json configs: cfg2 10kb/W=500 (default), cfg3 10kb/W=300, '
                         'cfg1 2kb/W=100, cfg4 RNA 3kb/W=500; cfg5: one job of --job-reads DISTINCT 10 kb '
                         'DNA reads (W=500) drawn batch by batch from the host work queue, every batch '
                         'synthesised on the device of the rank that drew it; longtail: log-normal '
                         '1-200 kb DNA reads (median 8 kb), W=500, batches cut by the planner')
    ap.add_argument('--job-reads', type=int, default=None,
                    help='cfg5: reads of the whole job (default 125000 per rank = a million on 8 GPUs; '
                         'given explicitly the job is fixed and the scaling strong)')
    ap.add_argument('--job-seed', type=int, default=20260927, help='cfg5: seed of the job (read r of the job is a function of (seed, r) alone)')
    ap.add_argument('--e2e', choices=['compact', 'full', 'none'], default='compact',
                    help='end-to-end (reads in -> results out) measurement: int16 in / records + '
                         'int32 boundaries out, float64 in / float64 signal + int64 boundaries '
                         'out, or skipped')
    ap.add_argument('--subsample', choices=['device', 'numpy'], default='device',
                    help='end-to-end leg: Theil-Sen subsample drawn on the device (keyed permutation) '
                         'or by np.random.choice on the feeder thread (the seeded-parity mode)')
    ap.add_argument('--stream-batch', type=int, default=None,
                    help='reads per streamed batch (default: 10000 compact, 5000 full: its float64 outputs are page-locked per slot)')
    ap.add_argument('--slots', type=int, default=2,
                    help='engine slots per GPU of the streaming pipeline (round 4: 2 slots 98.9 k reads/s end to end at '
                         'cfg2, 3 slots 92.3 k -- the event-detection kernel fills a CU alone now; RNA 125 k vs 112 k)')
    ap.add_argument('--serial-compute', action='store_true',
                    help='streaming: kernel sequences of the slots back to back (tba_batch_wait_for) instead of interleaved')
    ap.add_argument('--interleaved', action='store_true',
                    help='a second resident figure after the timed region (config.two_resident_batches_alternating): '
                         'two batches alternating on two engines; off by default -- a profile of the default '
                         'command then holds the launches of the timed passes only')
    ap.add_argument('--resident-split', type=int, default=1,
                    help='resident phase: cut the batch into this many sub-batches, each on its own '
                         'engine / stream (kernels of different sub-batches overlap)')
    ap.add_argument('--longtail-max-bases', type=int, default=200000,
                    help='longtail preset: clip of the log-normal read lengths (round 2 used 100000)')
    ap.add_argument('--tail-bases', type=int, default=30000,
                    help='longtail preset: reads longer than this form batches of their own (planner.plan_batches)')
    ap.add_argument('--api-reads', type=int, default=5000, help='reads of the resquiggle_batch API leg (0: skip)')
    ap.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 counter passes behind roofline.traffic')
    ap.add_argument('--pmc-reads', type=int, default=1 << 30,
                    help='reads of the counter passes: the whole timed batch by default (a 2 048-read sub-batch read 5 %% low in round 5: '
                         'per-workgroup tiles amortise differently); they take the dispatch forms of the timed batch whatever this is')
    ap.add_argument('--pmc-child', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-child', default=None, help=argparse.SUPPRESS)  # JSON job of the CPU legs
    ap.add_argument('--engine-stub', default=None, help=argparse.SUPPRESS)  # tests/: host logic without a GPU
    a = ap.parse_args()
    if a.pmc_child:
        return pmc_child(a.pmc_child)
    if a.cpu_child:
        return cpu_child(a.cpu_child)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(a.gpus)
    samp_name = 'DNA'
    if a.preset == 'cfg3':
        a.bandwidth = 300
    elif a.preset == 'cfg1':
        a.bases, a.bandwidth = 2000, 100
    elif a.preset == 'cfg4':
        samp_name, a.bases, a.bandwidth = 'RNA', 3000, 500
    cfg5 = a.preset == 'cfg5'
    if cfg5:
        a.bases, a.bandwidth = 10000, 500
    longtail = a.preset == 'longtail'
    if longtail and a.reads == 10000:
        a.reads = 16000   # enough work per pass to hide the serial time of a 200 kb read
    if longtail and a.slots == 2:
        a.slots = 4       # (measured in round 3: 3 -> 39.2 k, 4 -> 40.9 k, 6 -> 27 k reads/s end to end)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    t_start = time.perf_counter()
    # stdout carries exactly one line, the JSON of rank 0: library chatter written to fd 1 (gloo
    # banners, HIP warnings) is sent to stderr for the rest of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # synthetic input first: worker processes must be forked before HIP is initialised
    from tombo_amd import _native, planner, sharding, streaming, tombo_stats as ts, tombo_helper as th
    from tombo_amd._default_parameters import SIG_MATCH_THRESH, STALL_PARAMS
    stub = None
    if a.engine_stub:
        import importlib.util
        spec = importlib.util.spec_from_file_location('tba_engine_stub', a.engine_stub)
        stub = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(stub)
        stub.install(_native)
    samp = th.seqSampleType(samp_name, samp_name == 'RNA')
    rna = samp_name == 'RNA'
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=a.bandwidth)
    if a.bandwidth <= 100:
        params = params._replace(band_bound_thresh=10)  # the default 40 fails every read at W=100
    stall_params = th.stallParams(**STALL_PARAMS) if rna else None
    # host threads per rank (synthesis workers, pack threads): the ranks of a node share its cores, and
    # every rank also runs a feeder thread, a generator helper and its own main thread beside them
    ncpu = os.cpu_count() or 8
    workers = max(1, min(32, (ncpu - 3 * world) // max(world, 1) if world > 1 else ncpu))
    # ranks > 0 stay off the host's cores until rank 0's CPU baseline legs are over
    es = early_store(world) if world > 1 else None
    t_wait_cpu = time.perf_counter()
    if es is not None and rank != 0 and not a.no_cpu_baseline:
        try:
            es.wait(['tba_cpu_legs_done'])
        except Exception:
            pass
    t_wait_cpu = time.perf_counter() - t_wait_cpu
    seed0 = 1000003 * (rank + 1)    # every rank has its own, distinct reads
    bases = longtail_bases(a.reads, seed0, a.longtail_max_bases) if longtail else np.full(a.reads, a.bases, np.int64)
    # The CPU legs run on rank 0 at every N, FIRST: before any HIP state exists, before the control
    # plane is up and before any rank loads the host with read synthesis (the other ranks wait for
    # the flag below; the child process makes its own copy of the first reads from their seeds)
    cpu_legs = None
    t_cpu = time.perf_counter()
    if rank == 0 and not a.no_cpu_baseline:
        cores = os.cpu_count() or 1
        n_cpu = min(a.reads, max(a.cpu_sample, a.cpu_per_core * cores if cores > 1 else 0, 1))
        job = json.dumps(dict(samp=samp_name, bandwidth=a.bandwidth, bases=[int(b) for b in bases[:n_cpu]],
                              seed0=seed0, n_single=a.cpu_sample, n_per_core=a.cpu_per_core))
        env = dict(os.environ)
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'TBA_STORE_PORT'):
            env.pop(k, None)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-child', job], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=1200)
        lines = [x for x in out.stdout.decode().splitlines() if x.startswith('[')]
        cpu_legs = json.loads(lines[-1]) if out.returncode == 0 and lines else None
    t_cpu = time.perf_counter() - t_cpu
    if es is not None and rank == 0:
        try:
            es.set('tba_cpu_legs_done', '1')
        except Exception:
            pass
    es = None   # (the client socket goes before the synthesis workers are forked)

    want_dac = a.e2e == 'compact' or a.api_reads > 0
    t_gen = time.perf_counter()
    if cfg5:   # no rank synthesises anything on the host: every batch of the job is drawn on the device
        seqs, raws, dacs = [], [], []
    else:
        seqs, raws, dacs = make_reads(bases, seed0, workers, samp_name, want_dac)
    t_gen = time.perf_counter() - t_gen
    rng = np.random.RandomState(12345 + rank)
    si = np.zeros((a.reads, 1000), np.int64)
    for i in range(a.reads if not cfg5 else 0):
        if bases[i] > 1000:
            si[i] = rng.choice(int(bases[i]), 1000, replace=False)
    if not (bases > 1000).any():
        si = None
    n_raw = np.array([len(r) for r in raws], np.int64)
    seq_len = np.array([len(s) for s in seqs], np.int64)

    dist = init_control_plane(rank, world) if world > 1 else None
    import torch
    have_cuda = stub is None and torch.cuda.is_available()
    ndev = max(torch.cuda.device_count(), 1) if have_cuda else max(_native.lib().tba_device_count(), 1)
    dev = local_rank % ndev
    dev_name = 'stub'
    if have_cuda:
        torch.cuda.set_device(dev)
        dev_name = torch.cuda.get_device_name(dev)

    def dev_sync():
        if have_cuda:
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        dev_sync()

    def reduce(x, op):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=op)
        return float(t.item())

    def max_over_ranks(x):
        return reduce(x, dist.ReduceOp.MAX) if dist is not None else x

    def sum_over_ranks(x):
        return reduce(x, dist.ReduceOp.SUM) if dist is not None else x

    p = _native.make_params(params)
    o = _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH[samp_name],
                          stall_params=stall_params)

    if cfg5 or (world > 1 and not longtail and not rna and a.bandwidth == 500 and a.bases == 10000):
        # the job of BASELINE.json's cfg5: distinct reads, made on the device batch by batch
        job_reads = a.job_reads if (cfg5 and a.job_reads) else (125000 if cfg5 else 4 * a.reads) * world
        ctx = dict(_native=_native, streaming=streaming, sharding=sharding, model=model, params=params, samp=samp,
                   dev=dev, rank=rank, world=world, barrier=barrier, max_over_ranks=max_over_ranks,
                   sum_over_ranks=sum_over_ranks, dev_sync=dev_sync, slots=a.slots, workers=workers,
                   t_wait_cpu=t_wait_cpu)
    if cfg5:
        job = run_job(ctx, job_reads, a.reads, a.bases, a.job_seed)
        return finish_cfg5(a, job, ctx, dist, json_fd, cpu_legs, dev_name, ndev, t_start, t_cpu,
                           'strong' if a.job_reads else 'weak')

    # ---- phase 1: resident ---------------------------------------------------------------
    # one engine per planned batch (the uniform presets are a single batch); a pass = every
    # batch's kernel sequence, each on its own stream
    probe = _native.Engine(dev)
    probe.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
    free_b, total_b = probe.device_mem()
    if longtail:
        # several batches resident at once, each on its own stream: the one-wave-per-read kernels
        # of the batch holding the longest reads run for a long time, the other batches fill the machine
        tot = planner.exact_bytes(n_raw, seq_len, p, o, model.kmer_width)
        plan = planner.plan_batches(n_raw, seq_len, p, o, model.kmer_width,
                                    min(0.2 * free_b, max(tot / 6.0, 2e9)), tail_bases=a.tail_bases)
    else:
        plan = [x for x in np.array_split(np.arange(a.reads), max(1, a.resident_split)) if len(x)]
    engines = [probe] + [_native.Engine(dev) for _ in plan[1:]]
    t_up = time.perf_counter()
    up_bytes = 0
    codes = [ts.encode_seq(s) for s in seqs]
    for eng, idx in zip(engines, plan):
        eng.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
        eng.upload(p, o, [raws[i] for i in idx], [codes[i] for i in idx],
                   samp_ind=None if si is None else si[idx])
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

    def batch_digest():
        """CRC of every resident batch's boundaries + status, and the rows the traceback's verifier disagreed on
        (TBA_GET_TB_VERIFY_FAIL): taken after the warm-up and again after the last timed pass, outside the clock --
        the timed passes run ONE resident batch again and again, so the two must be equal and the count zero"""
        if stub is not None:
            return None, 0
        crc, vf = 0, 0
        for eng in engines:
            d = eng.download(want_norm=False)
            crc = zlib.crc32(d['status'].tobytes(), zlib.crc32(d['segs'].tobytes(), crc))
            vf += int(eng.get(_native.GET_TB_VERIFY_FAIL).sum())
        return crc, vf

    if a.warmup == 0:
        one_pass()
    digest_first, vf_first = batch_digest()
    queue = sharding.BatchQueue(a.steps * world, key='resident')
    barrier()
    t0 = time.perf_counter()
    stage = np.zeros(32)
    my_steps = 0
    for _ in queue:
        one_pass()
        for eng in engines:
            stage += eng.get(_native.GET_KERNEL_MS)
        my_steps += 1
    dev_sync()
    my_dt = time.perf_counter() - t0     # this rank's own busy time (imbalance shows in per_rank)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    steps_done = int(round(sum_over_ranks(float(my_steps))))
    n_ok = 0
    for eng in engines:
        n_ok += int((eng.download(want_norm=False)['status'] == 0).sum())
    digest_last, vf_last = batch_digest()
    if os.environ.get('TBA_DBG_PHASES'):
        # profiling aid: per-read debug counters of a -DTBA_PHASE_DEBUG / -DTBA_SWEEP_STATS build
        # (TBA_EXTRA_HIPCC_FLAGS, see tombo_amd/_native.py), to stderr
        d = engines[0].get(_native.GET_DEBUG_COUNTERS)
        print('dbg mean', ' '.join('%.1f' % x for x in d.mean(axis=0)), file=sys.stderr)
        print('dbg median', ' '.join(str(int(x)) for x in np.median(d, axis=0)), file=sys.stderr)
        print('dbg max', ' '.join(str(int(x)) for x in d.max(axis=0)), file=sys.stderr)
        print('dbg p99', ' '.join(str(int(x)) for x in np.percentile(d, 99, axis=0)), file=sys.stderr)
    stage /= max(my_steps, 1)
    # a second figure, not `value`: TWO resident batches (the same reads uploaded twice) whose passes
    # alternate on two engines / streams, the next pass enqueued before the previous one is waited
    # for -- the memory-bound stages of one batch run beside the tail of the other's forward pass
    two_batches = None
    if world == 1 and len(plan) == 1 and a.interleaved and stub is None:
        eng2 = _native.Engine(dev)
        eng2.set_model(model.level_means, model.level_sds, model.kmer_width, model.central_pos)
        eng2.upload(p, o, [raws[i] for i in plan[0]], [codes[i] for i in plan[0]],
                    samp_ind=None if si is None else si[plan[0]])
        pair = [engines[0], eng2]

        def alternate(k):
            pair[0].enqueue()
            for j in range(1, k):
                pair[j % 2].enqueue()
                pair[(j - 1) % 2].sync()
            pair[(k - 1) % 2].sync()

        alternate(max(2, a.warmup))
        dev_sync()
        t1 = time.perf_counter()
        alternate(a.steps)
        dev_sync()
        dt2 = time.perf_counter() - t1
        ok2 = int((eng2.download(want_norm=False)['status'] == 0).sum())
        two_batches = {'reads_per_s': round(a.reads * a.steps / dt2, 2), 'ms_per_step': round(dt2 / a.steps * 1e3, 3),
                       'steps': a.steps, 'ok_second_batch': ok2,
                       'note': 'two resident batches, passes alternating on two engines (streams), at most two in flight'}
        eng2.close()
        del eng2, pair
    for eng in engines:
        eng.close()
    del engines, probe
    rank_rec = dict(rank=rank, device=dev, device_name=dev_name, steps=my_steps,
                    resident_reads_per_s=round(a.reads * my_steps / my_dt, 2) if my_steps else 0.0,
                    resident_busy_s=round(my_dt, 4),
                    digest_stable=None if digest_first is None else bool(digest_first == digest_last),
                    tb_verify_fail_rows=int(vf_first + vf_last))

    # ---- phase 2: end to end through the streaming pipeline ---------------------------------
    e2e = None
    if a.e2e != 'none':
        compact = a.e2e == 'compact'
        if a.stream_batch is None:   # (ranks sharing one device -- a rehearsal -- share its memory too)
            a.stream_batch = 10000 if compact and ndev >= world else 5000
        # what a reader hands over: one array per read (the FAST5 `Signal` dataset for compact:
        # int16, RNA in acquisition order -> flipped on the device) and the sequence string
        src = dacs if compact else raws
        dev_flip = rna and compact
        if longtail:
            o_s = _native.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH[samp_name],
                                    skip_norm_out=compact)
            splan = planner.plan_batches(n_raw, seq_len, p, o_s, model.kmer_width,
                                         min(0.2, 0.6 / a.slots) * free_b,
                                         np.int16 if compact else np.float64, max_reads=a.stream_batch,
                                         tail_bases=a.tail_bases)
        else:
            splan = [np.arange(s, min(s + a.stream_batch, a.reads)) for s in range(0, a.reads, a.stream_batch)]
        host_draw = a.subsample == 'numpy' and si is not None
        pipe = streaming.StreamPipeline(model, params, n_slots=a.slots, device=dev, outlier_thresh=5.0,
                                        seq_samp_type=samp, want_norm=not compact,
                                        segs_dtype=np.int32 if compact else np.int64,
                                        reverse_raw=dev_flip, stall_params=stall_params,
                                        subsample_seed=None if host_draw else 20260927 + rank,
                                        in_order=not longtail,
                                        serial_compute=a.serial_compute and not longtail)
        feeder = streaming.ReadFeeder(n_slots=a.slots, n_threads=workers)
        pools = [([src[i] for i in idx], [seqs[i] for i in idx], idx) for idx in splan]
        cnt = dict(reads=0, ok=0, out=0, moved=0, submit_s=0.0, pack_wait_s=0.0, nb=0)
        est = np.zeros(32)

        def start_pack(k):
            sub_r, sub_s, idx = pools[k]
            smp = None
            if host_draw:   # the reference's own draw, per read, on the feeder thread
                smp = [np.random.choice(int(bases[i]), 1000, replace=False) if bases[i] > 1000 else None
                       for i in idx]
            feeder.prefetch(sub_r, sub_s, samp_inds=smp, tag=k)

        def consume(res):
            if os.environ.get('TBA_BENCH_VERBOSE'):
                print('batch', res.tag, 'reads', res.n, ' '.join('%s=%.1f' % (k, v) for k, v in zip(
                    _native.STAGE_NAMES, res.stage_ms[:16]) if v > 0.5), file=sys.stderr)
            cnt['reads'] += res.n
            cnt['nb'] += 1
            est[:] += res.stage_ms
            cnt['ok'] += int((res.results['status'] == 0).sum())
            cnt['out'] += res.results.nbytes + res.segs.nbytes + (0 if res.norm is None else res.norm.nbytes)

        def stream(batch_ids):
            """pack batch b+1 on the feeder thread while batch b is submitted"""
            it = iter(batch_ids)
            nxt = next(it, None)
            if nxt is not None:
                start_pack(nxt % len(pools))
            while nxt is not None:
                tw = time.perf_counter()
                batch = feeder.take()
                cnt['pack_wait_s'] += time.perf_counter() - tw
                nxt = next(it, None)
                if nxt is not None:
                    start_pack(nxt % len(pools))
                cnt['moved'] += batch.raw.nbytes + batch.seq.nbytes + (0 if batch.samp_ind is None else batch.samp_ind.nbytes)
                ts0 = time.perf_counter()
                done = pipe.submit(batch)
                cnt['submit_s'] += time.perf_counter() - ts0
                if os.environ.get('TBA_BENCH_VERBOSE'):
                    print('t=%.3f submit tag %s took %.3f (pack wait %.3f) -> done %s' % (
                        time.perf_counter() - t_start, batch.tag, time.perf_counter() - ts0, ts0 - tw,
                        None if done is None else done.tag), file=sys.stderr)
                if done is not None:
                    consume(done)
            for done in pipe.flush():
                consume(done)
        # warm-up: every slot and every staging set sees the largest batch (buffers sized, code
        # paged in); the transfer rate of one isolated upload is taken on the way
        big = max(range(len(pools)), key=lambda k: int(n_raw[pools[k][2]].sum()))
        t_pin = time.perf_counter()
        stream([big] * (2 * a.slots + 2))
        pipe.reserve(max(len(x[2]) for x in pools), max(int(bases[x[2]].sum()) + len(x[2]) for x in pools),
                     max(int(n_raw[x[2]].sum()) for x in pools))
        if len(pools) > 1:
            # ragged pools: every slot sees every batch shape once (upload only), so that no engine
            # buffer has to grow -- hipFree / hipMalloc stall the whole device -- inside the clock
            for k in range(len(pools)):
                wbk = feeder.pack(pools[k][0], pools[k][1])
                for sl_ in pipe.slots:
                    sl_.eng.upload_packed(pipe.params, pipe.opts, wbk.raw, wbk.raw_off, wbk.seq, wbk.seq_off, wait=True)
                wbk.release()
        t_pin = time.perf_counter() - t_pin
        wb = feeder.pack(pools[big][0], pools[big][1])
        dev_sync()
        th2d = time.perf_counter()
        pipe.slots[0].eng.upload_packed(pipe.params, pipe.opts, wb.raw, wb.raw_off, wb.seq, wb.seq_off, wait=True)
        th2d = time.perf_counter() - th2d
        t_pk = time.perf_counter()
        wb2 = feeder.pack(pools[big][0], pools[big][1])
        t_pk = time.perf_counter() - t_pk
        wb.release()
        wb2.release()
        big_bytes = wb.raw.nbytes + wb.seq.nbytes
        for k in cnt:
            cnt[k] = 0 if isinstance(cnt[k], int) else 0.0
        est[:] = 0
        # batches of the whole job: K passes over the pool, but never so few that filling and
        # draining the slots is most of the measurement
        # (filling and draining the slots costs about one batch time: at 8 batches that is 10 % of the
        # measurement -- 80-85 k reads/s at cfg2 against 91-93 k over 24 batches; at least 16 here)
        n_stream = max(len(pools) * a.steps, 16 if not longtail else 2 * a.slots + 2) * world
        queue = sharding.BatchQueue(n_stream, key='stream')
        barrier()
        t0 = time.perf_counter()
        stream(queue)
        dev_sync()
        my_dt_e = time.perf_counter() - t0
        barrier()
        dt_e = max_over_ranks(time.perf_counter() - t0)
        tot_reads = sum_over_ranks(float(cnt['reads']))
        rank_rec.update(stream_batches=cnt['nb'], stream_reads_per_s=round(cnt['reads'] / my_dt_e, 2),
                        stream_busy_s=round(my_dt_e, 4))
        e2e = {
            'value': round(tot_reads / dt_e, 2), 'unit': 'reads/s', 'mode': a.e2e,
            'what': ('per-read int16 DAC arrays (RNA: acquisition order) + sequence strings -> packed '
                     'into page-locked CSR staging by native threads -> upload -> %sfull pipeline '
                     '(Theil-Sen subsample: %s) -> 64-byte record + int32 base boundaries per read in '
                     'page-locked host memory' % ('flip + stall detection + ' if rna else '',
                                                  'np.random.choice on the feeder thread' if host_draw else
                                                  'drawn on the device') if compact else
                     'per-read float64 arrays + sequence strings -> packed -> upload -> full pipeline -> '
                     'record + int64 boundaries + float64 normalised signal in page-locked host memory'),
            'slots_per_gpu': a.slots, 'reads_per_batch': int(np.mean([len(x[2]) for x in pools])),
            'batches': n_stream, 'reads': int(tot_reads), 'seconds': round(dt_e, 4),
            'success_rate': round(sum_over_ranks(float(cnt['ok'])) / max(tot_reads, 1), 4),
            'in_GB_per_10k_reads': round(cnt['moved'] / max(cnt['reads'], 1) * 1e4 / 1e9, 3),
            'out_GB_per_10k_reads': round(cnt['out'] / max(cnt['reads'], 1) * 1e4 / 1e9, 3),
            'h2d_GBps': round(big_bytes / th2d / 1e9, 2),
            'pack_GBps': round(big_bytes / t_pk / 1e9, 2), 'pack_threads': workers,
            'host_s_in_submit': round(cnt['submit_s'], 4),
            'host_s_waiting_for_packer': round(cnt['pack_wait_s'], 4),
            'stage_ms_per_batch': {k: round(float(v) / max(cnt['nb'], 1), 3) for k, v in
                                   zip(_native.STAGE_NAMES, est[:16]) if v > 0},
            'warmup_incl_pinning_s': round(t_pin, 2)}
        pipe.close()
        feeder.close()

    # ---- phase 3 (N = 1): the drop-in API itself -------------------------------------------
    api = None
    if world == 1 and a.api_reads > 0 and not longtail:
        from tombo_amd import resquiggle as rq
        n_api = min(a.api_reads, a.reads)
        mk = lambda i, sig: th.resquiggleResults(
            align_info=th.alignInfo('read_%d' % i, 'BaseCalled_template', 0, 0, 0, 0, int(bases[i]), 0),
            genome_loc=th.genomeLocation(0, '+', 'synth'), genome_seq=seqs[i], mean_q_score=10.0,
            raw_signal=sig)
        # what _io_and_map_read hands the worker: the file's int16 samples (RNA: acquisition order)
        mrs = [mk(i, dacs[i]) for i in range(n_api)]
        eng = rq.get_engine(dev)
        kw = dict(outlier_thresh=5.0, seq_samp_type=samp, reverse_raw=rna, stall_params=stall_params)
        if dev != rq.default_device():
            kw['engine'] = eng   # (an explicit engine: one batch at a time, no streaming inside the call)
        api = {'reads': n_api, 'raw_dtype': 'int16', 'returns': 'list of resquiggleResults (float64 '
               'normalised signal + int64 boundaries per read), same as the reference'}
        for mode, extra in (('numpy_subsample', {}), ('device_subsample', dict(subsample_seed=1)),
                            ('device_subsample_no_signal', dict(subsample_seed=1, return_signal=False))):
            rq.resquiggle_batch(mrs[:64], model, params, **kw, **extra)    # staging sized, code paged in
            rq.resquiggle_batch(mrs, model, params, **kw, **extra)
            res = None   # (the previous results -- gigabytes of per-read arrays -- are freed outside the clock)
            gc.collect()
            prof = None
            if os.environ.get('TBA_BENCH_VERBOSE'):
                import cProfile
                prof = cProfile.Profile()
                prof.enable()
            t0 = time.perf_counter()
            res = rq.resquiggle_batch(mrs, model, params, **kw, **extra)
            dta = time.perf_counter() - t0
            if prof is not None:
                import pstats
                prof.disable()
                print('api leg', mode, '%.1f ms' % (dta * 1e3), file=sys.stderr)
                pstats.Stats(prof, stream=sys.stderr).sort_stats('cumulative').print_stats(8)
            api['resquiggle_batch_' + mode] = {
                'reads_per_s': round(n_api / dta, 1), 'seconds': round(dta, 4),
                'ok': sum(not isinstance(r, Exception) for r in res)}
        # the floor of the legs that hand back the float64 signal: the results cross PCIe (page-locked pool blocks,
        # one D2H copy per sub-batch) -- measured here with a copy of the same size out of the engine's device memory
        sig_bytes = float(sum(int(len(dacs[i])) for i in range(n_api))) * 8.0
        try:
            src = torch.empty(1 << 28, dtype=torch.uint8, device='cuda:%d' % dev)      # 256 MiB probe
            dst = torch.empty(1 << 28, dtype=torch.uint8).pin_memory()
            dst.copy_(src)
            torch.cuda.synchronize(dev)
            d2h = 0.0
            for _ in range(3):                     # (best of three: the first timed copies sometimes run behind the legs' tails)
                t0 = time.perf_counter()
                for _ in range(4):
                    dst.copy_(src)
                torch.cuda.synchronize(dev)
                d2h = max(d2h, 4 * float(1 << 28) / (time.perf_counter() - t0))
            del src, dst
            api['d2h_GBps_page_locked'] = round(d2h / 1e9, 1)
            api['d2h_floor_reads_per_s'] = round(n_api / (sig_bytes / d2h), 1)
            api['d2h_floor_note'] = ('%.2f GB of float64 signal per %d reads at the measured page-locked D2H rate: what the two '
                                     'legs that return the signal cannot exceed' % (sig_bytes / 1e9, n_api))
        except Exception as e:   # (evidence, not the metric)
            api['d2h_floor_note'] = 'not measured: %s' % (str(e)[:120],)
        # where the time of one call goes (second mode: no host RNG)
        t0 = time.perf_counter()
        for i in range(n_api):
            np.random.choice(int(bases[i]), 1000, replace=False) if bases[i] > 1000 else None
        api['host_s_numpy_subsample_draws'] = round(time.perf_counter() - t0, 4)
        lat = []
        for i in range(24):   # the worker's per-read sequence: adjust_map_res, then resquiggle_read
            t0 = time.perf_counter()
            rq.resquiggle_read(rq.adjust_map_res(mrs[i % n_api], samp), model, params, 5.0, seq_samp_type=samp)
            lat.append(time.perf_counter() - t0)
        api['resquiggle_read_latency_ms'] = {'median': round(float(np.median(lat[4:])) * 1e3, 3),
                                             'p90': round(float(np.percentile(lat[4:], 90)) * 1e3, 3),
                                             'calls': len(lat) - 4}

    # ---- phase 4 (N > 1, the default configuration): a short job of distinct reads ---------------
    job = None
    if world > 1 and not longtail and not rna and a.bandwidth == 500 and a.bases == 10000:
        job = run_job(ctx, job_reads, a.reads, a.bases, a.job_seed)
        rank_rec.update(job_batches=job['my_batches'], job_reads=job['my_reads'])
    rank_rec.update(host_threads=workers, waited_for_cpu_legs_s=round(t_wait_cpu, 1))

    per_rank = [rank_rec]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, rank_rec)

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = a.reads * steps_done / dt
        tot_bases = float(bases.sum())
        names = _native.STAGE_NAMES
        sms = dict(zip(names, [float(x) for x in stage[:16]]))
        # dominant stage group of THIS configuration (HIP events on the engine's own stream)
        grp = max(STAGE_GROUPS, key=lambda g: sum(sms.get(k, 0.0) for k in g[1]))
        dom_ms = sum(sms.get(k, 0.0) for k in grp[1])
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        dp_ms = sms['main_dp']
        traffic = traffic_up = None
        traffic_note = 'not measured (--no-pmc, or already under a profiler)'
        traffic_kernels = traffic_forms = None
        t_pmc = time.perf_counter()
        if not a.no_pmc and not _under_profiler() and stub is None:
            try:   # rank 0's child passes, after the timed regions, on rank 0's device
                sub = np.arange(min(a.pmc_reads, a.reads))
                packed = pack_lists([raws[i] for i in sub], [seqs[i] for i in sub],
                                    None if si is None else si[sub])
                per_raw, per_up, traffic_kernels, traffic_forms = measure_pmc_traffic(
                    packed, dict(samp=samp_name, bandwidth=a.bandwidth, timed_reads=int(a.reads)), device=dev)
                # the counter passes run a sub-batch of the same reads; traffic scales with the
                # samples / bases processed
                scale = float(n_raw.sum()) / float(n_raw[sub].sum())
                traffic = per_raw * len(sub) * scale
                traffic_up = per_up * len(sub) * scale
                traffic_note = ('whole pipeline, FETCH_SIZE + WRITE_SIZE (KiB) from two rocprofv3 --pmc '
                                'passes of a %d-read sub-batch of this run forced through the dispatch '
                                'forms of the timed %d-read batch, scaled by samples to the '
                                'launch; traffic_fetch_doubled applies the gfx950 wide-read correction '
                                'to every read (upper bound)' % (len(sub), a.reads))
            except Exception as e:  # counters are evidence, not the metric: never fail the bench
                traffic_note = 'counter passes failed: %s' % (str(e)[:200],)
        t_pmc = time.perf_counter() - t_pmc
        rates = [r['resident_reads_per_s'] for r in per_rank if r['steps']]
        res = {
            'metric': 'resquiggle reads/s (%s, bw=%d)' % (
                '10 kb DNA' if (samp_name, a.bases, longtail) == ('DNA', 10000, False) else
                'long-tailed 1-%d kb DNA' % (a.longtail_max_bases // 1000) if longtail else
                '%g kb %s' % (a.bases / 1000.0, samp_name), a.bandwidth), 'value': round(value, 2),
            'unit': 'reads/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%d synthetic %s %s reads per GPU per step, bandwidth=%d, full '
                                   'resquiggle_read path%s, float64 inputs resident in HBM; %d passes '
                                   'drawn from the host work queue by %d rank(s)' % (
                                       a.reads, 'long-tailed (1-%d kb)' % (a.longtail_max_bases // 1000) if longtail else '%d-base' % a.bases,
                                       samp_name, a.bandwidth,
                                       ' incl. the worker\'s stall detection (ts.identify_stalls) on the device' if rna else '',
                                       a.steps * world, world),
                       'reads_per_gpu': a.reads, 'bases': int(a.bases) if not longtail else None,
                       'mean_bases': round(tot_bases / a.reads, 1), 'max_bases': int(bases.max()),
                       'bandwidth': a.bandwidth,
                       'resident_batches_per_gpu': len(plan),
                       'two_resident_batches_alternating': two_batches,
                       'bases_per_s': round(tot_bases * steps_done / dt, 1),
                       'success_rate': round(n_ok / float(a.reads), 4),
                       # the resident batch gave the same boundaries + status after the warm-up and after the last
                       # timed pass (every rank), and the traceback's verifier found nothing (rows, all ranks)
                       'digest_stable': None if any(r['digest_stable'] is None for r in per_rank) else all(r['digest_stable'] for r in per_rank),
                       'tb_verify_fail_rows': sum(r['tb_verify_fail_rows'] for r in per_rank),
                       'parallelism': 'reads sharded over %d process(es) through a shared batch '
                                      'counter; no collective on the data path, gloo control plane' % world,
                       'devices': [r['device'] for r in per_rank], 'visible_devices': ndev,
                       'ranks_share_devices': ndev < world,
                       'h2d_upload_s': round(t_up, 3),
                       'h2d_upload_GBps_pageable': round(up_bytes / t_up / 1e9, 2),
                       'setup_s': {'synthesis': round(t_gen, 1), 'cpu_legs': round(t_cpu, 1),
                                   'pmc_child_passes': round(t_pmc, 1),
                                   'total_wall': round(time.perf_counter() - t_start, 1)},
                       'stage_ms': {k: round(v, 3) for k, v in sms.items() if v > 0}},
            'per_rank': per_rank,
            'per_rank_reads_per_s': {'min': min(rates) if rates else None, 'max': max(rates) if rates else None},
            'end_to_end': e2e,
            'api': api,
            'distinct_read_job': None if job is None else job['report'],
            'roofline': {'bound': 'hbm', 'kernel': grp[2], 'stage_group': grp[0],
                         'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 5),
                         'traffic': traffic, 'traffic_fetch_doubled': traffic_up,
                         'traffic_scope': traffic_note,
                         'traffic_over_algorithmic': None if traffic is None else round(traffic / algo_bytes, 3),
                         'traffic_bytes_per_read_by_kernel': traffic_kernels,
                         # TBA_ED_FORM_* / TBA_TB_FORM_* -> reads of the counted sub-batch (2: k_detect + k_pick,
                         # 1: workgroup scan + k_peaks; 16 / 64: lanes per read of k_main_tb_par)
                         'traffic_dispatch_forms': traffic_forms,
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
                    'bound': 'valu_f64_issue', 'kernel': vc[key].get('kernel', 'k_dp') + ' (main adaptive banded forward pass)',
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
