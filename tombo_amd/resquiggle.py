"""Drop-in for the per-read API of tombo.resquiggle (reference: tombo/resquiggle.py:1122-1214).

`resquiggle_read` keeps the reference signature, return type (`resquiggleResults`) and
`TomboError` messages; the work is done by the HIP batch engine (tombo_amd/csrc) through the C
ABI in include/tombo_amd.h.  `resquiggle_batch` is the throughput entry point: many reads per
kernel sequence.  Host code here only marshals buffers.
"""
import os

import numpy as np

from . import tombo_helper as th
from . import tombo_stats as ts
from . import errors
from . import _native
from ._default_parameters import (
    DEL_FIX_WINDOW, MAX_DEL_FIX_WINDOW, EXTRA_SIG_FACTOR,
    MAX_RAW_CPTS, MIN_EVENT_TO_SEQ_RATIO, SIG_MATCH_THRESH, DNA_SAMP_TYPE, RNA_SAMP_TYPE,
    MAX_POINTS_FOR_THEIL_SEN)
from .mapping import get_read_seq, map_read   # (tombo.resquiggle's names; the glue lives there)

__all__ = ['resquiggle_read', 'resquiggle_batch', 'resquiggle_batch_iters', 'adjust_map_res',
           'get_read_seq', 'map_read',
           'resquiggle_batch_events', 'batch_de_novo_stats', 'get_engine', 'default_device', 'segment_signal',
           'find_adaptive_base_assignment', 'find_seq_start_in_events',
           'find_static_base_assignment', 'resolve_skipped_bases_with_raw']

_ENGINES = {}


def default_device():
    """$LOCAL_RANK (one process per GPU) modulo the visible devices, else 0"""
    device = int(os.environ.get('LOCAL_RANK', '0'))
    n = _native.lib().tba_device_count()
    return device % n if n > 0 else device


def get_engine(device=None):
    """Process-wide engine for `device` (default: $LOCAL_RANK or 0): one process per GPU."""
    if device is None:
        device = default_device()
    if device not in _ENGINES:
        _ENGINES[device] = _native.Engine(device)
    return _ENGINES[device]


def _draw_samp_ind(n_bases):
    """The subsample of calc_kmer_fitted_shift_scale (tombo_stats.py:411-416): same call on
    numpy's global legacy RNG, so a seeded caller gets the reference's indices."""
    return np.random.choice(n_bases, MAX_POINTS_FOR_THEIL_SEN, replace=False)


def resquiggle_batch(map_results, std_ref, rsqgl_params, outlier_thresh=None,
                     all_raw_signals=None, max_raw_cpts=MAX_RAW_CPTS,
                     min_event_to_seq_ratio=MIN_EVENT_TO_SEQ_RATIO, const_scale=None,
                     skip_seq_scaling=False,
                     seq_samp_type=th.seqSampleType(DNA_SAMP_TYPE, False),
                     samp_inds=None, engine=None, return_debug=False, mem_budget=None,
                     reverse_raw=False, stall_params=None, subsample_seed=None, return_signal=True):
    """resquiggle_read over a list of `resquiggleResults` (mapping results).

    Returns a list with, per read, either a `resquiggleResults` or a `TomboError` instance
    (same message the reference raises).  `samp_inds[i]`: optional precomputed Theil-Sen
    subsample for read i (1000 indices); when omitted it is drawn from numpy's global RNG in
    read order for every read longer than 1000 bases -- or, with `subsample_seed` (an int), on
    the device by a keyed permutation (the reference's production RNG is unseeded; the numpy
    draw costs 0.1 ms per 10 kb read on the host and is the default for seeded callers).

    `reverse_raw` / `stall_params`: the worker's RNA preparation (`adjust_map_res`,
    resquiggle.py:1506-1530) as part of the batch: the signal is flipped and
    `ts.identify_stalls` runs on the device; the returned results then carry the flipped
    signal's `stall_ints` like the worker's `map_res` does.

    A list of any size is accepted: when its device footprint (`tba_batch_footprint`) exceeds
    `mem_budget` bytes (default: 60 % of the device memory that is free or already held by this
    engine) the list is cut into consecutive sub-batches that fit and the results are returned
    in input order.

    `return_signal=False`: the results carry `raw_signal=None` -- the float64 normalised signal
    (0.74 MB per 10 kb read, most of what crosses PCIe on the way back) is neither materialised nor
    downloaded.  Tombo itself stores only boundaries and scale values and re-normalises the raw
    signal when a read is loaded (tombo_helper.py:2341-2460); `resquiggle_batch_events` /
    `batch_de_novo_stats` compute what needs the signal on the device.

    Marshalling is native: the per-read arrays are packed into page-locked CSR staging by
    threads (`tba_pack_reads`), transfers are DMA, and the per-read result arrays are cut out of
    the flat downloads by threads (`tba_unpack_reads`).
    """
    eng = get_engine() if engine is None else engine
    n = len(map_results)
    if n == 0:
        return []
    eng.ensure_model(std_ref)
    K = std_ref.kmer_width
    raws = []
    pre_err = [None] * n
    for i, mr in enumerate(map_results):
        raw = mr.raw_signal if all_raw_signals is None or all_raw_signals[i] is None \
            else all_raw_signals[i]
        if raw is None:
            pre_err[i] = th.TomboError(errors.MESSAGES[21])
            raw = np.zeros(1)
        raws.append(raw if isinstance(raw, np.ndarray) else np.asarray(raw))
    args = dict(std_ref=std_ref, rsqgl_params=rsqgl_params, outlier_thresh=outlier_thresh,
                max_raw_cpts=max_raw_cpts, min_event_to_seq_ratio=min_event_to_seq_ratio,
                const_scale=const_scale, skip_seq_scaling=skip_seq_scaling,
                seq_samp_type=seq_samp_type, reverse_raw=reverse_raw, stall_params=stall_params,
                return_signal=return_signal, return_debug=return_debug)
    cuts = [(0, n)]
    if n > 1 and not return_debug:
        from . import planner
        if mem_budget is None:
            # (what is free now plus what this engine already holds; no floor: on a device that
            # other engines share a floor would stop the planner from cutting, and the upload
            # would fail with TBA_E_NOMEM instead)
            held = eng.held_bytes() + sum(e.held_bytes() for e in _STREAM_ENGINES.get(eng.device, []))
            mem_budget = 0.6 * (eng.device_mem()[0] + held)
        p_ = _native.make_params(rsqgl_params)
        o_ = _native.make_opts(min_event_to_seq_ratio=min_event_to_seq_ratio, reverse_raw=reverse_raw,
                               stall_params=stall_params, subsample_seed=subsample_seed,
                               skip_norm_out=not return_signal)
        n_raw = [r.shape[0] for r in raws]
        seq_len = [len(mr.genome_seq) for mr in map_results]
        # Sub-batches.  By memory (the engine's own footprint function) and by host staging, and --
        # for a list worth it -- by the stream: with several sub-batches in flight on as many
        # engines the packing of one, the kernels of another and the download + unpacking of a
        # third overlap (_stream_batches); one sub-batch should still fill the machine's DP
        # wavefront slots together with its neighbours (~1 600 reads each: 6 cuts of a 5 000-read list
        # took 110 ms of GPU time for what one batch does in 50).
        stream_cuts = 1
        stream_min = int(os.environ.get('TBA_API_STREAM_MIN', '2400'))   # (tests lower it)
        if engine is None and n >= stream_min and os.environ.get('TBA_API_STREAM', '1') != '0':
            stream_cuts = max(3, min(12, n // max(2 * stream_min // 3, 1)))
            if os.environ.get('TBA_API_CUTS'):     # (measurement aid)
                stream_cuts = max(2, int(os.environ['TBA_API_CUTS']))
        slot_budget = mem_budget if stream_cuts == 1 else mem_budget / _STREAM_SLOTS
        host_cap = 1 << 28   # samples per sub-batch: 2 GiB of float64 each way in page-locked staging
        target = -(-n // stream_cuts)
        if n > planner.MAX_READS or sum(n_raw) > host_cap or stream_cuts > 1 or \
                planner.exact_bytes(n_raw, seq_len, p_, o_, K) > slot_budget:
            # consecutive cuts (sort=False): the Theil-Sen subsamples are drawn from the global
            # RNG in read order, exactly as for one big batch
            parts = planner.plan_batches(n_raw, seq_len, p_, o_, K, slot_budget, sort=False,
                                         max_reads=min(planner.MAX_READS, max(target, 1)))
            cuts = []
            for idx in parts:   # ... and by host staging
                a0, acc = int(idx[0]), 0
                for i in idx:
                    if acc + n_raw[i] > host_cap and i > a0:
                        cuts.append((a0, int(i)))
                        a0, acc = int(i), 0
                    acc += n_raw[i]
                cuts.append((a0, int(idx[-1]) + 1))
    if len(cuts) > 1:
        engines = _stream_engines(eng, std_ref) if engine is None and stream_cuts > 1 else [eng]
        return _stream_batches(engines, cuts, map_results, raws, pre_err, samp_inds, subsample_seed, args)
    ctx = _submit_batch(eng, 0, n, map_results, raws, pre_err, samp_inds, subsample_seed, args)
    _sync_batch(ctx)
    _unpack_batch(ctx)
    results = _build_results(ctx)
    if return_debug:
        return results, ctx['out']
    return results


_STREAM_SLOTS = 3
_STREAM_ENGINES = {}


def release_stream_engines(device=None):
    """Destroy the extra engines a streamed `resquiggle_batch` keeps per device (their grow-only device
    buffers and page-locked staging) and hand the idle result blocks back to the system.  They are
    created again on the next call that streams."""
    for dev in ([device] if device is not None else list(_STREAM_ENGINES)):
        for e in _STREAM_ENGINES.pop(dev, []):
            e.close()
    _native.result_pool().trim()


def _stream_engines(eng, std_ref):
    """the engines a streamed `resquiggle_batch` rotates over: the process-wide one of the device
    and two more, created on first use (each has its own stream, device buffers and staging)"""
    extra = _STREAM_ENGINES.get(eng.device)
    if extra is None:
        extra = _STREAM_ENGINES[eng.device] = [_native.Engine(eng.device) for _ in range(_STREAM_SLOTS - 1)]
    engines = [eng] + extra
    for e in engines:
        e.ensure_model(std_ref)
        e.set_sharing(len(engines))
    return engines


def _stream_batches(engines, cuts, map_results, raws, pre_err, samp_inds, subsample_seed, args):
    """Sub-batches through `engines` in rotation: while one computes, the next is packed and
    uploaded and the one before is downloaded, cut into per-read arrays (a helper thread: native
    copies, GIL released) and turned into results (this thread).  Results in input order."""
    from concurrent.futures import ThreadPoolExecutor
    import time
    trace = [] if os.environ.get('TBA_API_TRACE') else None   # (what, sub-batch, seconds since the call)
    t_call = time.perf_counter()

    def mark(what, k):
        if trace is not None:
            trace.append((what, k, round(time.perf_counter() - t_call, 4)))
    out, pending = [], []          # pending: contexts in submission order
    S = len(engines)
    last_ctx = [None] * S          # the context that used engine k last (its staging is reused)
    # With the signal coming back the call is bound by the download (0.74 MB per 10 kb read): the
    # sub-batches then compute one after the other (tba_batch_wait_for), so that the download of one
    # runs under the kernels of the next; without it they share the device (small batches do not
    # fill it alone) -- measured on 5 000 reads of 10 kb, three cuts: all three finished computing
    # together after 65 ms and only then 74 ms of downloads began.
    chain = bool(args['return_signal'])
    if os.environ.get('TBA_API_CHAIN'):            # (measurement aid)
        chain = os.environ['TBA_API_CHAIN'] == '1'
    ex = ThreadPoolExecutor(1)
    try:
        def unpack(ctx):
            _unpack_batch(ctx)
            mark('unpacked', ctx['k'])

        def pump():
            """retire what the device has finished (sync is immediate then; the unpacking goes to
            the helper), build the results of what the helper is done with, in order"""
            moved = False
            for c in pending:
                if 'unpack' not in c and not c['eng'].query():
                    _sync_batch(c)
                    mark('synced', c['k'])
                    c['unpack'] = ex.submit(unpack, c)
                    moved = True
            while pending and 'unpack' in pending[0] and pending[0]['unpack'].done():
                c = pending.pop(0)
                c['unpack'].result()
                out.extend(_build_results(c))
                mark('built', c['k'])
                moved = True
            return moved

        for k, (a, b) in enumerate(cuts):
            e = k % S
            old = last_ctx[e]
            while old is not None and not ('unpack' in old and old['unpack'].done()):
                if not pump():           # its page-locked outputs must be free again
                    time.sleep(0.0002)
            mark('submit', k)
            # (the device-side draw of a read is keyed by its index in the whole list -- _submit_batch hands
            # `a` to the engine -- so the result does not depend on how the list was cut)
            ctx = _submit_batch(engines[e], a, b, map_results, raws, pre_err, samp_inds, subsample_seed, args,
                                after=engines[(k - 1) % S] if chain and k > 0 else None)
            ctx['k'] = k
            mark('submitted', k)
            last_ctx[e] = ctx
            pending.append(ctx)
            pump()
        while pending:
            if not pump():
                time.sleep(0.0002)
    finally:
        # whatever happened (TBA_E_NOMEM on one slot, an interrupt): nothing of this call may still be in
        # flight on staging the next call reuses, and the process-wide engine goes back to its defaults
        for e in engines:
            try:
                e.sync()
            except Exception:
                pass
            try:
                e.set_sharing(1)
            except Exception:
                pass
        ex.shutdown(wait=True)
    if trace is not None:
        import sys
        print('resquiggle_batch stream trace:', trace, file=sys.stderr)
    return out


def _submit_batch(eng, a, b, map_results, raws, pre_err, samp_inds, subsample_seed, args, after=None):
    """pack reads [a, b), upload, enqueue the kernel sequence and the downloads; returns the
    context `_sync_batch` / `_unpack_batch` / `_build_results` finish"""
    std_ref, rsqgl_params = args['std_ref'], args['rsqgl_params']
    return_signal, return_debug = args['return_signal'], args['return_debug']
    mrs, rws = map_results[a:b], raws[a:b]
    n = b - a
    K = std_ref.kmer_width
    stage = eng.host_stage()
    raw, raw_off, seq, seq_off, _ = _native.pack_reads(
        rws, [mr.genome_seq for mr in mrs], stage=stage)
    sv_in = sv_flags = None
    if any(mr.scale_values is not None for mr in mrs):
        sv_in = np.zeros((n, 4))
        sv_flags = np.zeros(n, np.int32)
        for i, mr in enumerate(mrs):
            sv = mr.scale_values
            if sv is None:
                continue
            sv_in[i, 0], sv_in[i, 1] = sv.shift, sv.scale
            sv_flags[i] = 1
            if sv.lower_lim is not None and sv.upper_lim is not None:
                sv_in[i, 2], sv_in[i, 3] = sv.lower_lim, sv.upper_lim
                sv_flags[i] |= 2
    st = sto = None
    if args['stall_params'] is None:
        stalls = [mr.stall_ints for mr in mrs]
        if any(s is not None and len(s) for s in stalls):
            st, sto = _native.pack_stalls(stalls)
    nb = np.diff(seq_off) - K + 1
    si = None
    rng_state = np.random.get_state() if len(map_results) == 1 else None
    if not args['skip_seq_scaling'] and subsample_seed is None and (nb > MAX_POINTS_FOR_THEIL_SEN).any():
        si = stage.get('si', n * MAX_POINTS_FOR_THEIL_SEN, np.int64).reshape(n, MAX_POINTS_FOR_THEIL_SEN)
        si[:] = -1   # (rows of reads that need none; the kernel rejects a negative index)
        for i in np.flatnonzero(nb > MAX_POINTS_FOR_THEIL_SEN):
            si[i] = _draw_samp_ind(int(nb[i])) if samp_inds is None or samp_inds[a + i] is None \
                else samp_inds[a + i]
    p = _native.make_params(rsqgl_params)
    seq_samp_type = args['seq_samp_type']
    o = _native.make_opts(
        outlier_thresh=args['outlier_thresh'], const_scale=args['const_scale'],
        skip_seq_scaling=args['skip_seq_scaling'],
        sig_match_thresh=None if seq_samp_type is None else SIG_MATCH_THRESH[seq_samp_type.name],
        max_raw_cpts=args['max_raw_cpts'], min_event_to_seq_ratio=args['min_event_to_seq_ratio'],
        reverse_raw=args['reverse_raw'], stall_params=args['stall_params'], subsample_seed=subsample_seed,
        subsample_first_read=a, skip_norm_out=not return_signal and not return_debug)
    eng.upload_packed(p, o, raw, raw_off, seq, seq_off, sv_in=sv_in, sv_flags=sv_flags,
                      samp_ind=si, stall_ints=st, stall_off=sto)
    if after is not None:
        eng.wait_for(after)   # this batch's kernels start when that engine's sequence has finished
    eng.enqueue()
    ctx = dict(eng=eng, a=a, b=b, n=n, nb=nb, raw_off=raw_off, seg_off=eng.seg_off.copy(),
               map_results=mrs, pre_err=pre_err[a:b], rng_state=rng_state, args=args)
    if not return_debug:
        ctx['o_res'] = stage.get('res', n, _native.RESULT_DTYPE)
        # The boundaries and the normalised signal are downloaded into page-locked blocks leased from the
        # process-wide result pool, and the per-read arrays of the results are VIEWS of those blocks (no
        # second copy of 0.8 MB per 10 kb read; a block returns to the pool when the last result that
        # looks into it is gone).  Without a lease (pool budget spent, TBA_API_ZERO_COPY=0): the engine's
        # reusable staging + a native copy into pageable memory, as before.
        # (small batches copy: a page-locked block per handful of reads is not worth holding)
        pool = _native.result_pool() if n >= int(os.environ.get('TBA_API_ZERO_COPY_MIN', '32')) and \
            os.environ.get('TBA_API_ZERO_COPY', '1') != '0' else None
        n_segs, n_norm = int(eng.seg_off[-1]), int(eng.n_raw_total)
        l_segs = pool.lease(n_segs, np.int64) if pool is not None else None
        l_norm = pool.lease(n_norm, np.float64) if pool is not None and return_signal and l_segs is not None else None
        ctx['views'] = l_segs is not None and (l_norm is not None or not return_signal)
        if not ctx['views']:
            l_segs = l_norm = None
        ctx['o_segs'] = l_segs if ctx['views'] else stage.get('segs', n_segs, np.int64)
        ctx['o_norm'] = (l_norm if ctx['views'] else stage.get('norm', n_norm, np.float64)) if return_signal else None
        eng.download_async(results=ctx['o_res'], segs64=ctx['o_segs'], norm=ctx['o_norm'])
    return ctx


def _sync_batch(ctx):
    """wait for the batch; the small per-read outputs (copies: the staging is reused)"""
    eng, args = ctx['eng'], ctx['args']
    eng.sync()
    if args['return_debug']:
        out = eng.download()
    else:
        o_res = ctx['o_res']
        out = dict(status=o_res['status'].copy(), read_start=o_res['read_start_rel_to_raw'].copy(),
                   norm_len=o_res['norm_len'].copy(), score=o_res['sig_match_score'].copy(),
                   changed=o_res['norm_params_changed'].copy(),
                   sv=np.stack([o_res['shift'], o_res['scale'], o_res['lower_lim'],
                                o_res['upper_lim']], axis=1))
    ctx['out'] = out
    status = np.asarray(out['status'])
    if ctx['rng_state'] is not None and int(status[0]) not in (0, 19, 20):
        # a batch of one is the reference's call: it only touches the RNG once the read reaches
        # sequence rescaling (calc_kmer_fitted_shift_scale), so a read that failed earlier leaves
        # the seeded stream where it was
        np.random.set_state(ctx['rng_state'])
    ctx['ok'] = np.flatnonzero((status == 0) & np.array([e is None for e in ctx['pre_err']]))
    ctx['dev_stalls'] = eng.stall_ints() if args['stall_params'] is not None else None


def _unpack_batch(ctx):
    """the per-read boundary / signal arrays out of the flat downloads (native threads)"""
    out, ok, nb, args = ctx['out'], ctx['ok'], ctx['nb'], ctx['args']
    if not args['return_debug'] and ctx.get('views'):
        so, ro = ctx['seg_off'].tolist(), ctx['raw_off'].tolist()
        o_segs, o_norm = ctx['o_segs'], ctx['o_norm']
        okl = ok.tolist()
        ctx['segs_l'] = [o_segs[so[i]:so[i + 1]] for i in okl]
        if args['return_signal']:
            nl = out['norm_len'].tolist()
            ctx['norm_l'] = [o_norm[ro[i]:ro[i] + nl[i]] for i in okl]
        else:
            ctx['norm_l'] = [None] * len(okl)
        ctx['o_segs'] = ctx['o_norm'] = None     # (only the results hold the blocks now)
    elif not args['return_debug']:
        ctx['segs_l'] = _native.unpack_reads(ctx['o_segs'], ctx['seg_off'][:-1][ok], nb[ok] + 1)
        ctx['norm_l'] = _native.unpack_reads(ctx['o_norm'], ctx['raw_off'][:-1][ok], out['norm_len'][ok]) \
            if args['return_signal'] else [None] * len(ok)
    else:
        so, ro = ctx['seg_off'], ctx['raw_off']
        ctx['segs_l'] = [out['segs'][so[i]:so[i + 1]].copy() for i in ok]
        ctx['norm_l'] = [out['norm'][ro[i]:ro[i] + int(out['norm_len'][i])].copy() for i in ok]


def _build_results(ctx):
    """resquiggleResults / TomboError per read of the batch"""
    out, ok, args, n = ctx['out'], ctx['ok'], ctx['args'], ctx['n']
    std_ref = args['std_ref']
    K, cp = std_ref.kmer_width, std_ref.central_pos
    dn = K - cp - 1
    results = [None] * n
    status = np.asarray(out['status'])
    # (thousands of small tuples are born here and none of them is garbage: with the cyclic
    # collector running, its generation passes over the caller's live objects were a quarter of the
    # host time of a 5 000-read call)
    import gc
    gc_was = gc.isenabled()
    gc.disable()
    try:
        _fill_results(results, ok, ctx['map_results'], out['sv'], out['read_start'], out['score'],
                      out['changed'], ctx['segs_l'], ctx['norm_l'], ctx['dev_stalls'],
                      args['skip_seq_scaling'], args['outlier_thresh'], args['rsqgl_params'],
                      args['const_scale'], cp, dn)
    finally:
        if gc_was:
            gc.enable()
    pre_err = ctx['pre_err']
    for i in range(n):
        if results[i] is not None:
            continue
        if pre_err[i] is not None:
            results[i] = pre_err[i]
            continue
        st_i = int(status[i])
        if st_i in errors.MESSAGES:
            results[i] = th.TomboError(errors.MESSAGES[st_i])
        else:
            results[i] = RuntimeError('Unexpected error in resquiggle engine (status %d)' % st_i)
    return results


def _fill_results(results, ok, map_results, svs, rstart, score, changed, segs_l, norm_l, dev_stalls,
                  skip_seq_scaling, outlier_thresh, rsqgl_params, const_scale, cp, dn):
    """the resquiggleResults of the successful reads of one batch (resquiggle.py:1210-1214)"""
    # Built field by field through tuple.__new__ (namedtuple._replace walks a keyword dict per call: 3 us
    # per read, the largest item of a 5 000-read call's host time), scalars converted once per batch.
    RR, SV, new = th.resquiggleResults, th.scaleValues, tuple.__new__
    I_SEQ, I_RAW, I_START, I_SEGS, I_SV, I_SCORE, I_CH, I_STALL = (
        RR._fields.index(f) for f in ('genome_seq', 'raw_signal', 'read_start_rel_to_raw', 'segs',
                                      'scale_values', 'sig_match_score', 'norm_params_changed', 'stall_ints'))
    okl = ok.tolist() if hasattr(ok, 'tolist') else list(ok)
    sv_rows = np.asarray(svs)[okl][:, :4].tolist() if len(okl) else []
    rs_l, sc_l, ch_l = (np.asarray(x)[okl].tolist() if len(okl) else [] for x in (rstart, score, changed))
    t_test = bool(rsqgl_params.use_t_test_seg)
    for k, i in enumerate(okl):
        mr = map_results[i]
        sh, scl, lo, hi = sv_rows[k]
        lo = None if lo != lo else lo          # NaN: no limit
        hi = None if hi != hi else hi
        # scaleValues.outlier_thresh: the reference stores the argument after sequence
        # rescaling (resquiggle.py:1187-1188) and normalize_raw_signal's value otherwise
        if skip_seq_scaling:
            ot = None if mr.scale_values is not None else outlier_thresh
            if t_test and mr.scale_values is None and const_scale is None:
                ot = None
        else:
            ot = outlier_thresh
        sv_new = new(SV, (sh, scl, lo, hi, ot))
        if type(mr) is not RR:
            # (a subclass, or another tuple with these field names: the field-by-field build below assumes RR's own
            # layout -- such a result goes through _replace, as every result did before round 5)
            gs = mr.genome_seq
            kw = dict(genome_seq=gs[cp:len(gs) - dn], raw_signal=norm_l[k], read_start_rel_to_raw=int(rs_l[k]),
                      segs=segs_l[k], scale_values=sv_new, sig_match_score=sc_l[k], norm_params_changed=bool(ch_l[k]))
            if dev_stalls is not None:
                kw['stall_ints'] = [list(map(int, x)) for x in dev_stalls[i]]
            results[i] = mr._replace(**kw)
            continue
        f = list(mr)
        gs = f[I_SEQ]
        f[I_SEQ] = gs[cp:len(gs) - dn]
        f[I_RAW] = norm_l[k]
        f[I_START] = int(rs_l[k])
        f[I_SEGS] = segs_l[k]
        f[I_SV] = sv_new
        f[I_SCORE] = sc_l[k]
        f[I_CH] = bool(ch_l[k])
        if dev_stalls is not None:
            f[I_STALL] = [list(map(int, x)) for x in dev_stalls[i]]
        results[i] = new(RR, f)


def resquiggle_read(map_res, std_ref, rsqgl_params, outlier_thresh=None, all_raw_signal=None,
                    max_raw_cpts=MAX_RAW_CPTS, min_event_to_seq_ratio=MIN_EVENT_TO_SEQ_RATIO,
                    const_scale=None, skip_seq_scaling=False,
                    seq_samp_type=th.seqSampleType(DNA_SAMP_TYPE, False)):
    """Identify raw signal to genome sequence assignment (adaptive banded DP) -- same
    arguments, return value and TomboError messages as tombo.resquiggle.resquiggle_read."""
    if all_raw_signal is not None:
        map_res = map_res._replace(raw_signal=all_raw_signal)
    if map_res.raw_signal is None:
        raise th.TomboError(errors.MESSAGES[21])
    res = resquiggle_batch(
        [map_res], std_ref, rsqgl_params, outlier_thresh=outlier_thresh,
        max_raw_cpts=max_raw_cpts, min_event_to_seq_ratio=min_event_to_seq_ratio,
        const_scale=const_scale, skip_seq_scaling=skip_seq_scaling,
        seq_samp_type=seq_samp_type)[0]
    if isinstance(res, Exception):
        raise res    # (a batch of one rewinds the RNG itself when the read failed before rescaling)
    return res


# ---------------------------------------------------------------------------------------------
# The reference's public per-stage API (tombo/resquiggle.py:63-67), same names and signatures.
# Each runs the corresponding stage(s) of the batch engine on a batch of one, injecting the
# stage's inputs instead of recomputing the earlier stages (include/tombo_amd.h, "stepwise
# execution").
def _set_model(eng, std_ref):
    eng.ensure_model(std_ref)


def _status_or_raise(eng):
    st = int(eng.get(_native.GET_STATUS)[0])
    errors.raise_for_status(st)


class _LevelsOnly(object):
    """stand-in model for stages that are handed expected levels instead of a sequence"""
    # K=2: the reference's own sequence trimming needs a non-empty k-mer tail (resquiggle.py:976)
    kmer_width, central_pos = 2, 0
    level_means = np.zeros(16)
    level_sds = np.ones(16)


def segment_signal(map_res, num_events, rsqgl_params, outlier_thresh=None, const_scale=None):
    """Normalize and segment raw signal into `num_events` events (resquiggle.py:1057-1120).
    Returns (valid_cpts, norm_signal, scaleValues)."""
    eng = get_engine()
    if eng.kmer_width is None:
        _set_model(eng, _LevelsOnly)
    raw = np.ascontiguousarray(map_res.raw_signal, dtype=np.float64)
    sv = map_res.scale_values
    sv_in = sv_flags = None
    if sv is not None:
        has_lims = sv.lower_lim is not None and sv.upper_lim is not None
        sv_in = np.array([[sv.shift, sv.scale, sv.lower_lim if has_lims else 0.0,
                           sv.upper_lim if has_lims else 0.0]])
        sv_flags = np.array([1 | (2 if has_lims else 0)], np.int32)
    stalls = map_res.stall_ints
    K = eng.kmer_width
    eng.set_num_events([int(num_events)])
    eng.upload(_native.make_params(rsqgl_params),
               _native.make_opts(outlier_thresh=outlier_thresh, const_scale=const_scale),
               [raw], [np.zeros(K + 1, np.uint8)], sv_in=sv_in, sv_flags=sv_flags,
               stall_ints=[stalls] if stalls is not None and len(stalls) else None)
    eng.run_stages(_native.STAGE_SEGMENT, _native.STAGE_SEGMENT)
    _status_or_raise(eng)
    n = int(eng.get(_native.GET_N_CPTS)[0])
    cpts = eng.get(_native.GET_VALID_CPTS)[:n].copy()
    norm = eng.get(_native.GET_SEG_NORM)[:raw.shape[0]].copy()
    s = eng.get(_native.GET_SEG_SV)[0]
    use_sv = sv is not None or (bool(rsqgl_params.use_t_test_seg) and const_scale is None)
    have_lims = (sv is None and (outlier_thresh is not None)) or \
        (sv is not None and sv.lower_lim is not None and sv.upper_lim is not None)
    return cpts, norm, th.scaleValues(
        float(s[0]), float(s[1]), float(s[2]) if have_lims else None,
        float(s[3]) if have_lims else None, None if use_sv else outlier_thresh)


def _upload_events(eng, rsqgl_params, opts, valid_cpts, event_means, seq_codes):
    valid_cpts = np.ascontiguousarray(valid_cpts, dtype=np.int64)
    event_means = np.ascontiguousarray(event_means, dtype=np.float64)
    n_cpts = event_means.shape[0] + 1
    if valid_cpts.shape[0] != n_cpts:
        raise ValueError('valid_cpts must have one more entry than event_means')
    w = int(rsqgl_params.running_stat_width)
    n_raw = max(int(valid_cpts[-1]) + 1, 4 * w + 2)
    eng.set_num_events([n_cpts])
    eng.upload(_native.make_params(rsqgl_params), opts, [np.zeros(n_raw)], [seq_codes])
    eng.put(_native.PUT_VALID_CPTS, valid_cpts, per_read=[n_cpts])
    eng.put(_native.PUT_EVENT_MEANS, event_means)


def find_adaptive_base_assignment(
        valid_cpts, event_means, rsqgl_params, std_ref, genome_seq, start_clip_bases=None,
        start_clip_params=None, seq_samp_type=th.seqSampleType(DNA_SAMP_TYPE, False),
        reg_id=None):
    """Align expected signal levels to observed events with the adaptive banded DP
    (resquiggle.py:866-1050).  Returns :class:`dpResults`."""
    if start_clip_bases is not None:
        raise NotImplementedError('start-clip based start discovery (USE_START_CLIP_BASES) is '
                                  'off in the reference and not part of this engine')
    eng = get_engine()
    _set_model(eng, std_ref)
    opts = _native.make_opts(
        sig_match_thresh=None if seq_samp_type is None else SIG_MATCH_THRESH[seq_samp_type.name])
    _upload_events(eng, rsqgl_params, opts, valid_cpts, event_means, ts.encode_seq(genome_seq))
    eng.run_stages(_native.STAGE_REF_LEVELS, _native.STAGE_ASSIGN)
    _status_or_raise(eng)
    K, cp = std_ref.kmer_width, std_ref.central_pos
    return th.dpResults(
        read_start_rel_to_raw=int(eng.get(_native.GET_DP_READ_START)[0]),
        segs=eng.get(_native.GET_DP_SEGS).copy(), ref_means=eng.get(_native.GET_REF_MEANS).copy(),
        ref_sds=eng.get(_native.GET_REF_SDS).copy(),
        genome_seq=genome_seq[cp:len(genome_seq) - (K - cp - 1)])


def _upload_levels(eng, rsqgl_params, opts, event_means, r_ref_means, r_ref_sds):
    _set_model(eng, _LevelsOnly)
    event_means = np.ascontiguousarray(event_means, dtype=np.float64)
    n_ev = event_means.shape[0]
    B = len(r_ref_means)
    _upload_events(eng, rsqgl_params, opts, np.arange(n_ev + 1, dtype=np.int64), event_means,
                   np.zeros(B + 1, np.uint8))
    eng.put(_native.PUT_REF_MEANS, np.ascontiguousarray(r_ref_means, dtype=np.float64))
    eng.put(_native.PUT_REF_SDS, np.ascontiguousarray(r_ref_sds, dtype=np.float64))


def find_seq_start_in_events(event_means, r_ref_means, r_ref_sds, rsqgl_params, num_bases,
                             num_events, seq_samp_type=None, reg_id=None):
    """Most probable start of the expected levels within the events (resquiggle.py:685-752).
    Returns (start_loc, events_per_base)."""
    if event_means.shape[0] < num_events + num_bases:
        raise th.TomboError(errors.MESSAGES[3])
    if r_ref_means.shape[0] < num_bases:
        raise th.TomboError(errors.MESSAGES[4])
    eng = get_engine()
    p = rsqgl_params._replace(start_n_bases=int(num_bases), start_bw=int(num_events),
                              start_save_bw=int(num_events))
    opts = _native.make_opts(
        sig_match_thresh=None if seq_samp_type is None else SIG_MATCH_THRESH[seq_samp_type.name])
    _upload_levels(eng, p, opts, event_means, r_ref_means, r_ref_sds)
    eng.run_stages(_native.STAGE_START, _native.STAGE_START)
    _status_or_raise(eng)
    fail = int(eng.get(_native.GET_START_FAIL)[0])
    if fail != 0:
        # the first try failed (poor score / invalid path); inside resquiggle_read the engine
        # goes on to the retry or the static fallback, stand-alone it is the reference's exception
        raise th.TomboError(errors.MESSAGES[fail])
    start = eng.get(_native.GET_START)[0]
    return int(start[0]), float(start[1])


def find_static_base_assignment(event_means, r_ref_means, r_ref_sds, rsqgl_params, reg_id=None):
    """Whole-read static-band assignment for short reads (resquiggle.py:547-600).  Returns the
    event index of every base start (`read_tb`)."""
    eng = get_engine()
    _upload_levels(eng, rsqgl_params, _native.make_opts(), event_means, r_ref_means, r_ref_sds)
    eng.put(_native.PUT_START_STATE, np.zeros(1), per_read=[4])
    eng.run_stages(_native.STAGE_ASSIGN, _native.STAGE_ASSIGN)
    _status_or_raise(eng)
    return eng.get(_native.GET_READ_TB).copy()


def resolve_skipped_bases_with_raw(dp_res, norm_signal, rsqgl_params, max_raw_cpts=MAX_RAW_CPTS,
                                   del_fix_window=DEL_FIX_WINDOW, max_del_fix_window=MAX_DEL_FIX_WINDOW,
                                   extra_sig_factor=EXTRA_SIG_FACTOR):
    """Raw-signal DP over the windows around skipped bases (resquiggle.py:402-540).  Returns
    the resolved segment boundaries."""
    del_fix_window = DEL_FIX_WINDOW if del_fix_window is None else int(del_fix_window)
    max_del_fix_window = MAX_DEL_FIX_WINDOW if max_del_fix_window is None else int(max_del_fix_window)
    extra_sig_factor = EXTRA_SIG_FACTOR if extra_sig_factor is None else float(extra_sig_factor)
    eng = get_engine()
    _set_model(eng, _LevelsOnly)
    norm = np.ascontiguousarray(norm_signal, dtype=np.float64)
    segs = np.ascontiguousarray(dp_res.segs, dtype=np.int64)
    B = segs.shape[0] - 1
    w = int(rsqgl_params.running_stat_width)
    pad = max(0, 4 * w + 2 - norm.shape[0])
    eng.set_num_events([2])
    eng.upload(_native.make_params(rsqgl_params),
               _native.make_opts(max_raw_cpts=max_raw_cpts, del_fix_window=del_fix_window,
                                 max_del_fix_window=max_del_fix_window, extra_sig_factor=extra_sig_factor),
               [np.concatenate([norm, np.zeros(pad)])], [np.zeros(B + 1, np.uint8)])
    eng.put(_native.PUT_NORM, np.concatenate([norm, np.zeros(pad)]))
    eng.put(_native.PUT_REF_MEANS, np.ascontiguousarray(dp_res.ref_means, dtype=np.float64))
    eng.put(_native.PUT_REF_SDS, np.ascontiguousarray(dp_res.ref_sds, dtype=np.float64))
    eng.put(_native.PUT_DP_SEGS, segs, per_read=[0, norm.shape[0]])
    eng.run_stages(_native.STAGE_SKIP, _native.STAGE_SKIP)
    _status_or_raise(eng)
    return eng.get(_native.GET_SEGS).copy()


# ---- the caller's loop: _resquiggle_worker, resquiggle.py:1488-1602 ------
def adjust_map_res(map_res, seq_samp_type):
    """What `_resquiggle_worker.adjust_map_res` (resquiggle.py:1506-1530) does to a mapped read
    before resquiggling: RNA signal is flipped into 5'->3' order and stall intervals are
    detected on it (`COLLAPSE_RNA_STALLS`); DNA is passed through (`COLLAPSE_DNA_STALLS` and
    `USE_START_CLIP_BASES` are off in the reference, `TRIM_RNA_ADAPTER` too)."""
    from ._default_parameters import COLLAPSE_RNA_STALLS, COLLAPSE_DNA_STALLS
    if seq_samp_type.name == RNA_SAMP_TYPE:
        map_res = map_res._replace(raw_signal=map_res.raw_signal[::-1])
    if (COLLAPSE_RNA_STALLS and seq_samp_type.name == RNA_SAMP_TYPE) or \
            (COLLAPSE_DNA_STALLS and seq_samp_type.name == DNA_SAMP_TYPE):
        map_res = map_res._replace(stall_ints=ts.identify_stalls(map_res.raw_signal))
    return map_res


def _run_iters(map_results, idx, std_ref, params, outlier_thresh, const_scale, skip_seq_scaling,
               seq_samp_type, max_scaling_iters, engine, n_passes, prep):
    """run_rsqgl_iters (resquiggle.py:1492-1504) for the reads `idx`, round by round"""
    res = dict(zip(idx, resquiggle_batch(
        [map_results[i] for i in idx], std_ref, params, outlier_thresh, const_scale=const_scale,
        skip_seq_scaling=skip_seq_scaling, seq_samp_type=seq_samp_type, engine=engine, **prep)))
    for i in idx:
        n_passes[i] += 1
    n_iters = 1
    while n_iters < max_scaling_iters:
        again = [i for i in idx if not isinstance(res[i], Exception) and
                 res[i].norm_params_changed]
        if not again:
            break
        # the re-runs take the fitted scale values and the un-normalised signal; const_scale /
        # skip_seq_scaling are NOT forwarded (resquiggle.py:1499-1502)
        sub = resquiggle_batch(
            [map_results[i]._replace(scale_values=res[i].scale_values) for i in again], std_ref,
            params, outlier_thresh, all_raw_signals=[map_results[i].raw_signal for i in again],
            seq_samp_type=seq_samp_type, engine=engine, **prep)
        for i, r in zip(again, sub):
            res[i] = r
            n_passes[i] += 1
        n_iters += 1
    return res


def resquiggle_batch_iters(map_results, std_ref, rsqgl_params, save_params=None,
                           outlier_thresh=None, const_scale=None, skip_seq_scaling=False,
                           seq_samp_type=th.seqSampleType(DNA_SAMP_TYPE, False),
                           max_scaling_iters=None, engine=None, return_passes=False,
                           device_prep=False, subsample_seed=None, rng_order='round_major'):
    """The per-read loop of `_resquiggle_worker` (resquiggle.py:1578-1589) over a batch.

    Every read is resquiggled; while `norm_params_changed` it is re-run with the fitted
    `scale_values` (at most `max_scaling_iters` passes, default MAX_SCALING_ITERS = 3); a read
    that fails for any reason is started over with `save_params` (the wide "save" bandwidth,
    `load_resquiggle_parameters(..., use_save_bandwidth=True)`).  `map_results` are what
    `adjust_map_res` returns.  Returns per read a `resquiggleResults` or the exception of the
    last attempt.  Reads are grouped into rounds (all first passes, then all second passes, ...),
    so the Theil-Sen subsamples are drawn from numpy's global RNG in round-major order instead
    of the worker's read-major order; each individual pass is the same computation as
    `resquiggle_read` with the subsample it was handed.

    `rng_order='read_major'`: the reads are taken one at a time, exactly the worker's sequence of
    `resquiggle_read` calls, so that under a seeded numpy RNG every read gets the subsamples the
    reference's loop would have drawn for it (bit-identical `raw_signal` / `scale_values`; pinned on
    a loop recorded from the live reference, tests/golden/loop_dna.npz).  Which pass of which read
    draws next depends on the results of the passes before it, so this order cannot be batched;
    it is the parity mode, `round_major` the throughput mode: another draw means a fitted scale that
    differs in the third digit, and in a re-run pass that can move a handful of event boundaries
    (3 of 1801 on one read of the recorded loop) -- the spread the reference shows between two seeds.

    `device_prep`: `map_results` are the mapped reads as `_io_and_map_read` left them (RNA signal
    in acquisition order, no `stall_ints`) and `adjust_map_res` -- the flip and
    `ts.identify_stalls` -- happens on the device inside every pass.  (The device computes the stall
    metric in float64 whatever the sample type; the reference's worker hands `identify_stalls` the
    file's int16 samples and gets an int16-truncated metric: intervals within one unit of the
    threshold or at the signal's edges can differ -- INTEGRATION.md section 6.  Pass `stall_ints`
    on the reads, without `device_prep`, for the reference's own intervals.)
    """
    from ._default_parameters import MAX_SCALING_ITERS
    if max_scaling_iters is None:
        max_scaling_iters = MAX_SCALING_ITERS
    n = len(map_results)
    n_passes = [0] * n
    prep = dict(subsample_seed=subsample_seed)
    if device_prep:
        from ._default_parameters import COLLAPSE_RNA_STALLS, STALL_PARAMS
        rna = seq_samp_type is not None and seq_samp_type.name == RNA_SAMP_TYPE
        prep.update(reverse_raw=rna, stall_params=th.stallParams(**STALL_PARAMS)
                    if rna and COLLAPSE_RNA_STALLS else None)
    if rng_order == 'read_major':
        res = {}
        for i in range(n):
            res.update(_run_iters(map_results, [i], std_ref, rsqgl_params, outlier_thresh, const_scale,
                                  skip_seq_scaling, seq_samp_type, max_scaling_iters, engine, n_passes, prep))
            if isinstance(res[i], Exception) and save_params is not None:
                res.update(_run_iters(map_results, [i], std_ref, save_params, outlier_thresh, const_scale,
                                      skip_seq_scaling, seq_samp_type, max_scaling_iters, engine,
                                      n_passes, prep))
        out = [res[i] for i in range(n)]
        return (out, n_passes) if return_passes else out
    if rng_order != 'round_major':
        raise ValueError("rng_order is 'round_major' or 'read_major'")
    res = _run_iters(map_results, list(range(n)), std_ref, rsqgl_params, outlier_thresh,
                     const_scale, skip_seq_scaling, seq_samp_type, max_scaling_iters, engine,
                     n_passes, prep)
    failed = [i for i in range(n) if isinstance(res[i], Exception)]
    if failed and save_params is not None:
        res.update(_run_iters(map_results, failed, std_ref, save_params, outlier_thresh,
                              const_scale, skip_seq_scaling, seq_samp_type, max_scaling_iters,
                              engine, n_passes, prep))
    out = [res[i] for i in range(n)]
    return (out, n_passes) if return_passes else out


def resquiggle_batch_events(map_results, std_ref, rsqgl_params, outlier_thresh=None,
                            compute_sd=True, engine=None, **kw):
    """`resquiggle_batch` plus, per successful read, the Events table the reference writes to the
    FAST5 file (`tombo_helper.write_new_fast5_group`, tombo_helper.py:2341-2362): the per-base
    statistics are computed on the device from the batch's resident signal and boundaries
    (`tba_batch_base_stats`), nothing is uploaded again.  Returns (results, tables); tables[i] is
    None where results[i] is an exception."""
    eng = get_engine() if engine is None else engine
    results = resquiggle_batch(map_results, std_ref, rsqgl_params, outlier_thresh, engine=eng, **kw)
    means, stds = eng.base_stats()
    tables = []
    for i, res in enumerate(results):
        if isinstance(res, Exception):
            tables.append(None)
            continue
        a, b = int(eng.ref_off[i]), int(eng.ref_off[i + 1])
        tables.append(th.events_table(res, means[a:b], stds[a:b] if compute_sd else None))
    return results, tables


def batch_de_novo_stats(fm_offset=1, starts=None, strands=None, engine=None):
    """De novo test statistic (`ts.compute_de_novo_read_stats`, tombo_stats.py:3771-3873, whole
    read as the region) of every read of the batch the engine has just finished -- per-base
    means, expected levels and p-values never leave the device until the result is copied back
    (`tba_batch_de_novo_stats`).  Returns per read (pvals, genomic positions), None for reads
    that failed or are too short to test.  `starts[i]` / `strands[i]`: mapped start and strand
    (default 0 / '+'); minus-strand results are flipped into genomic order like the
    reference's."""
    from ._default_parameters import SMALLEST_PVAL
    eng = get_engine() if engine is None else engine
    pv = eng.de_novo_stats(fm_offset, SMALLEST_PVAL)
    st = eng.download(want_norm=False)['status']
    K = eng.kmer_width
    cp = getattr(eng, '_model_key', None)[1] if getattr(eng, '_model_key', None) else None
    if cp is None:
        raise RuntimeError('the engine model was not set through ensure_model')
    dn = K - cp - 1
    out = []
    for i in range(eng.n):
        B = int(eng.B[i])
        if st[i] != 0 or B - cp - dn < max(1, 2 * fm_offset + 1):
            out.append(None)
            continue
        a = int(eng.ref_off[i])
        p = pv[a + cp:a + B - dn].copy()
        s0 = 0 if starts is None else int(starts[i])
        minus = strands is not None and strands[i] == '-'
        lag_b = dn if minus else cp
        out.append((p[::-1].copy() if minus else p, np.arange(s0 + lag_b, s0 + lag_b + p.shape[0])))
    return out
