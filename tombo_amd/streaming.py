"""Streaming resquiggle: batches flow host -> GPU -> host through a few engine slots per GPU.

The reference streams reads through three kinds of processes -- a FAST5 reader, N
`_resquiggle_worker`s and a writer -- connected by queues (tombo/resquiggle.py:1488-1602,
1859-1950), so reading read k+1, resquiggling read k and writing read k-1 overlap.  Here the unit
is a batch and the overlap is done by the hardware: every slot is one engine (its own HIP stream
and device buffers); a batch is uploaded, run and downloaded in stream order on its slot, and
because different slots use different streams the copy engines move batch N+1 in and batch N-1
out while the compute units work on batch N.  With page-locked host arrays (`PinnedArray`) the
transfers are DMA and nothing blocks the host but `sync`.

    pipe = StreamPipeline(std_ref, rsqgl_params, n_slots=2, outlier_thresh=5.0)
    for res in pipe.run(batches):          # batches: iterable of ReadBatch
        ...res.results['status'], res.segs_of(i)...

Compact traffic: int16 raw in (2 bytes / sample, the FAST5 `Signal` dtype), and out the 64-byte
record per read + int32 base boundaries (4 bytes / base); the normalised signal is optional
(`want_norm`) -- Tombo itself stores only scale values and boundaries and re-normalises the raw
signal when a resquiggled read is loaded (tombo_helper.py:2341-2460, tombo_stats.py:482-573).
"""
import time

import numpy as np

from . import _native
from . import tombo_helper as th
from ._default_parameters import (
    MAX_RAW_CPTS, MIN_EVENT_TO_SEQ_RATIO, SIG_MATCH_THRESH, DNA_SAMP_TYPE,
    MAX_POINTS_FOR_THEIL_SEN)

__all__ = ['ReadBatch', 'BatchResults', 'StreamPipeline', 'ReadFeeder']


class ReadBatch(object):
    """One batch as flat CSR host arrays: raw (int16 / float32 / float64) + raw_off[n+1],
    seq (uint8 codes 0..3, genome_seq with k-mer flanks) + seq_off[n+1]; optional samp_ind
    [n, 1000] (Theil-Sen subsamples, required for reads longer than 1000 bases unless sequence
    rescaling is skipped), stall_ints [m, 2] + stall_off[n+1] (RNA).  `tag` travels with it."""

    def __init__(self, raw, raw_off, seq, seq_off, samp_ind=None, stall_ints=None,
                 stall_off=None, tag=None):
        self.raw, self.raw_off, self.seq, self.seq_off = raw, raw_off, seq, seq_off
        self.samp_ind, self.stall_ints, self.stall_off, self.tag = samp_ind, stall_ints, stall_off, tag
        self.n = int(np.asarray(raw_off).shape[0]) - 1

    @classmethod
    def from_lists(cls, raws, seqs, samp_inds=None, stalls=None, tag=None, pinned=False):
        """pack per-read arrays; with `pinned` the big arrays are built in page-locked memory
        (returned batch keeps the PinnedArray objects alive)"""
        n = len(raws)
        raw_off = np.zeros(n + 1, np.int64)
        np.cumsum([len(r) for r in raws], out=raw_off[1:])
        seq_off = np.zeros(n + 1, np.int64)
        np.cumsum([len(s) for s in seqs], out=seq_off[1:])
        dts = set(np.asarray(r).dtype for r in raws)
        dt = next(iter(dts)) if len(dts) == 1 and next(iter(dts)) in _native.RAW_DTYPES \
            else np.dtype(np.float64)
        keep = []
        if pinned:
            pr = _native.PinnedArray(int(raw_off[-1]), dt)
            ps = _native.PinnedArray(int(seq_off[-1]), np.uint8)
            keep = [pr, ps]
            raw, seq = pr.a, ps.a
        else:
            raw, seq = np.empty(int(raw_off[-1]), dt), np.empty(int(seq_off[-1]), np.uint8)
        for i in range(n):
            raw[raw_off[i]:raw_off[i + 1]] = raws[i]
            seq[seq_off[i]:seq_off[i + 1]] = seqs[i]
        si = None
        if samp_inds is not None and any(s is not None for s in samp_inds):
            # A read of more than 1000 bases (len(seq) - kmer_width + 1: the engine knows K, this
            # function does not) needs its row.  Rows without one are poisoned with -1: the kernel
            # rejects a negative index (TBA_INTERNAL for that read) instead of fitting a line
            # through whatever the row held -- it cannot tell a missing subsample from a given one.
            if pinned:
                psi = _native.PinnedArray((n, 1000), np.int64)
                keep.append(psi)
                si = psi.a
                si[:] = -1
            else:
                si = np.full((n, 1000), -1, np.int64)
            for i, s in enumerate(samp_inds):
                if s is not None:
                    si[i] = s
        st = sto = None
        if stalls is not None and any(s is not None and len(s) for s in stalls):
            sto = np.zeros(n + 1, np.int64)
            np.cumsum([0 if s is None else len(s) for s in stalls], out=sto[1:])
            st = np.array([[int(a), int(b)] for s in stalls if s is not None for a, b in s],
                          dtype=np.int64).reshape(-1, 2)
        b = cls(raw, raw_off, seq, seq_off, si, st, sto, tag)
        b._keep = keep
        return b


class BatchResults(object):
    """Outputs of one batch.  `results`: structured array (_native.RESULT_DTYPE) per read;
    `segs`: flat boundaries (int32 or int64), read i at seg_off[i]:seg_off[i+1]; `norm`: flat
    normalised signal (float64, same CSR as raw; first norm_len[i] entries valid) or None.
    The arrays live in the pipeline's page-locked buffers: valid until the pipeline hands out the
    results of `n_slots` later batches -- copy what must outlive that."""

    def __init__(self, tag, n, results, segs, seg_off, norm, raw_off, stage_ms):
        self.tag, self.n, self.results, self.segs, self.seg_off = tag, n, results, segs, seg_off
        self.norm, self.raw_off, self.stage_ms = norm, raw_off, stage_ms

    def segs_of(self, i):
        return self.segs[self.seg_off[i]:self.seg_off[i + 1]]

    def norm_of(self, i):
        a = int(self.raw_off[i])
        return self.norm[a:a + int(self.results['norm_len'][i])]

    def scale_values(self, i, outlier_thresh=None):
        r = self.results[i]
        lo = None if np.isnan(r['lower_lim']) else float(r['lower_lim'])
        hi = None if np.isnan(r['upper_lim']) else float(r['upper_lim'])
        return th.scaleValues(float(r['shift']), float(r['scale']), lo, hi, outlier_thresh)


class _Slot(object):
    def __init__(self, device):
        self.eng = _native.Engine(device)
        self.pending = None       # (batch, output set index)
        self.seq = 0              # submission number of the batch it holds
        self.outs = [dict(), dict()]
        self.flip = 0

    def out_arrays(self, which, n, n_segs, n_raw, segs_dtype, want_norm):
        o = self.outs[which]

        def grow(key, count, dt):
            pa = o.get(key)
            if pa is None or pa.a.shape[0] < count:
                if pa is not None:
                    pa.close()
                pa = _native.PinnedArray(int(count * 1.125) + 16, dt)
                o[key] = pa
            return pa.a
        res = grow('res', n, _native.RESULT_DTYPE)
        segs = grow('segs', n_segs, segs_dtype)
        norm = grow('norm', n_raw, np.float64) if want_norm else None
        return res, segs, norm


class StreamPipeline(object):
    """`n_slots` engines on one GPU, used round-robin; see the module docstring."""

    def __init__(self, std_ref, rsqgl_params, n_slots=2, device=None, outlier_thresh=None,
                 seq_samp_type=th.seqSampleType(DNA_SAMP_TYPE, False), const_scale=None,
                 skip_seq_scaling=False, max_raw_cpts=MAX_RAW_CPTS,
                 min_event_to_seq_ratio=MIN_EVENT_TO_SEQ_RATIO, want_norm=False,
                 segs_dtype=np.int32, reverse_raw=False, stall_params=None, subsample_seed=None,
                 in_order=True, serial_compute=False):
        from . import resquiggle as rq
        if device is None:
            device = rq.default_device()
        self.device = device
        self.slots = [_Slot(device) for _ in range(max(1, int(n_slots)))]
        for s in self.slots:
            s.eng.ensure_model(std_ref)
            s.eng.set_sharing(len(self.slots))
        self.params = _native.make_params(rsqgl_params)
        self.opts = _native.make_opts(
            outlier_thresh=outlier_thresh, const_scale=const_scale,
            skip_seq_scaling=skip_seq_scaling,
            sig_match_thresh=None if seq_samp_type is None else SIG_MATCH_THRESH[seq_samp_type.name],
            max_raw_cpts=max_raw_cpts, min_event_to_seq_ratio=min_event_to_seq_ratio,
            skip_norm_out=not want_norm, reverse_raw=reverse_raw, stall_params=stall_params,
            subsample_seed=subsample_seed)
        self.subsample_seed = subsample_seed
        self.want_norm = bool(want_norm)
        self.segs_dtype = np.dtype(segs_dtype)
        assert self.segs_dtype in (np.dtype(np.int32), np.dtype(np.int64))
        self._next = 0
        self.n_submitted = 0
        # in_order=False: a batch goes to ANY slot that is free or has finished (polled, never
        # blocking while one is idle) and results come back as they complete (match them by `tag`).
        # With ragged jobs -- a batch of 100 kb reads runs ten times longer than its neighbours --
        # round-robin slots make the host wait behind the long batch while other slots sit idle.
        self.in_order = bool(in_order)
        self._seq = 0
        # serial_compute: the kernel sequence of a batch starts when the previous batch's has
        # finished (tba_batch_wait_for) -- uploads and downloads still overlap compute, but the
        # kernels of two batches do not interleave.  Measured at cfg2 over 24 batches: 2 slots back
        # to back 93.3 k reads/s, 3 slots back to back 91.0 k, 3 slots interleaved 93.0 k, 2 slots
        # interleaved 84.5 k -- with three slots it makes no difference, so it stays an option.
        self.serial_compute = bool(serial_compute)
        self._last_eng = None

    def reserve(self, max_reads, max_segs, max_raw=0):
        """Size the page-locked output arrays of every slot for batches of up to `max_reads` reads,
        `max_segs` boundaries (bases + reads) and `max_raw` samples now.  Growing one later means
        hipHostFree, which waits for ALL work on the device: with ragged batches landing on any slot
        (`in_order=False`) that stalls the pipeline behind the longest batch in flight."""
        for s in self.slots:
            for which in (0, 1):
                s.out_arrays(which, int(max_reads), int(max_segs), int(max_raw), self.segs_dtype, self.want_norm)

    def _finish(self, slot):
        batch, which, res, segs, norm = slot.pending
        slot.pending = None
        eng = slot.eng
        eng.sync()
        rel = getattr(batch, 'release', None)
        if rel is not None:   # the batch's staging buffers (ReadFeeder) are free again
            rel()
        n_segs = int(eng.seg_off[-1])
        return BatchResults(batch.tag, eng.n, res[:eng.n], segs[:n_segs], eng.seg_off,
                            None if norm is None else norm[:eng.n_raw_total], eng.raw_off,
                            eng.get(_native.GET_KERNEL_MS))

    def submit(self, batch):
        """Enqueue one batch (upload, kernels, download -- nothing here waits for them) on the
        next slot.  If that slot still held an earlier batch, that one is finished first and its
        results are returned (else None)."""
        if self.in_order:
            slot = self.slots[self._next]
            self._next = (self._next + 1) % len(self.slots)
        else:
            slot = next((s for s in self.slots if s.pending is None), None)
            while slot is None:   # all busy: take the first one to finish, whichever it is
                slot = next((s for s in self.slots if not s.eng.query()), None)
                if slot is None:
                    time.sleep(0.0002)
        if batch.samp_ind is None and self.subsample_seed is None and not self.opts.skip_seq_scaling and \
                int(np.max(np.diff(batch.seq_off), initial=0)) - (slot.eng.kmer_width or 1) + 1 > MAX_POINTS_FOR_THEIL_SEN:
            # (on the device this is the generic TBA_INTERNAL status of every long read; say it here)
            raise ValueError('batch %r has reads longer than %d bases but no Theil-Sen subsamples: pass '
                             'samp_inds, or subsample_seed to the pipeline' % (batch.tag, MAX_POINTS_FOR_THEIL_SEN))
        if batch.samp_ind is not None and self.subsample_seed is None and not self.opts.skip_seq_scaling:
            # per read, with the model's k-mer width: a read of more than 1000 bases whose row was
            # left without a subsample (poisoned with -1 by from_lists / ReadFeeder.pack)
            n_bases = np.diff(batch.seq_off) - (slot.eng.kmer_width or 1) + 1
            si0 = np.asarray(batch.samp_ind).reshape(batch.n, -1)[:, 0]
            if np.any((n_bases > MAX_POINTS_FOR_THEIL_SEN) & (si0 < 0)):
                raise ValueError('batch %r: no Theil-Sen subsample for a read longer than %d bases' %
                                 (batch.tag, MAX_POINTS_FOR_THEIL_SEN))
        done = self._finish(slot) if slot.pending is not None else None
        self._seq += 1
        slot.seq = self._seq
        eng = slot.eng
        if self.subsample_seed is not None:   # another key for every batch of the job
            self.opts.subsample_seed = (int(self.subsample_seed) + 0x9e3779b97f4a7c15 * self.n_submitted) \
                & 0xffffffffffffffff
            # (a batch that names its own key -- a job sharded over ranks: the draw must not depend on
            # which rank got the batch, or when)
            if getattr(batch, 'subsample_seed', None) is not None:
                self.opts.subsample_seed = int(batch.subsample_seed) & 0xffffffffffffffff
        eng.upload_packed(self.params, self.opts, batch.raw, batch.raw_off, batch.seq,
                          batch.seq_off, samp_ind=batch.samp_ind, stall_ints=batch.stall_ints,
                          stall_off=batch.stall_off)
        if self.serial_compute and self._last_eng is not None and hasattr(eng, 'wait_for'):
            eng.wait_for(self._last_eng)
        eng.enqueue()
        self._last_eng = eng
        which = slot.flip
        slot.flip ^= 1
        res, segs, norm = slot.out_arrays(which, eng.n, int(eng.seg_off[-1]), eng.n_raw_total,
                                          self.segs_dtype, self.want_norm)
        eng.download_async(results=res,
                           segs32=segs if self.segs_dtype == np.int32 else None,
                           segs64=segs if self.segs_dtype == np.int64 else None, norm=norm)
        slot.pending = (batch, which, res, segs, norm)
        self.n_submitted += 1
        return done

    def flush(self):
        """finish everything in flight, oldest first"""
        out = []
        order = [self.slots[(self._next + k) % len(self.slots)] for k in range(len(self.slots))] \
            if self.in_order else sorted(self.slots, key=lambda s: s.seq)
        for slot in order:
            if slot.pending is not None:
                out.append(self._finish(slot))
        return out

    def run(self, batches):
        """generator over the results of an iterable of batches, in submission order"""
        for b in batches:
            done = self.submit(b)
            if done is not None:
                yield done
        for done in self.flush():
            yield done

    def close(self):
        for s in self.slots:
            if s.pending is not None:
                s.eng.sync()
                s.pending = None
            for o in s.outs:
                for pa in o.values():
                    pa.close()
                o.clear()
            s.eng.close()


class ReadFeeder(object):
    """Per-read arrays -> `ReadBatch`es in reusable page-locked staging, one batch ahead.

    The reader of the reference's worker pool (`_io_and_map_read`, resquiggle.py:1385-1486) hands
    over one read at a time; a batch engine wants flat CSR buffers.  `pack(raws, seqs)` copies the
    reads of one batch into the next of `n_stages` staging sets with native threads
    (`tba_pack_reads`, GIL released); `prefetch` does the same on a helper thread while the caller
    submits the previous batch.  A staging set is busy from `pack` until the pipeline has finished
    the batch packed into it (`StreamPipeline._finish` calls the batch's `release`); batches may
    finish in any order, and a new set is allocated when all are busy (`n_slots + 2` in steady
    state: `n_slots` in flight, one being submitted, one being packed).
    """

    def __init__(self, n_slots=3, reverse=False, n_threads=None):
        import threading
        from concurrent.futures import ThreadPoolExecutor
        self.stages = [_native.PinnedStage() for _ in range(int(n_slots) + 2)]
        self._busy = [False] * len(self.stages)
        self._lock = threading.Lock()
        self.reverse, self.n_threads = bool(reverse), n_threads
        self._ex = ThreadPoolExecutor(1)
        self._pending = None

    def _take_stage(self):
        with self._lock:
            for k, b in enumerate(self._busy):
                if not b:
                    self._busy[k] = True
                    return k
            self.stages.append(_native.PinnedStage())
            self._busy.append(True)
            return len(self.stages) - 1

    def _release_stage(self, k):
        with self._lock:
            self._busy[k] = False

    def pack(self, raws, seqs, samp_inds=None, stalls=None, tag=None):
        """one batch, packed now; seqs: str / bytes of ACGT"""
        k = self._take_stage()
        try:
            stage = self.stages[k]
            raw, raw_off, seq, seq_off, _ = _native.pack_reads(
                raws, seqs, reverse=self.reverse, stage=stage, n_threads=self.n_threads)
            si = None
            if samp_inds is not None:
                n = len(raws)
                si = stage.get('si', n * MAX_POINTS_FOR_THEIL_SEN, np.int64).reshape(n, MAX_POINTS_FOR_THEIL_SEN)
                for i, s in enumerate(samp_inds):
                    if s is not None:
                        si[i] = s
                    else:
                        # the staging buffer is reused: a row without a subsample must not keep an
                        # earlier batch's indices (the kernel rejects -1, see ReadBatch.from_lists)
                        si[i] = -1
            st, sto = _native.pack_stalls(stalls) if stalls is not None and \
                any(s is not None and len(s) for s in stalls) else (None, None)
        except BaseException:
            self._release_stage(k)  # (a failed pack must not leave the staging set busy for good)
            raise
        b = ReadBatch(raw, raw_off, seq, seq_off, si, st, sto, tag)
        b.release = lambda: self._release_stage(k)
        return b

    def prefetch(self, raws, seqs, **kw):
        """start packing a batch on the helper thread; `take()` returns it"""
        assert self._pending is None
        self._pending = self._ex.submit(self.pack, raws, seqs, **kw)

    def take(self):
        fut, self._pending = self._pending, None
        return None if fut is None else fut.result()

    def close(self):
        self.take()
        self._ex.shutdown()
        for st in self.stages:
            st.close()
