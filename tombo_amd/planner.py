"""Host-side batch planner: cuts a list of reads into device batches.

The reference hands reads to its workers one at a time (tombo/resquiggle.py:1859-1950), so read
length only matters to the worker that drew the read.  A batch engine has two more concerns:

* memory -- a batch occupies device buffers proportional to its samples, bases and DP cells
  (`tba_batch_footprint` is the exact figure the engine would allocate), so the list is cut
  where a budget would be exceeded instead of failing with TBA_E_NOMEM;
* divergence -- the banded DP runs one read per wavefront and workgroups are dispatched in
  index order, so reads are sorted by length (bases, then samples: the (ceil(B/1k), ceil(S/16k))
  buckets of SURVEY.md 8e, refined to a total order): a batch holds reads of similar length,
  longest first (longest-processing-time-first scheduling inside every launch), and the batches
  themselves come in order of decreasing read length (they hold about the same number of bytes,
  hence about the same work; what differs is the serial time of their longest read), so that a
  shared work queue hands out the batch with the longest critical path first.
"""
import ctypes as C

import numpy as np

from . import _native

__all__ = ['estimate_bytes', 'exact_bytes', 'plan_batches']


def _num_events(n_raw, n_bases, mean_obs_per_event, min_event_to_seq_ratio):
    # ts.compute_num_events (tombo_stats.py:1558-1574)
    return np.maximum(n_raw // int(mean_obs_per_event),
                      (n_bases * float(min_event_to_seq_ratio)).astype(np.int64))


def _mv_row_bytes(width):
    for cpl in (4, 5, 8, 12, 16, 24, 32, 48):
        if cpl * 64 >= width:
            return cpl * 16
    return ((int(width) + 255) // 256) * 64


def estimate_bytes(n_raw, seq_len, params, opts, kmer_width, raw_dtype=np.float64):
    """Per-read device bytes (vectorised restatement of the engine's buffer list; the planner's
    working figure -- every finished batch is checked against `exact_bytes`)."""
    S = np.asarray(n_raw, dtype=np.int64)
    L = np.asarray(seq_len, dtype=np.int64)
    B = np.maximum(L - int(kmer_width) + 1, 0)
    ne = _num_events(S, B, params.mean_obs_per_event, opts.min_event_to_seq_ratio)
    raw_b = np.dtype(raw_dtype).itemsize
    per_sample = raw_b + 8 + (0 if opts.skip_norm_out else 8) + 8 + 8 + 1
    if opts.detect_stalls:   # the detector's own scratch (it runs beside event detection)
        per_sample += 0.125 + (0 if (raw_b == 2 and opts.stall_window_size <= 1024) else 8)
    per_base = 8 * 3 + 4 + 4 + 8 * 3 + 24 + 8 + 4
    start_w = max(int(params.start_bw), int(params.start_save_bw))
    fixed = 512 + 32 + 8000 + 3072 * 8 + int(params.start_n_bases) * 8 + \
        (int(params.start_n_bases) + 1) * _mv_row_bytes(start_w) + 32768 * 8 + 64 + 24
    moves = (B + 1) * (_mv_row_bytes(int(params.bandwidth)) + 16) * 1.125   # (+ the centre strip, k_dp.h)
    return (S * per_sample + ne * 16 + L + B * per_base + fixed + moves) * 1.13


def exact_bytes(n_raw, seq_len, params, opts, kmer_width, raw_dtype=np.float64):
    """tba_batch_footprint: the bytes the engine allocates for this batch (host-only call)"""
    nr = np.ascontiguousarray(n_raw, dtype=np.int64)
    sl = np.ascontiguousarray(seq_len, dtype=np.int64)
    out = C.c_double(0)
    L = _native.lib()
    rc = L.tba_batch_footprint(C.byref(params), C.byref(opts), C.c_int64(int(kmer_width)),
                               C.c_int(_native.RAW_DTYPES[np.dtype(raw_dtype)]),
                               C.c_int64(nr.shape[0]), nr.ctypes.data_as(C.POINTER(C.c_int64)),
                               sl.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(out))
    if rc != 0:
        raise _native.EngineError('tba_batch_footprint failed (%d): %s' % (
            rc, L.tba_last_error().decode()))
    return out.value


MAX_READS = 16384   # reads per batch (the engine's hard limit is TBA_MAX_BATCH_READS = 65535)
_FIXED = 512 << 20  # arenas' constant slack (moves 64 MB + skip 256 MB + rounding)


def plan_batches(n_raw, seq_len, params, opts, kmer_width, mem_budget, raw_dtype=np.float64,
                 max_reads=MAX_READS, sort=True, tail_bases=None):
    """Cut reads 0..n-1 into batches.

    params / opts: `_native.Params` / `_native.Opts` of the job; mem_budget: device bytes one
    batch may occupy.  Returns a list of int64 index arrays; with `sort` each batch holds reads of
    similar length in descending order and the batches come in order of decreasing read length,
    without it the input order is kept and the list is only cut (callers that need results in input order
    scatter by the returned indices either way).  A single read that exceeds the budget on its own
    still gets a batch (the engine will report what it cannot do).

    tail_bases: reads with more bases than this go into batches of their own, in front.  The
    banded DP and the traceback of a read are serial chains (about 1.5 us per base for a lone
    wavefront: 0.3 s for 200 kb), so a batch is only done when its longest read is; a batch that
    mixes the tail with thousands of ordinary reads holds most of the machine's wave slots for that
    long and starves the batches running beside it (measured: co-running kernels 10x slower).  A
    tail batch of a few hundred reads occupies a few hundred wave slots; the ordinary batches run
    at full speed next to it and the tail batches of successive passes overlap each other.
    """
    S = np.asarray(n_raw, dtype=np.int64)
    L = np.asarray(seq_len, dtype=np.int64)
    n = S.shape[0]
    if n == 0:
        return []
    B = np.maximum(L - int(kmer_width) + 1, 0)
    if tail_bases is not None and sort:
        tail = np.flatnonzero(B > int(tail_bases))
        if 0 < tail.shape[0] < n:
            rest = np.flatnonzero(B <= int(tail_bases))
            head = plan_batches(S[tail], L[tail], params, opts, kmer_width, mem_budget, raw_dtype, max_reads)
            body = plan_batches(S[rest], L[rest], params, opts, kmer_width, mem_budget, raw_dtype, max_reads)
            return [tail[x] for x in head] + [rest[x] for x in body]
    order = np.lexsort((-S, -B)) if sort else np.arange(n)
    est = estimate_bytes(S, L, params, opts, kmer_width, raw_dtype)[order]
    budget = max(float(mem_budget) - _FIXED, 1.0)
    # greedy cut of the (sorted) list: prefix sums + searchsorted, O(n_batches log n)
    cs = np.concatenate([[0.0], np.cumsum(est)])
    cuts = [0]
    while cuts[-1] < n:
        a = cuts[-1]
        b = int(np.searchsorted(cs, cs[a] + budget, side='right')) - 1
        b = min(max(b, a + 1), a + int(max_reads), n)
        cuts.append(b)
    batches = [order[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    # check every batch against the engine's own figure; halve the rare one that is over
    out = []
    stack = list(reversed(batches))
    while stack:
        idx = stack.pop()
        if idx.shape[0] > 1 and exact_bytes(S[idx], L[idx], params, opts, kmer_width, raw_dtype) > mem_budget:
            h = idx.shape[0] // 2
            stack.append(idx[h:])
            stack.append(idx[:h])
        else:
            out.append(idx)
    return out
