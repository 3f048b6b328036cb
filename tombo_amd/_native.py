"""ctypes binding of libtombo_amd.so (the C ABI in include/tombo_amd.h).

There is no CPU fallback: importing works anywhere (so the symbol table can be checked on a
box without a GPU), but creating an engine raises when no gfx950 device is usable.
"""
import os
import ctypes as C
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TBA_LIB_PATH: an alternative build of the same library (profiling builds with
# -DTBA_PHASE_DEBUG / -DTBA_SWEEP_STATS, A/B comparisons of a kernel variant)
LIB_PATH = os.environ.get('TBA_LIB_PATH') or os.path.join(_HERE, 'libtombo_amd.so')
CSRC = os.path.join(_HERE, 'csrc')
i64, f64, i32 = C.c_int64, C.c_double, C.c_int32

HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
               '-shared']


# a second build of the same source for ONE test (tests/test_gpu_determinism.py): the chunk-parallel traceback with a
# deterministic fault in it, which the verifier has to catch.  Never loaded by the product (LIB_PATH is).
INJECT_LIB_PATH = os.path.join(_HERE, 'libtombo_amd_inject.so')
INJECT_FLAGS = ['-DTBA_TB_INJECT=5']


def build(force=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU): libtombo_amd.so and,
    beside it, the fault-injection build one GPU test loads in a child process."""
    tree_lib = os.path.join(_HERE, 'libtombo_amd.so')
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + \
        [os.path.join(_HERE, '..', 'include', 'tombo_amd.h')]
    newest = max(os.path.getmtime(s) for s in srcs)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    jobs = []
    for path, extra in ((tree_lib, os.environ.get('TBA_EXTRA_HIPCC_FLAGS', '').split()), (INJECT_LIB_PATH, INJECT_FLAGS)):
        if force or not os.path.exists(path) or os.path.getmtime(path) < newest:
            jobs.append((path, subprocess.Popen([hipcc] + HIPCC_FLAGS + extra +
                                                ['-o', path, os.path.join(CSRC, 'tba_engine.hip')])))
    for path, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, 'hipcc -> ' + path)
    return LIB_PATH


class Params(C.Structure):
    _fields_ = [(n, f64) for n in ('match_evalue', 'skip_pen', 'max_half_z_score', 'z_shift',
                                   'stay_pen')] + \
               [(n, i64) for n in ('bandwidth', 'running_stat_width', 'min_obs_per_base',
                                   'raw_min_obs_per_base', 'mean_obs_per_event',
                                   'use_t_test_seg', 'band_bound_thresh', 'start_bw',
                                   'start_save_bw', 'start_n_bases', 'do_winsorize_z')]


class Opts(C.Structure):
    _fields_ = [('has_outlier_thresh', i64), ('outlier_thresh', f64),
                ('has_const_scale', i64), ('const_scale', f64),
                ('skip_seq_scaling', i64),
                ('check_start_score', i64), ('sig_match_thresh', f64),
                ('max_raw_cpts', i64), ('min_event_to_seq_ratio', f64),
                ('use_rna_event_scale', i64), ('rna_scale_num_events', i64),
                ('rna_scale_max_frac_events', f64), ('skip_norm_out', i64),
                ('reverse_raw', i64), ('detect_stalls', i64),
                ('stall_window_size', i64), ('stall_n_windows', i64),
                ('stall_mini_window_size', i64), ('stall_min_consecutive_obs', i64),
                ('stall_edge_buffer', i64), ('stall_threshold', f64),
                ('device_subsample', i64), ('subsample_seed', C.c_uint64), ('subsample_first_read', i64),
                ('del_fix_window', i64), ('max_del_fix_window', i64), ('extra_sig_factor', f64)]


class ReadResult(C.Structure):
    """tba_read_result: the scalar part of what resquiggle_read returns, 64 bytes per read"""
    _fields_ = [('status', i32), ('norm_params_changed', i32),
                ('read_start_rel_to_raw', i64), ('norm_len', i64),
                ('shift', f64), ('scale', f64), ('lower_lim', f64), ('upper_lim', f64),
                ('sig_match_score', f64)]


RESULT_DTYPE = np.dtype([
    ('status', np.int32), ('norm_params_changed', np.int32),
    ('read_start_rel_to_raw', np.int64), ('norm_len', np.int64),
    ('shift', np.float64), ('scale', np.float64), ('lower_lim', np.float64),
    ('upper_lim', np.float64), ('sig_match_score', np.float64)])
assert RESULT_DTYPE.itemsize == C.sizeof(ReadResult) == 64

RAW_F64, RAW_F32, RAW_I16 = 0, 1, 2
RAW_DTYPES = {np.dtype(np.float64): RAW_F64, np.dtype(np.float32): RAW_F32,
              np.dtype(np.int16): RAW_I16}


# TBA_GET_* selectors
GET_VALID_CPTS, GET_N_CPTS, GET_EVENT_MEANS, GET_SEG_NORM, GET_SEG_SV, GET_START, \
    GET_BAND_STARTS, GET_READ_TB, GET_DP_SEGS, GET_THEIL_SEN, GET_PATH, GET_LAST_ROW, \
    GET_DP_READ_START, GET_KERNEL_MS, GET_REF_MEANS, GET_REF_SDS, GET_SEGS, GET_STATUS, GET_START_FAIL, \
    GET_STALL_INTS, GET_N_STALL, GET_STALL_OFF, GET_SAMP_IND, GET_TB_PARALLEL, GET_ED_FUSED, GET_ED_TAKEN_POS, GET_ED_N_TAKEN, \
    GET_DP_WORKGROUP, GET_ED_FORM, GET_TB_FORM, GET_TB_VERIFY_FAIL = range(1, 32)
# TBA_ED_FORM_* / TBA_TB_FORM_*: which kernels produced a read's change points / main traceback
ED_FORM_NONE, ED_FORM_WG_SCAN_PEAKS, ED_FORM_DETECT_PICK, ED_FORM_SCORES_PEAKS, ED_FORM_DETECT_TT_PICK, \
    ED_FORM_TTEST_PEAKS = range(6)
TB_FORM_NONE, TB_FORM_LANE, TB_FORM_LONG, TB_FORM_PAR16, TB_FORM_PAR64 = 0, 1, 2, 16, 64
GET_DEBUG_COUNTERS = 99  # ReadState.dbg of a -DTBA_PHASE_DEBUG / -DTBA_SWEEP_STATS profiling build
STAGE_SEGMENT, STAGE_EVENT_MEANS, STAGE_REF_LEVELS, STAGE_START, STAGE_ASSIGN, STAGE_SKIP, \
    STAGE_RESCALE = range(7)
PUT_VALID_CPTS, PUT_EVENT_MEANS, PUT_NORM, PUT_REF_MEANS, PUT_REF_SDS, PUT_DP_SEGS, \
    PUT_START_STATE = range(1, 8)
MAX_BAND = 3072
ABI_VERSION = 9  # TBA_ABI_VERSION of include/tombo_amd.h
STAGE_NAMES = ["normalize", "cumsum", "scores", "peaks", "event_means", "ref_levels",
               "start_dp", "start_tb", "prep", "main_dp", "main_tb", "skip_resolve", "theil_sen",
               "rescale_score", "stalls", "total"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libtombo_amd.so is not built (run `python -c "import __graft_entry__ as g; '
                'g.build()"`); the resquiggle engine has no CPU fallback')
        L = C.CDLL(LIB_PATH)
        L.tba_last_error.restype = C.c_char_p
        # a stale build (the .so is not tracked by git) must not be driven through newer struct
        # mirrors: the engine would read past tba_opts, or miss fields, without any error
        try:
            out = (i64 * 4)()
            ok = L.tba_abi_sizes(out, i64(4)) == 0 and list(out) == [
                C.sizeof(Params), C.sizeof(Opts), C.sizeof(ReadResult), ABI_VERSION]
        except AttributeError:
            ok = False
        if not ok:
            raise RuntimeError(
                '%s does not match this binding (struct sizes / TBA_ABI_VERSION %d): rebuild it '
                '(`python -c "import __graft_entry__ as g; g.build()"`)' % (LIB_PATH, ABI_VERSION))
        _lib = L
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def make_params(rp):
    p = Params()
    p.match_evalue, p.skip_pen = rp.match_evalue, rp.skip_pen
    p.do_winsorize_z = 0 if rp.max_half_z_score is None else 1
    p.max_half_z_score = 0.0 if rp.max_half_z_score is None else rp.max_half_z_score
    p.z_shift, p.stay_pen = rp.z_shift, rp.stay_pen
    for n in ('bandwidth', 'running_stat_width', 'min_obs_per_base', 'raw_min_obs_per_base',
              'mean_obs_per_event', 'band_bound_thresh', 'start_bw', 'start_save_bw',
              'start_n_bases'):
        setattr(p, n, int(getattr(rp, n)))
    p.use_t_test_seg = int(bool(rp.use_t_test_seg))
    return p


def make_opts(outlier_thresh=None, const_scale=None, skip_seq_scaling=False,
              sig_match_thresh=None, max_raw_cpts=200, min_event_to_seq_ratio=1.1,
              skip_norm_out=False, reverse_raw=False, stall_params=None, subsample_seed=None,
              del_fix_window=2, max_del_fix_window=10, extra_sig_factor=1.1, subsample_first_read=0):
    """tba_opts.  `reverse_raw` / `stall_params` (a th.stallParams of the running-window-mean
    method): the worker's RNA preparation on the device (resquiggle.py:1506-1530);
    `subsample_seed` (int): the Theil-Sen subsample is drawn on the device;
    `del_fix_window` / `max_del_fix_window` / `extra_sig_factor`: the keyword arguments of
    resolve_skipped_bases_with_raw (resquiggle.py:405-407)."""
    o = Opts()
    o.del_fix_window, o.max_del_fix_window = int(del_fix_window), int(max_del_fix_window)
    o.extra_sig_factor = float(extra_sig_factor)
    o.reverse_raw = int(bool(reverse_raw))
    if stall_params is not None:
        sp = stall_params
        if sp.n_windows is None or sp.mini_window_size is None or \
                getattr(sp, 'lower_pctl', None) is not None:
            raise NotImplementedError('only the running-window-mean stall detector '
                                      '(MEAN_STALL_PARAMS, the reference default) is on the device')
        o.detect_stalls = 1
        o.stall_window_size, o.stall_n_windows = int(sp.window_size), int(sp.n_windows)
        o.stall_mini_window_size = int(sp.mini_window_size)
        o.stall_min_consecutive_obs, o.stall_edge_buffer = int(sp.min_consecutive_obs), int(sp.edge_buffer)
        o.stall_threshold = float(sp.threshold)
    if subsample_seed is not None:
        o.device_subsample, o.subsample_seed = 1, int(subsample_seed) & 0xffffffffffffffff
        o.subsample_first_read = int(subsample_first_read)   # (the draw of a read is keyed by its index in the JOB)
    o.has_outlier_thresh = int(outlier_thresh is not None)
    o.outlier_thresh = 0.0 if outlier_thresh is None else float(outlier_thresh)
    o.has_const_scale = int(const_scale is not None)
    o.const_scale = 0.0 if const_scale is None else float(const_scale)
    o.skip_seq_scaling = int(bool(skip_seq_scaling))
    o.check_start_score = int(sig_match_thresh is not None)
    o.sig_match_thresh = 0.0 if sig_match_thresh is None else float(sig_match_thresh)
    o.max_raw_cpts = -1 if max_raw_cpts is None else int(max_raw_cpts)
    o.min_event_to_seq_ratio = float(min_event_to_seq_ratio)
    o.use_rna_event_scale, o.rna_scale_num_events, o.rna_scale_max_frac_events = 1, 10000, 0.75
    o.skip_norm_out = int(bool(skip_norm_out))
    return o


class PinnedArray(object):
    """A numpy array over page-locked host memory (tba_pinned_alloc): uploads from it and
    downloads into it are asynchronous DMA transfers.  `.a` is the array; freed on close / GC."""

    def __init__(self, shape, dtype):
        self._L = lib()
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) if not np.isscalar(shape) else int(shape)
        self.nbytes = max(n * dtype.itemsize, 1)
        ptr = C.c_void_p()
        rc = self._L.tba_pinned_alloc(i64(self.nbytes), C.byref(ptr))
        if rc != 0:
            raise EngineError('tba_pinned_alloc failed (%d): %s' % (rc, self._L.tba_last_error().decode()))
        self._ptr = ptr
        buf = (C.c_char * self.nbytes).from_address(ptr.value)
        self.a = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)

    def close(self):
        if getattr(self, '_ptr', None):
            self.a = None
            self._L.tba_pinned_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EngineError(RuntimeError):
    pass


class PinnedPool(object):
    """Process-wide pool of page-locked blocks that RESULTS live in.  `lease(count, dtype)` returns
    a fresh ndarray over a block (or None when the pool's budget is spent); every per-read result array
    is a view of it, and the block goes back to the pool when the last view is gone (the views keep the
    lease array alive; a finalizer on it returns the block).  Blocks are reused, never freed while the
    pool is under its idle cap: hipHostMalloc is slow and hipHostFree waits for all work on the device.

    Budget (bytes of page-locked memory this PROCESS may hold in the pool, leased to live results + idle):
    $TBA_PINNED_POOL_BYTES; default a quarter of the host's memory divided by the ranks of the node
    ($LOCAL_WORLD_SIZE, else $WORLD_SIZE, else 1) and at most 16 GiB -- page-locked memory cannot be swapped,
    and eight ranks with a quarter of the host each would have pinned twice the host.  Beyond the budget
    `lease` returns None and the caller copies into pageable memory, as every result did before round 5.
    NOTE for callers: one retained result keeps its whole batch block alive (segs + signal of every read of
    the batch); copy what is kept for long (`np.array(r.segs)`), or run with TBA_PINNED_POOL_BYTES=0."""

    @staticmethod
    def _alloc_bytes(need):
        return need + need // 16 + 4096         # (what a fresh block really takes: counted against the budget)

    def __init__(self):
        import threading
        self._lock = threading.RLock()   # (a finalizer may run -- and give a block back -- inside lease)
        self._idle = []          # PinnedArray blocks (uint8), by size
        self.leased_bytes = 0
        self.idle_bytes = 0
        env = os.environ.get('TBA_PINNED_POOL_BYTES')
        if env:
            self.budget = int(env)
        else:
            try:
                ranks = max(int(os.environ.get('LOCAL_WORLD_SIZE') or os.environ.get('WORLD_SIZE') or 1), 1)
            except ValueError:
                ranks = 1
            try:
                host = os.sysconf('SC_PHYS_PAGES') * os.sysconf('SC_PAGE_SIZE')
            except (ValueError, OSError):
                host = 32 << 30
            self.budget = min(host // 4 // ranks, 16 << 30)

    def lease(self, count, dtype):
        import weakref
        dtype = np.dtype(dtype)
        need = max(int(count) * dtype.itemsize, 1)
        drop = []
        with self._lock:
            best = None
            for k, pa in enumerate(self._idle):     # smallest idle block that holds it without wasting half
                if need <= pa.nbytes <= 2 * need + (1 << 20) and (best is None or pa.nbytes < self._idle[best].nbytes):
                    best = k
            pa = self._idle.pop(best) if best is not None else None
            fresh = self._alloc_bytes(need)
            if pa is not None:
                self.idle_bytes -= pa.nbytes
                self.leased_bytes += pa.nbytes
            else:
                # make room out of idle blocks of the wrong size before giving up
                while self._idle and self.leased_bytes + self.idle_bytes + fresh > self.budget:
                    old = self._idle.pop()
                    self.idle_bytes -= old.nbytes
                    drop.append(old)
                if self.leased_bytes + self.idle_bytes + fresh > self.budget:
                    fresh = 0
                else:
                    self.leased_bytes += fresh      # reserved before the lock goes: two threads cannot both take the last room
        for old in drop:                            # hipHostFree waits for the device: never under the lock
            old.close()
        if pa is None:
            if fresh == 0:
                return None
            try:
                pa = PinnedArray(fresh, np.uint8)
            except EngineError:
                with self._lock:
                    self.leased_bytes -= fresh
                return None
        buf = (C.c_char * pa.nbytes).from_address(pa._ptr.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(count))
        weakref.finalize(arr, self._give, pa)
        return arr

    def _give(self, pa):
        with self._lock:
            self.leased_bytes -= pa.nbytes
            self._idle.append(pa)
            self.idle_bytes += pa.nbytes

    def trim(self):
        """free the idle blocks (waits for the device: hipHostFree)"""
        with self._lock:
            idle, self._idle, self.idle_bytes = self._idle, [], 0
        for pa in idle:
            pa.close()


_result_pool = None


def result_pool():
    global _result_pool
    if _result_pool is None:
        _result_pool = PinnedPool()
    return _result_pool


class Engine(object):
    """One engine per process per GPU."""

    def __init__(self, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        rc = self._L.tba_engine_create(C.c_int(device), C.byref(self._h))
        if rc != 0:
            raise EngineError('tba_engine_create failed (%d): %s' % (
                rc, self._L.tba_last_error().decode()))
        self.device = int(device)
        self.kmer_width = None
        self._keep = None
        self._model_key = None

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError('%s failed (%d): %s' % (what, rc,
                                                     self._L.tba_last_error().decode()))

    def close(self):
        if getattr(self, '_stage', None) is not None:
            self._stage.close()
            self._stage = None
        if self._h:
            self._L.tba_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_model(self, level_means, level_sds, kmer_width, central_pos):
        m = np.ascontiguousarray(level_means, dtype=np.float64)
        s = np.ascontiguousarray(level_sds, dtype=np.float64)
        assert m.shape[0] == 4 ** kmer_width == s.shape[0]
        self._check(self._L.tba_set_model(self._h, _p(m, f64), _p(s, f64), i64(kmer_width),
                                          i64(central_pos)), 'tba_set_model')
        self.kmer_width = int(kmer_width)
        self._model_key = None

    def ensure_model(self, std_ref):
        """upload std_ref's level table unless the engine already holds the same table (compared
        by content: ids are reused after garbage collection and TomboModel arrays are mutable)"""
        import hashlib
        m = np.ascontiguousarray(std_ref.level_means, dtype=np.float64)
        sd = np.ascontiguousarray(std_ref.level_sds, dtype=np.float64)
        key = (int(std_ref.kmer_width), int(std_ref.central_pos),
               hashlib.blake2b(m.tobytes() + sd.tobytes(), digest_size=16).digest())
        if key != self._model_key:
            self.set_model(m, sd, std_ref.kmer_width, std_ref.central_pos)
            self._model_key = key

    def upload(self, params, opts, raws, seqs, sv_in=None, sv_flags=None, samp_ind=None,
               stall_ints=None):
        """raws: list of sample arrays (all int16, all float32, or anything else -> float64);
        seqs: list of uint8 code arrays."""
        n = len(raws)
        raw_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([r.shape[0] for r in raws], out=raw_off[1:])
        seq_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([s.shape[0] for s in seqs], out=seq_off[1:])
        dts = set(np.asarray(r).dtype for r in raws)
        dt = next(iter(dts)) if len(dts) == 1 and next(iter(dts)) in RAW_DTYPES \
            else np.dtype(np.float64)
        raw = np.ascontiguousarray(np.concatenate(raws), dtype=dt) \
            if n > 1 else np.ascontiguousarray(raws[0], dtype=dt)
        seq = np.ascontiguousarray(np.concatenate(seqs), dtype=np.uint8)
        st, sto = pack_stalls(stall_ints)
        self.upload_packed(params, opts, raw, raw_off, seq, seq_off, sv_in=sv_in,
                           sv_flags=sv_flags, samp_ind=samp_ind, stall_ints=st, stall_off=sto,
                           wait=True)

    def upload_packed(self, params, opts, raw, raw_off, seq, seq_off, sv_in=None, sv_flags=None,
                      samp_ind=None, stall_ints=None, stall_off=None, wait=False):
        """One batch as flat CSR arrays (tba_batch_upload_async): raw (int16 / float32 / float64)
        and seq (uint8 codes) concatenated, offsets int64[n+1].  Enqueue only unless `wait`; the
        arrays are kept referenced until the next upload, but must not be modified before
        `sync()`.  Arrays in PinnedArray memory are transferred by DMA."""
        raw_off = np.ascontiguousarray(raw_off, dtype=np.int64)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        n = raw_off.shape[0] - 1
        self.n = n
        on_device = isinstance(raw, DeviceArray)   # (a Synth batch: copied device to device)
        if not on_device:
            if raw.dtype not in RAW_DTYPES or not raw.flags.c_contiguous:
                raw = np.ascontiguousarray(raw, dtype=np.float64)
            seq = np.ascontiguousarray(seq, dtype=np.uint8)
        K = self.kmer_width
        self.raw_off, self.seq_off = raw_off, seq_off
        self.B = np.maximum(np.diff(seq_off) - K + 1, 0)
        self.ref_off = np.concatenate([[0], np.cumsum(self.B)]).astype(np.int64)
        self.seg_off = self.ref_off + np.arange(n + 1)
        n_raw = np.diff(raw_off)
        ne = np.maximum(n_raw // int(params.mean_obs_per_event),
                        (self.B * float(opts.min_event_to_seq_ratio)).astype(np.int64))
        ne[(self.B <= 0) | (n_raw <= 0)] = 0
        ov = getattr(self, '_ne_override', None)
        if ov is not None and ov.shape[0] == n:
            ne = np.where((ov > 0) & (self.B > 0) & (n_raw > 0), ov, ne)
        self._ne_override = None
        self.num_events = ne
        self.ev_off = np.concatenate([[0], np.cumsum(ne)]).astype(np.int64)
        svi = None if sv_in is None else np.ascontiguousarray(sv_in, dtype=np.float64)
        svf = None if sv_flags is None else np.ascontiguousarray(sv_flags, dtype=np.int32)
        si = None if samp_ind is None else np.ascontiguousarray(samp_ind, dtype=np.int64)
        st = None if stall_ints is None else np.ascontiguousarray(stall_ints, dtype=np.int64)
        sto = None if stall_off is None else np.ascontiguousarray(stall_off, dtype=np.int64)
        self._keep = (raw, seq, raw_off, seq_off, svi, svf, si, st, sto)
        self.skip_norm_out = bool(opts.skip_norm_out)
        self._check(self._L.tba_batch_upload_async(
            self._h, C.byref(params), C.byref(opts), i64(n),
            C.c_void_p(raw.ptr) if on_device else raw.ctypes.data_as(C.c_void_p),
            C.c_int(RAW_DTYPES[raw.dtype]), _p(raw_off, i64),
            C.cast(C.c_void_p(seq.ptr), C.POINTER(C.c_uint8)) if on_device else _p(seq, C.c_uint8), _p(seq_off, i64), _p(svi, f64), _p(svf, i32), _p(si, i64),
            _p(st, i64), _p(sto, i64)), 'tba_batch_upload_async')
        self.n_raw_total = int(raw_off[-1])
        if wait:
            self.sync()

    def run(self):
        self._check(self._L.tba_batch_run(self._h), 'tba_batch_run')

    def enqueue(self):
        self._check(self._L.tba_batch_enqueue(self._h), 'tba_batch_enqueue')

    def sync(self):
        # (the targets of the finished downloads belong to the caller alone from here on: a block of
        # the result pool must not stay leased because this engine still points at it)
        self._check(self._L.tba_batch_sync(self._h), 'tba_batch_sync')
        self._keep_out = None

    def wait_for(self, other):
        """kernels enqueued next on this engine start after `other`'s last enqueued sequence"""
        self._check(self._L.tba_batch_wait_for(self._h, other._h), 'tba_batch_wait_for')

    def device_mem(self):
        """(free, total) bytes of this engine's device"""
        a, b = i64(0), i64(0)
        self._check(self._L.tba_device_mem(self._h, C.byref(a), C.byref(b)), 'tba_device_mem')
        return a.value, b.value

    def held_bytes(self):
        """device bytes of this engine's grow-only batch buffers"""
        a = i64(0)
        self._check(self._L.tba_engine_held_bytes(self._h, C.byref(a)), 'tba_engine_held_bytes')
        return a.value

    def set_sharing(self, n_engines):
        """scheduling hint: `n_engines` engines are fed concurrently on this device (tba_engine_set_sharing)"""
        self._check(self._L.tba_engine_set_sharing(self._h, int(n_engines)), 'tba_engine_set_sharing')

    def set_dispatch(self, small_batch_reads=-1, tb_wave_below=-1):
        """read-count thresholds between the latency and the throughput forms of DNA event detection
        and of the main traceback (tba_engine_set_dispatch; identical results; default 1 024 each;
        0: every batch takes the throughput form; negative: unchanged); from the next run"""
        self._check(self._L.tba_engine_set_dispatch(self._h, i64(int(small_batch_reads)), i64(int(tb_wave_below))),
                    'tba_engine_set_dispatch')

    def set_side_stream(self, mode=-1):
        """stall detection and expected levels beside normalisation / event detection on a second stream
        (tba_engine_set_side_stream): -1 while at most two engines are alive on the device, 0 never, 1 always"""
        self._check(self._L.tba_engine_set_side_stream(self._h, int(mode)), 'tba_engine_set_side_stream')

    def last_side_stream(self):
        return bool(self._L.tba_engine_last_side_stream(self._h))

    def get_dispatch(self):
        a, b = i64(0), i64(0)
        self._check(self._L.tba_engine_get_dispatch(self._h, C.byref(a), C.byref(b)), 'tba_engine_get_dispatch')
        return int(a.value), int(b.value)

    def host_stage(self):
        """this engine's reusable page-locked staging arrays (PinnedStage)"""
        st = getattr(self, '_stage', None)
        if st is None:
            st = self._stage = PinnedStage()
        return st

    def query(self):
        """True while work of this engine is still in flight (never blocks)"""
        rc = self._L.tba_batch_query(self._h)
        if rc < 0:
            self._check(rc, 'tba_batch_query')
        return rc == 1

    def download_async(self, results=None, segs32=None, segs64=None, norm=None):
        """Enqueue the copies of the finished batch's outputs into the given arrays (ideally
        PinnedArray memory): results RESULT_DTYPE[n], segs32 int32 / segs64 int64 [seg_off[-1]],
        norm float64[n_raw_total].  Valid after sync()."""
        for a, dt, cnt in ((results, RESULT_DTYPE, self.n), (segs32, np.int32, int(self.seg_off[-1])),
                           (segs64, np.int64, int(self.seg_off[-1])),
                           (norm, np.float64, self.n_raw_total)):
            if a is not None and (a.dtype != dt or a.size < cnt or not a.flags.c_contiguous):
                raise ValueError('output array has the wrong dtype / size')
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        self._keep_out = (results, segs32, segs64, norm)
        self._check(self._L.tba_batch_download_async(
            self._h, vp(results), vp(segs32), vp(segs64), vp(norm)), 'tba_batch_download_async')

    def footprint(self, params, opts, n_raw, seq_len, raw_dtype=np.float64):
        """device bytes a batch of reads with these lengths would occupy"""
        nr = np.ascontiguousarray(n_raw, dtype=np.int64)
        sl = np.ascontiguousarray(seq_len, dtype=np.int64)
        out = f64(0)
        self._check(self._L.tba_batch_footprint(
            C.byref(params), C.byref(opts), i64(self.kmer_width),
            C.c_int(RAW_DTYPES[np.dtype(raw_dtype)]), i64(nr.shape[0]), _p(nr, i64), _p(sl, i64),
            C.byref(out)), 'tba_batch_footprint')
        return out.value

    def download(self, want_norm=True):
        n = self.n
        want_norm = want_norm and not getattr(self, 'skip_norm_out', False)
        status = np.zeros(n, np.int32)
        segs = np.zeros(int(self.seg_off[-1]), np.int64)
        rs = np.zeros(n, np.int64)
        norm = np.zeros(self.n_raw_total, np.float64) if want_norm else None
        nl = np.zeros(n, np.int64)
        sv = np.zeros((n, 4))
        score = np.zeros(n)
        changed = np.zeros(n, np.int32)
        self._check(self._L.tba_batch_download(
            self._h, _p(status, i32), _p(segs, i64), _p(rs, i64), _p(norm, f64), _p(nl, i64),
            _p(sv, f64), _p(score, f64), _p(changed, i32)), 'tba_batch_download')
        return dict(status=status, segs=segs, read_start=rs, norm=norm, norm_len=nl, sv=sv,
                    score=score, changed=changed)

    def get(self, what):
        n = self.n
        shapes = {
            GET_N_CPTS: (np.int64, n), GET_DP_READ_START: (np.int64, n),
            GET_SEG_SV: (np.float64, (n, 4)), GET_START: (np.float64, (n, 4)),
            GET_THEIL_SEN: (np.float64, (n, 4)), GET_PATH: (np.int32, (n, 4)),
            GET_LAST_ROW: (np.float64, (n, MAX_BAND)), GET_KERNEL_MS: (np.float32, 32),
            GET_DEBUG_COUNTERS: (np.int64, (n, 8)),
            GET_SEG_NORM: (np.float64, self.n_raw_total),
            GET_BAND_STARTS: (np.int64, int(self.ref_off[-1])),
            GET_READ_TB: (np.int64, int(self.seg_off[-1])),
            GET_DP_SEGS: (np.int64, int(self.seg_off[-1])),
            GET_SEGS: (np.int64, int(self.seg_off[-1])),
            GET_REF_MEANS: (np.float64, int(self.ref_off[-1])),
            GET_REF_SDS: (np.float64, int(self.ref_off[-1])),
            GET_STATUS: (np.int32, n), GET_START_FAIL: (np.int32, n),
            GET_N_STALL: (np.int64, n), GET_STALL_OFF: (np.int64, n),
            GET_SAMP_IND: (np.int64, (n, 1000)),
            GET_TB_PARALLEL: (np.int32, n), GET_ED_FUSED: (np.int32, n), GET_DP_WORKGROUP: (np.int32, n),
            GET_ED_FORM: (np.int32, n), GET_TB_FORM: (np.int32, n), GET_TB_VERIFY_FAIL: (np.int32, n),
            97: (np.int64, 2 * max(int(self.seg_off[-1]), 1)),  # (a -DTBA_TB_B2 experiment build: phase B's second / third array)
            GET_ED_TAKEN_POS: (np.int32, 2 * self.n_raw_total), GET_ED_N_TAKEN: (np.int64, n),
        }
        if what in (GET_VALID_CPTS, GET_EVENT_MEANS):
            out = np.zeros(max(int(self.ev_off[-1]), 1),
                           np.int64 if what == GET_VALID_CPTS else np.float64)
            self._check(self._L.tba_batch_get(self._h, C.c_int(what), out.ctypes.data_as(C.c_void_p),
                                              i64(out.nbytes)), 'tba_batch_get')
            return out
        dt, shp = shapes[what]
        out = np.zeros(shp, dt)
        self._check(self._L.tba_batch_get(self._h, C.c_int(what), out.ctypes.data_as(C.c_void_p),
                                          i64(out.nbytes)), 'tba_batch_get')
        return out

    def stall_ints(self):
        """per read the stall intervals in force ([k, 2] int64; given, or detected on the device)"""
        cnt, off = self.get(GET_N_STALL), self.get(GET_STALL_OFF)
        if not cnt.any():
            return [np.zeros((0, 2), np.int64) for _ in range(self.n)]
        flat = np.zeros((int((off + cnt).max()), 2), np.int64)
        self._check(self._L.tba_batch_get(self._h, C.c_int(GET_STALL_INTS),
                                          flat.ctypes.data_as(C.c_void_p), i64(flat.nbytes)),
                    'tba_batch_get')
        return [flat[int(o):int(o) + int(c)].copy() for o, c in zip(off, cnt)]

    def run_stages(self, first, last):
        self._check(self._L.tba_batch_run_stages(self._h, C.c_int(first), C.c_int(last)),
                    'tba_batch_run_stages')

    def put(self, what, data, per_read=None):
        data = np.ascontiguousarray(data)
        pr = None if per_read is None else np.ascontiguousarray(per_read, dtype=np.int64)
        self._check(self._L.tba_batch_put(
            self._h, C.c_int(what), data.ctypes.data_as(C.c_void_p), i64(data.nbytes),
            _p(pr, i64)), 'tba_batch_put')

    def set_num_events(self, num_events):
        ne = None if num_events is None else np.ascontiguousarray(num_events, dtype=np.int64)
        self._check(self._L.tba_set_num_events(
            self._h, _p(ne, i64), i64(0 if ne is None else ne.shape[0])), 'tba_set_num_events')
        self._ne_override = ne

    def base_stats(self):
        """(means, stds) of every base of the finished batch, concatenated by ref_off"""
        nb = int(self.ref_off[-1])
        m, s = np.zeros(nb), np.zeros(nb)
        self._check(self._L.tba_batch_base_stats(self._h, _p(m, f64), _p(s, f64), i64(nb)),
                    'tba_batch_base_stats')
        return m, s

    def de_novo_stats(self, fm_offset, smallest_pval):
        """tba_batch_de_novo_stats: p-values of every base of the finished batch (CSR by ref_off;
        NaN outside the testable part of a read and for failed reads)"""
        nb = int(self.ref_off[-1])
        out = np.full(max(nb, 1), np.nan)
        self._check(self._L.tba_batch_de_novo_stats(
            self._h, i64(int(fm_offset)), f64(float(smallest_pval)), _p(out, f64), i64(nb)),
            'tba_batch_de_novo_stats')
        return out[:nb]

    def stats(self):
        a, c = f64(0), f64(0)
        self._check(self._L.tba_batch_stats(self._h, C.byref(a), C.byref(c)), 'tba_batch_stats')
        return a.value, c.value


class DeviceArray(object):
    """A flat array in device memory owned by someone else (a Synth's last batch): address, element
    type, element count.  Engine.upload_packed takes it in place of a numpy array."""

    def __init__(self, ptr, dtype, size, owner=None):
        self.ptr, self.dtype, self.size, self.owner = int(ptr or 0), np.dtype(dtype), int(size), owner
        self.nbytes = self.size * self.dtype.itemsize
        self.shape = (self.size,)


class SynthParams(C.Structure):
    _fields_ = [('mean_dwell', i64), ('min_dwell', i64), ('n_lead', i64), ('n_trail', i64),
                ('scale', f64), ('offset', f64), ('noise_sd', f64), ('dac_per_pa', f64), ('dac_offset', f64),
                ('reverse', i32), ('pad', i32)]


def make_synth_params(mean_dwell=9, min_dwell=2, scale=12.0, offset=90.0, noise_sd=0.25, n_lead=200,
                      n_trail=100, dac_per_pa=1.0 / 0.1709, dac_offset=10.0, reverse=False):
    """tba_synth_params: the keywords of synth.DNA_SYNTH / RNA_SYNTH + the digitisation"""
    return SynthParams(int(mean_dwell), int(min_dwell), int(n_lead), int(n_trail), float(scale), float(offset),
                       float(noise_sd), float(dac_per_pa), float(dac_offset), int(bool(reverse)), 0)


def synth_tables(sp):
    """(dwell thresholds uint32[256], noise constant) of the device generator under `sp` (host only)"""
    thr = np.zeros(256, np.uint32)
    c = f64(0.0)
    rc = lib().tba_synth_dwell_thresholds(C.byref(sp), _p(thr, C.c_uint32), i64(256), C.byref(c))
    if rc != 0:
        raise EngineError('tba_synth_dwell_thresholds failed (%d): %s' % (rc, lib().tba_last_error().decode()))
    return thr, float(c.value)


class Synth(object):
    """Synthetic reads drawn on the device (tba_synth_*, csrc/k_synth.h): a batch is a function of
    (seed, first_read, lengths, parameters) alone."""

    def __init__(self, std_ref, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        km = np.ascontiguousarray(std_ref.level_means, dtype=np.float64)
        rc = self._L.tba_synth_create(C.c_int(device), _p(km, f64), i64(int(std_ref.kmer_width)), C.byref(self._h))
        if rc != 0:
            raise EngineError('tba_synth_create failed (%d): %s' % (rc, self._L.tba_last_error().decode()))
        self.device = int(device)

    def generate(self, sp, seed, n_bases, raw_dtype=np.int16, first_read=0):
        """-> (raw DeviceArray, raw_off int64[n+1], seq DeviceArray, seq_off int64[n+1]); the device
        arrays are valid until the next call"""
        nb = np.ascontiguousarray(n_bases, dtype=np.int64)
        n = nb.shape[0]
        raw_off, seq_off = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
        d_raw, d_seq = C.c_void_p(), C.c_void_p()
        rc = self._L.tba_synth_generate(
            self._h, C.byref(sp), C.c_uint64(int(seed) & 0xffffffffffffffff), i64(int(first_read)), i64(n),
            _p(nb, i64), C.c_int(RAW_DTYPES[np.dtype(raw_dtype)]), _p(raw_off, i64), _p(seq_off, i64),
            C.byref(d_raw), C.byref(d_seq))
        if rc != 0:
            raise EngineError('tba_synth_generate failed (%d): %s' % (rc, self._L.tba_last_error().decode()))
        self._last = (np.dtype(raw_dtype), int(raw_off[-1]), int(seq_off[-1]))
        return (DeviceArray(d_raw.value, raw_dtype, raw_off[-1], self), raw_off,
                DeviceArray(d_seq.value, np.uint8, seq_off[-1], self), seq_off)

    def download(self):
        """the last batch as host arrays (raw, seq codes)"""
        dt, n_raw, n_seq = self._last
        raw, seq = np.empty(n_raw, dt), np.empty(n_seq, np.uint8)
        rc = self._L.tba_synth_download(self._h, raw.ctypes.data_as(C.c_void_p), _p(seq, C.c_uint8))
        if rc != 0:
            raise EngineError('tba_synth_download failed (%d): %s' % (rc, self._L.tba_last_error().decode()))
        return raw, seq

    def close(self):
        if self._h:
            self._L.tba_synth_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def identify_stalls(eng, raw, sp):
    """tba_identify_stalls: ts.identify_stalls (running-window-mean method) of one read on the
    device; returns an int64 [k, 2] array"""
    raw = np.asarray(raw)
    if raw.dtype not in RAW_DTYPES or not raw.flags.c_contiguous:
        raw = np.ascontiguousarray(raw, dtype=np.float64)
    n = raw.shape[0]
    cap = n // (int(sp.min_consecutive_obs) + 1) + 2
    out = np.zeros((cap, 2), np.int64)
    cnt = i64(0)
    if n == 0:
        return out[:0]
    eng._check(eng._L.tba_identify_stalls(
        eng._h, raw.ctypes.data_as(C.c_void_p), C.c_int(RAW_DTYPES[raw.dtype]), i64(n),
        i64(int(sp.window_size)), i64(int(sp.n_windows)), i64(int(sp.mini_window_size)),
        f64(float(sp.threshold)), i64(int(sp.min_consecutive_obs)), i64(int(sp.edge_buffer)),
        _p(out, i64), i64(cap), C.byref(cnt)), 'tba_identify_stalls')
    return out[:cnt.value]


def pack_stalls(stall_ints):
    """per-read stall interval lists (None / empty allowed) -> (int64 [m, 2], int64 offsets[n + 1]),
    or (None, None) for None"""
    if stall_ints is None:
        return None, None
    cnt = [0 if s is None else len(s) for s in stall_ints]
    sto = np.zeros(len(cnt) + 1, dtype=np.int64)
    np.cumsum(cnt, out=sto[1:])
    rows = [np.array([[int(a), int(b)] for a, b in s], dtype=np.int64).reshape(-1, 2)
            for s in stall_ints if s is not None and len(s)]
    st = np.ascontiguousarray(np.concatenate(rows)) if rows else np.zeros((1, 2), np.int64)
    return st, sto


class PinnedStage(object):
    """Grow-only page-locked arrays by name (the staging buffers of one batch slot): `get(name,
    count, dtype)` returns a view of `count` elements, reallocating (with 1/8 slack) only when the
    held buffer is too small or of another dtype."""

    def __init__(self):
        self._bufs = {}

    def get(self, name, count, dtype):
        dtype = np.dtype(dtype)
        count = int(count)
        pa = self._bufs.get(name)
        if pa is None or pa.a.dtype != dtype or pa.a.shape[0] < count:
            if pa is not None:
                pa.close()
            pa = PinnedArray(count + count // 8 + 16, dtype)
            self._bufs[name] = pa
        return pa.a[:count]

    def close(self):
        for pa in self._bufs.values():
            pa.close()
        self._bufs.clear()


def _addr(a):
    return a.__array_interface__['data'][0]


_str_ptr = C.pythonapi.PyUnicode_AsUTF8
_str_ptr.argtypes = [C.py_object]
_str_ptr.restype = C.c_void_p


def pack_reads(raws, seqs, reverse=False, pinned=False, n_threads=None, stage=None):
    """tba_pack_reads: per-read sample arrays (one dtype of RAW_DTYPES, else float64) and
    sequences (str / bytes of ACGT) -> (raw, raw_off, seq, seq_off, keep) CSR arrays, copied by
    native threads with the GIL released.  `stage` (a PinnedStage): the big arrays are views of its
    reusable page-locked buffers; else `pinned`: fresh page-locked memory (keep it referenced
    through `keep`); else plain numpy arrays."""
    L = lib()
    n = len(raws)
    dt = None
    for r in raws:
        if not isinstance(r, np.ndarray) or (dt is not None and r.dtype != dt) or \
                not r.flags.c_contiguous:
            dt = False
            break
        dt = r.dtype
    if dt is False or dt is None or dt not in RAW_DTYPES:
        raws = [np.asarray(r) for r in raws]
        dts = set(r.dtype for r in raws)
        dt = next(iter(dts)) if len(dts) == 1 and next(iter(dts)) in RAW_DTYPES else np.dtype(np.float64)
        raws = [r if r.dtype == dt and r.flags.c_contiguous else np.ascontiguousarray(r, dtype=dt)
                for r in raws]
    # sequences: the UTF-8 view of a str is its own buffer for ASCII text (no copy); the strings
    # themselves stay referenced by the caller's list for the duration of the call
    seq_ptr = [_str_ptr(s) if type(s) is str else None for s in seqs]
    if None in seq_ptr:
        seqs = [s if type(s) is str else bytes(s) for s in seqs]
        seq_ptr = [_str_ptr(s) if type(s) is str else C.cast(C.c_char_p(s), C.c_void_p).value
                   for s in seqs]
    raw_off = np.zeros(n + 1, np.int64)
    np.cumsum([r.shape[0] for r in raws], out=raw_off[1:])
    seq_off = np.zeros(n + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=seq_off[1:])
    if any(not s.isascii() for s in seqs if type(s) is str):
        raise ValueError('sequences must be ASCII')
    keep = []
    if stage is not None:
        raw, seq = stage.get('raw', raw_off[-1], dt), stage.get('seq', seq_off[-1], np.uint8)
    elif pinned:
        pr, ps = PinnedArray(int(raw_off[-1]), dt), PinnedArray(int(seq_off[-1]), np.uint8)
        keep = [pr, ps]
        raw, seq = pr.a, ps.a
    else:
        raw, seq = np.empty(int(raw_off[-1]), dt), np.empty(int(seq_off[-1]), np.uint8)
    rp = (C.c_void_p * n)(*[_addr(r) for r in raws])
    sp = (C.c_void_p * n)(*seq_ptr)
    if n_threads is None:
        n_threads = min(int(os.environ.get('TBA_PACK_THREADS', '16')), os.cpu_count() or 1)
    rc = L.tba_pack_reads(i64(n), rp, C.c_int(RAW_DTYPES[dt]), C.c_int(int(bool(reverse))),
                          _p(raw_off, i64), C.c_void_p(_addr(raw)), sp, _p(seq_off, i64),
                          C.cast(C.c_void_p(_addr(seq)), C.POINTER(C.c_uint8)), C.c_int(int(n_threads)))
    if rc != 0:
        raise EngineError('tba_pack_reads failed (%d): %s' % (rc, L.tba_last_error().decode()))
    return raw, raw_off, seq, seq_off, keep


def unpack_reads(src, src_off, count, dtype=None, n_threads=None):
    """tba_unpack_reads: [src[src_off[i]:src_off[i] + count[i]].copy() for i], the copies made
    by native threads.  The per-read arrays are views of ONE fresh allocation (a single large
    mapping takes huge pages where the kernel offers them: first-touch page faults, not the copy,
    bound this step with one malloc per read), so they keep each other's memory alive."""
    L = lib()
    n = len(count)
    dt = src.dtype if dtype is None else np.dtype(dtype)
    cnt = np.ascontiguousarray(count, dtype=np.int64)
    if n == 0:
        return []
    off = np.zeros(n + 1, np.int64)
    np.cumsum(cnt, out=off[1:])
    arena = np.empty(int(off[-1]), dt)
    so = np.ascontiguousarray(src_off, dtype=np.int64)
    dp = (np.uint64(_addr(arena)) + off[:-1].astype(np.uint64) * np.uint64(dt.itemsize)).astype(np.uint64)
    if n_threads is None:
        n_threads = min(32, os.cpu_count() or 1)
    rc = L.tba_unpack_reads(i64(n), C.c_void_p(_addr(src)), i64(dt.itemsize), _p(so, i64),
                            _p(cnt, i64), dp.ctypes.data_as(C.POINTER(C.c_void_p)), C.c_int(int(n_threads)))
    if rc != 0:
        raise EngineError('tba_unpack_reads failed (%d): %s' % (rc, L.tba_last_error().decode()))
    return np.split(arena, off[1:-1])
