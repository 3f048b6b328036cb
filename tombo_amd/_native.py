"""ctypes binding of libtombo_amd.so (the C ABI in include/tombo_amd.h).

There is no CPU fallback: importing works anywhere (so the symbol table can be checked on a
box without a GPU), but creating an engine raises when no gfx950 device is usable.
"""
import os
import ctypes as C
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libtombo_amd.so')
CSRC = os.path.join(_HERE, 'csrc')
i64, f64, i32 = C.c_int64, C.c_double, C.c_int32

HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
               '-shared']


def build(force=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + \
        [os.path.join(_HERE, '..', 'include', 'tombo_amd.h')]
    if (not force and os.path.exists(LIB_PATH) and
            os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    subprocess.check_call([hipcc] + HIPCC_FLAGS + os.environ.get('TBA_EXTRA_HIPCC_FLAGS', '').split() +
                          ['-o', LIB_PATH,
                                                   os.path.join(CSRC, 'tba_engine.hip')])
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
                ('rna_scale_max_frac_events', f64)]


# TBA_GET_* selectors
GET_VALID_CPTS, GET_N_CPTS, GET_EVENT_MEANS, GET_SEG_NORM, GET_SEG_SV, GET_START, \
    GET_BAND_STARTS, GET_READ_TB, GET_DP_SEGS, GET_THEIL_SEN, GET_PATH, GET_LAST_ROW, \
    GET_DP_READ_START, GET_KERNEL_MS, GET_REF_MEANS, GET_REF_SDS, GET_SEGS, GET_STATUS, GET_START_FAIL = range(1, 20)
STAGE_SEGMENT, STAGE_EVENT_MEANS, STAGE_REF_LEVELS, STAGE_START, STAGE_ASSIGN, STAGE_SKIP, \
    STAGE_RESCALE = range(7)
PUT_VALID_CPTS, PUT_EVENT_MEANS, PUT_NORM, PUT_REF_MEANS, PUT_REF_SDS, PUT_DP_SEGS, \
    PUT_START_STATE = range(1, 8)
MAX_BAND = 3072
STAGE_NAMES = ["normalize", "cumsum", "scores", "peaks", "event_means", "ref_levels",
               "start_dp", "start_tb", "prep", "main_dp", "main_tb", "skip_resolve", "theil_sen",
               "rescale_score", "rna_scale", "total"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libtombo_amd.so is not built (run `python -c "import __graft_entry__ as g; '
                'g.build()"`); the resquiggle engine has no CPU fallback')
        _lib = C.CDLL(LIB_PATH)
        _lib.tba_last_error.restype = C.c_char_p
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
              sig_match_thresh=None, max_raw_cpts=200, min_event_to_seq_ratio=1.1):
    o = Opts()
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
    return o


class EngineError(RuntimeError):
    pass


class Engine(object):
    """One engine per process per GPU."""

    def __init__(self, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        rc = self._L.tba_engine_create(C.c_int(device), C.byref(self._h))
        if rc != 0:
            raise EngineError('tba_engine_create failed (%d): %s' % (
                rc, self._L.tba_last_error().decode()))
        self.kmer_width = None
        self._keep = None

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError('%s failed (%d): %s' % (what, rc,
                                                     self._L.tba_last_error().decode()))

    def close(self):
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

    def upload(self, params, opts, raws, seqs, sv_in=None, sv_flags=None, samp_ind=None,
               stall_ints=None):
        """raws: list of float64 arrays; seqs: list of uint8 code arrays."""
        n = len(raws)
        self.n = n
        raw_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([r.shape[0] for r in raws], out=raw_off[1:])
        seq_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([s.shape[0] for s in seqs], out=seq_off[1:])
        raw = np.ascontiguousarray(np.concatenate(raws), dtype=np.float64) \
            if n > 1 else np.ascontiguousarray(raws[0], dtype=np.float64)
        seq = np.ascontiguousarray(np.concatenate(seqs), dtype=np.uint8)
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
        st = sto = None
        if stall_ints is not None:
            cnt = [0 if s is None else len(s) for s in stall_ints]
            sto = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(cnt, out=sto[1:])
            rows = [np.array([[int(a), int(b)] for a, b in s], dtype=np.int64).reshape(-1, 2)
                    for s in stall_ints if s is not None and len(s)]
            st = np.ascontiguousarray(np.concatenate(rows)) if rows else np.zeros((1, 2), np.int64)
        self._keep = (raw, seq, raw_off, seq_off, svi, svf, si, st, sto)
        self._check(self._L.tba_batch_upload(
            self._h, C.byref(params), C.byref(opts), i64(n), _p(raw, f64), _p(raw_off, i64),
            _p(seq, C.c_uint8), _p(seq_off, i64), _p(svi, f64), _p(svf, i32), _p(si, i64),
            _p(st, i64), _p(sto, i64)), 'tba_batch_upload')
        self.n_raw_total = int(raw_off[-1])

    def run(self):
        self._check(self._L.tba_batch_run(self._h), 'tba_batch_run')

    def enqueue(self):
        self._check(self._L.tba_batch_enqueue(self._h), 'tba_batch_enqueue')

    def sync(self):
        self._check(self._L.tba_batch_sync(self._h), 'tba_batch_sync')

    def download(self, want_norm=True):
        n = self.n
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
            99: (np.int64, (n, 8)),
            GET_SEG_NORM: (np.float64, self.n_raw_total),
            GET_BAND_STARTS: (np.int64, int(self.ref_off[-1])),
            GET_READ_TB: (np.int64, int(self.seg_off[-1])),
            GET_DP_SEGS: (np.int64, int(self.seg_off[-1])),
            GET_SEGS: (np.int64, int(self.seg_off[-1])),
            GET_REF_MEANS: (np.float64, int(self.ref_off[-1])),
            GET_REF_SDS: (np.float64, int(self.ref_off[-1])),
            GET_STATUS: (np.int32, n), GET_START_FAIL: (np.int32, n),
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

    def stats(self):
        a, c = f64(0), f64(0)
        self._check(self._L.tba_batch_stats(self._h, C.byref(a), C.byref(c)), 'tba_batch_stats')
        return a.value, c.value
