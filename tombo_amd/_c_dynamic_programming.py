"""Drop-in for the Python-callable functions of tombo/_c_dynamic_programming.pyx that the hot
path uses, executed by the HIP kernels through the per-kernel C ABI (tba_c_*, include/
tombo_amd.h).  Same names, argument meaning, in-place behaviour and error type
(NotImplementedError with the reference's message) as the Cython module:

  c_base_z_scores                 _c_dynamic_programming.pyx:17-32
  c_banded_forward_pass           :240-279
  c_banded_traceback              :281-310
  c_adaptive_banded_forward_pass  :314-412
  c_reg_z_scores                  :34-97
  c_base_forward_pass             :99-163
  c_base_traceback                :165-182
"""
import ctypes as C

import numpy as np

from . import _native, errors

_pd, _pi = C.POINTER(C.c_double), C.POINTER(C.c_int64)


def _engine():
    from .resquiggle import get_engine
    return get_engine()


def _f8(a, name):
    if a is None:
        raise TypeError("Argument '%s' must not be None" % name)
    if a.dtype != np.float64:
        raise ValueError("Buffer dtype mismatch, expected 'DTYPE_t' but got '%s'" % a.dtype)
    return np.ascontiguousarray(a)


def _i8(a, name):
    if a is None:
        raise TypeError("Argument '%s' must not be None" % name)
    if a.dtype != np.int64:
        raise ValueError("Buffer dtype mismatch, expected 'DTYPE_INT_t' but got '%s'" % a.dtype)
    return np.ascontiguousarray(a)


def _raise(rc, eng):
    if rc == 0:
        return
    if rc < 0:
        raise _native.EngineError(eng._L.tba_last_error().decode())
    if rc in errors.MESSAGES:
        raise NotImplementedError(errors.MESSAGES[rc])
    raise RuntimeError('Unexpected error in resquiggle engine (status %d)' % rc)


def c_base_z_scores(b_sig, ref_mean, ref_sd, do_winsorize_z=False, max_half_z_score=10.0):
    b_sig = _f8(b_sig, 'b_sig')
    out = np.empty(b_sig.shape[0], dtype=np.float64)
    eng = _engine()
    _raise(eng._L.tba_c_base_z_scores(
        eng._h, b_sig.ctypes.data_as(_pd), C.c_int64(b_sig.shape[0]), C.c_double(ref_mean),
        C.c_double(ref_sd), C.c_int(bool(do_winsorize_z)), C.c_double(max_half_z_score),
        out.ctypes.data_as(_pd)), eng)
    return out


def c_banded_forward_pass(shifted_z_scores, event_starts, skip_pen, stay_pen):
    z = _f8(shifted_z_scores, 'shifted_z_scores')
    es = _i8(event_starts, 'event_starts')
    n_bases, bw = z.shape
    fwd = np.empty((n_bases + 1, bw), dtype=np.float64)
    tb = np.empty((n_bases + 1, bw), dtype=np.int64)
    eng = _engine()
    _raise(eng._L.tba_c_banded_forward_pass(
        eng._h, z.ctypes.data_as(_pd), C.c_int64(n_bases), C.c_int64(bw), es.ctypes.data_as(_pi),
        C.c_double(skip_pen), C.c_double(stay_pen), fwd.ctypes.data_as(_pd),
        tb.ctypes.data_as(_pi)), eng)
    return fwd, tb


def c_banded_traceback(fwd_pass_tb, event_starts, band_pos, band_boundary_thresh=-1):
    tb = _i8(fwd_pass_tb, 'fwd_pass_tb')
    es = _i8(event_starts, 'event_starts')
    n_bases, bw = tb.shape[0] - 1, tb.shape[1]
    out = np.empty(n_bases + 1, dtype=np.int64)
    eng = _engine()
    _raise(eng._L.tba_c_banded_traceback(
        eng._h, tb.ctypes.data_as(_pi), C.c_int64(n_bases), C.c_int64(bw), es.ctypes.data_as(_pi),
        C.c_int64(int(band_pos)), C.c_int64(int(band_boundary_thresh)), out.ctypes.data_as(_pi)),
        eng)
    return out


def c_adaptive_banded_forward_pass(
        fwd_pass, fwd_pass_tb, event_starts, event_means, r_ref_means, r_ref_sds, z_shift,
        skip_pen, stay_pen, start_seq_pos, mask_fill_z_score, do_winsorize_z, max_half_z_score,
        return_z_scores=False):
    """fwd_pass, fwd_pass_tb and event_starts are updated in place (rows after start_seq_pos)."""
    for a, dt in ((fwd_pass, np.float64), (fwd_pass_tb, np.int64), (event_starts, np.int64)):
        if a.dtype != dt or not a.flags['C_CONTIGUOUS']:
            raise ValueError('in-place arguments must be C-contiguous %s arrays' % dt.__name__)
    ev, mu, sd = _f8(event_means, 'event_means'), _f8(r_ref_means, 'r_ref_means'), \
        _f8(r_ref_sds, 'r_ref_sds')
    n_bases, bw = fwd_pass.shape[0] - 1, fwd_pass.shape[1]
    eng = _engine()
    # return_z_scores (pyx:339,387-388,409-410): the shifted z-scores of the rows the pass computed
    z = np.empty((n_bases - int(start_seq_pos), bw), dtype=np.float64) if return_z_scores else None
    _raise(eng._L.tba_c_adaptive_banded_forward_pass_z(
        eng._h, fwd_pass.ctypes.data_as(_pd), fwd_pass_tb.ctypes.data_as(_pi), C.c_int64(n_bases),
        C.c_int64(bw), event_starts.ctypes.data_as(_pi), ev.ctypes.data_as(_pd),
        C.c_int64(ev.shape[0]), mu.ctypes.data_as(_pd), sd.ctypes.data_as(_pd),
        C.c_double(z_shift), C.c_double(skip_pen), C.c_double(stay_pen),
        C.c_int64(int(start_seq_pos)), C.c_double(mask_fill_z_score),
        C.c_int(bool(do_winsorize_z)), C.c_double(max_half_z_score),
        None if z is None else z.ctypes.data_as(_pd)), eng)
    return z


def _raise_index(rc, eng):
    # the bounds-checked Cython buffers raise IndexError where these kernels report TBA_INTERNAL
    if rc == 100:
        raise IndexError('Out of bounds on buffer access (axis 0)')
    _raise(rc, eng)


def c_reg_z_scores(r_sig, r_ref_means, r_ref_sds, r_b_starts, reg_start, reg_end,
                   max_base_shift, min_obs_per_base, max_half_z_score=None):
    """_c_dynamic_programming.pyx:34-97: list of (z_scores, (start, end)) per base of the region"""
    r_sig, mu, sd = _f8(r_sig, 'r_sig'), _f8(r_ref_means, 'r_ref_means'), _f8(r_ref_sds, 'r_ref_sds')
    starts = _i8(r_b_starts, 'r_b_starts')
    reg_start, reg_end = int(reg_start), int(reg_end)
    n = reg_end - reg_start
    if n <= 0:
        return []
    if reg_start < 0 or reg_end >= starts.shape[0] or reg_end > mu.shape[0]:
        raise IndexError('Out of bounds on buffer access (axis 0)')
    cap = max(1, n * int(starts[reg_end] - starts[reg_start]))
    bounds, off = np.empty((n, 2), np.int64), np.empty(n + 1, np.int64)
    z = np.empty(cap, np.float64)
    eng = _engine()
    _raise_index(eng._L.tba_c_reg_z_scores(
        eng._h, r_sig.ctypes.data_as(_pd), C.c_int64(r_sig.shape[0]), mu.ctypes.data_as(_pd),
        sd.ctypes.data_as(_pd), C.c_int64(min(mu.shape[0], sd.shape[0])),
        starts.ctypes.data_as(_pi), C.c_int64(starts.shape[0]), C.c_int64(reg_start),
        C.c_int64(reg_end), C.c_int64(int(max_base_shift)), C.c_int64(int(min_obs_per_base)),
        C.c_int(max_half_z_score is not None),
        C.c_double(0.0 if max_half_z_score is None else max_half_z_score),
        bounds.ctypes.data_as(_pi), off.ctypes.data_as(_pi), z.ctypes.data_as(_pd),
        C.c_int64(cap)), eng)
    return [(z[off[i]:off[i + 1]].copy(), (int(bounds[i, 0]), int(bounds[i, 1])))
            for i in range(n)]


def c_base_forward_pass(b_data, b_start, b_end, prev_b_data, prev_b_start, prev_b_end,
                        prev_b_fwd_data, prev_b_last_diag, min_obs_per_base):
    """_c_dynamic_programming.pyx:99-163: returns (b_fwd_data, b_last_diag)"""
    b, pb = _f8(b_data, 'b_data'), _f8(prev_b_data, 'prev_b_data')
    pf, pl = _f8(prev_b_fwd_data, 'prev_b_fwd_data'), _i8(prev_b_last_diag, 'prev_b_last_diag')
    b_start, b_end, prev_b_start, prev_b_end = (int(x) for x in (
        b_start, b_end, prev_b_start, prev_b_end))
    b_len, plen = b_end - b_start, prev_b_end - prev_b_start
    if b_len <= 0 or plen <= 0 or b.shape[0] < b_len or min(
            pb.shape[0], pf.shape[0], pl.shape[0]) < plen:
        raise IndexError('Out of bounds on buffer access (axis 0)')
    fwd, ld = np.empty(b_len, np.float64), np.empty(b_len, np.int64)
    eng = _engine()
    _raise_index(eng._L.tba_c_base_forward_pass(
        eng._h, b.ctypes.data_as(_pd), C.c_int64(b_start), C.c_int64(b_end),
        pb.ctypes.data_as(_pd), C.c_int64(prev_b_start), C.c_int64(prev_b_end),
        pf.ctypes.data_as(_pd), pl.ctypes.data_as(_pi), C.c_int64(int(min_obs_per_base)),
        fwd.ctypes.data_as(_pd), ld.ctypes.data_as(_pi)), eng)
    return fwd, ld


def c_base_traceback(curr_b_data, curr_start, next_b_data, next_start, next_end, sig_start,
                     min_obs_per_base):
    """_c_dynamic_programming.pyx:165-182: the new boundary, or None when the scan runs out"""
    cur, nxt = _f8(curr_b_data, 'curr_b_data'), _f8(next_b_data, 'next_b_data')
    out = C.c_int64(-1)
    eng = _engine()
    _raise_index(eng._L.tba_c_base_traceback(
        eng._h, cur.ctypes.data_as(_pd), C.c_int64(cur.shape[0]), C.c_int64(int(curr_start)),
        nxt.ctypes.data_as(_pd), C.c_int64(nxt.shape[0]), C.c_int64(int(next_start)),
        C.c_int64(int(next_end)), C.c_int64(int(sig_start)), C.c_int64(int(min_obs_per_base)),
        C.byref(out)), eng)
    return None if out.value < 0 else int(out.value)
