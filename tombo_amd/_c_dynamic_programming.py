"""Drop-in for the Python-callable functions of tombo/_c_dynamic_programming.pyx that the hot
path uses, executed by the HIP kernels through the per-kernel C ABI (tba_c_*, include/
tombo_amd.h).  Same names, argument meaning, in-place behaviour and error type
(NotImplementedError with the reference's message) as the Cython module:

  c_base_z_scores                 _c_dynamic_programming.pyx:17-32
  c_banded_forward_pass           :240-279
  c_banded_traceback              :281-310
  c_adaptive_banded_forward_pass  :314-412
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
    if return_z_scores:
        raise NotImplementedError('return_z_scores is a debug-plot feature of the reference')
    for a, dt in ((fwd_pass, np.float64), (fwd_pass_tb, np.int64), (event_starts, np.int64)):
        if a.dtype != dt or not a.flags['C_CONTIGUOUS']:
            raise ValueError('in-place arguments must be C-contiguous %s arrays' % dt.__name__)
    ev, mu, sd = _f8(event_means, 'event_means'), _f8(r_ref_means, 'r_ref_means'), \
        _f8(r_ref_sds, 'r_ref_sds')
    n_bases, bw = fwd_pass.shape[0] - 1, fwd_pass.shape[1]
    eng = _engine()
    _raise(eng._L.tba_c_adaptive_banded_forward_pass(
        eng._h, fwd_pass.ctypes.data_as(_pd), fwd_pass_tb.ctypes.data_as(_pi), C.c_int64(n_bases),
        C.c_int64(bw), event_starts.ctypes.data_as(_pi), ev.ctypes.data_as(_pd),
        C.c_int64(ev.shape[0]), mu.ctypes.data_as(_pd), sd.ctypes.data_as(_pd),
        C.c_double(z_shift), C.c_double(skip_pen), C.c_double(stay_pen),
        C.c_int64(int(start_seq_pos)), C.c_double(mask_fill_z_score),
        C.c_int(bool(do_winsorize_z)), C.c_double(max_half_z_score)), eng)
    return None
