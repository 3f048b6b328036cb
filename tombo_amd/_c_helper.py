"""Drop-in for the hot-path functions of tombo/_c_helper.pyx, executed by the HIP kernels through
the per-kernel C ABI (tba_c_*):

  c_new_means                _c_helper.pyx:59-71
  c_apply_outlier_thresh     :73-87
  c_valid_cpts_w_cap         :89-120   (unsorted in the reference; tombo_helper.valid_cpts_w_cap
                                        sorts -- this one returns them sorted already, which
                                        the wrapper's .sort() leaves unchanged)
  c_valid_cpts_w_cap_t_test  :144-202
  c_new_mean_stds            :38-57
  c_compute_slopes           :362-377
"""
import ctypes as C

import numpy as np

from ._c_dynamic_programming import _engine, _f8, _i8, _raise, _pd, _pi


def c_new_means(norm_signal, new_segs):
    sig, segs = _f8(norm_signal, 'norm_signal'), _i8(new_segs, 'new_segs')
    n = segs.shape[0] - 1
    out = np.empty(n, dtype=np.float64)
    eng = _engine()
    _raise(eng._L.tba_c_new_means(eng._h, sig.ctypes.data_as(_pd), C.c_int64(sig.shape[0]),
                                  segs.ctypes.data_as(_pi), C.c_int64(n),
                                  out.ctypes.data_as(_pd)), eng)
    return out


def c_apply_outlier_thresh(raw_signal, lower_lim, upper_lim):
    sig = _f8(raw_signal, 'raw_signal')
    out = np.empty(sig.shape[0], dtype=np.float64)
    eng = _engine()
    _raise(eng._L.tba_c_apply_outlier_thresh(
        eng._h, sig.ctypes.data_as(_pd), C.c_int64(sig.shape[0]), C.c_double(lower_lim),
        C.c_double(upper_lim), out.ctypes.data_as(_pd)), eng)
    return out


def _cpts(fn, raw_signal, min_base_obs, running_stat_width, num_cpts):
    sig = _f8(raw_signal, 'raw_signal')
    out = np.empty(int(num_cpts), dtype=np.int64)
    eng = _engine()
    _raise(getattr(eng._L, fn)(
        eng._h, sig.ctypes.data_as(_pd), C.c_int64(sig.shape[0]), C.c_int64(int(min_base_obs)),
        C.c_int64(int(running_stat_width)), C.c_int64(int(num_cpts)), out.ctypes.data_as(_pi)),
        eng)
    return out


def c_valid_cpts_w_cap(raw_signal, min_base_obs, running_stat_width, num_cpts):
    return _cpts('tba_c_valid_cpts_w_cap', raw_signal, min_base_obs, running_stat_width, num_cpts)


def c_valid_cpts_w_cap_t_test(raw_signal, min_base_obs, running_stat_width, num_cpts):
    return _cpts('tba_c_valid_cpts_w_cap_t_test', raw_signal, min_base_obs, running_stat_width,
                 num_cpts)


def c_new_mean_stds(norm_signal, new_segs):
    """_c_helper.pyx:38-57: (means, population sds) of the segments"""
    sig, segs = _f8(norm_signal, 'norm_signal'), _i8(new_segs, 'new_segs')
    n = segs.shape[0] - 1
    means, stds = np.empty(n, dtype=np.float64), np.empty(n, dtype=np.float64)
    eng = _engine()
    _raise(eng._L.tba_c_new_mean_stds(
        eng._h, sig.ctypes.data_as(_pd), C.c_int64(sig.shape[0]), segs.ctypes.data_as(_pi),
        C.c_int64(n), means.ctypes.data_as(_pd), stds.ctypes.data_as(_pd)), eng)
    return means, stds


def c_compute_slopes(r_event_means, r_model_means, max_slope=1000.0):
    """_c_helper.pyx:362-377: all pairwise slopes in itertools.combinations order"""
    ev, md = _f8(r_event_means, 'r_event_means'), _f8(r_model_means, 'r_model_means')
    n = ev.shape[0]
    assert md.shape[0] == n
    out = np.empty(n * (n - 1) // 2, dtype=np.float64)
    eng = _engine()
    _raise(eng._L.tba_c_compute_slopes(
        eng._h, ev.ctypes.data_as(_pd), md.ctypes.data_as(_pd), C.c_int64(n),
        C.c_double(max_slope), out.ctypes.data_as(_pd)), eng)
    return out
