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
  c_calc_llh_ratio, c_calc_llh_ratio_const_var, c_calc_scaled_llh_ratio_const_var  :277-358
                             (row N4: plus llh_ratio_windows, many windows per call)
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


def last_ed_form():
    """_native.ED_FORM_* of the kernels that produced the last c_valid_cpts_w_cap[_t_test] result
    (these entries follow the engine's dispatch like a batch of one read: Engine.set_dispatch)"""
    eng = _engine()
    return int(eng._L.tba_c_last_ed_form(eng._h))


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


def llh_ratio_windows(kind, means, ref_means, alt_means, ref_vars, starts, width, alt_vars=None,
                      scale_factor=None, density_height_factor=None, density_height_power=None):
    """Log-likelihood ratios of many k-mer-width windows in one call (kind 0: per-base variances,
    1: constant variance ref_vars[start], 2: the scaled form) -- the loop body of
    tombo_stats.py:4042-4074 over all tested positions of a read."""
    m, r, a, rv = (_f8(x, n) for x, n in ((means, 'reg_means'), (ref_means, 'reg_ref_means'),
                                          (alt_means, 'reg_alt_means'), (ref_vars, 'reg_ref_vars')))
    av = None if alt_vars is None else _f8(alt_vars, 'reg_alt_vars')
    # the reference's Cython indexes all arrays with the first one's length (bounds-checked)
    for name, arr in (('reg_ref_means', r), ('reg_alt_means', a), ('reg_ref_vars', rv),
                      ('reg_alt_vars', av)):
        if arr is not None and arr.shape[0] < m.shape[0]:
            raise IndexError('%s is shorter than reg_means' % name)
    st = _i8(np.asarray(starts, dtype=np.int64), 'starts')
    if st.shape[0] == 0:
        return np.empty(0, dtype=np.float64)
    out = np.empty(st.shape[0], dtype=np.float64)
    par = (C.c_double * 3)(scale_factor or 0.0, density_height_factor or 0.0,
                           density_height_power or 0.0)
    eng = _engine()
    _raise(eng._L.tba_llh_ratio_windows(
        eng._h, C.c_int(kind), m.ctypes.data_as(_pd), r.ctypes.data_as(_pd), a.ctypes.data_as(_pd),
        rv.ctypes.data_as(_pd), None if av is None else av.ctypes.data_as(_pd),
        C.c_int64(m.shape[0]), C.c_int64(int(width)), st.ctypes.data_as(_pi),
        C.c_int64(st.shape[0]), par, out.ctypes.data_as(_pd)), eng)
    return out


def c_calc_llh_ratio(reg_means, reg_ref_means, reg_alt_means, reg_ref_vars, reg_alt_vars):
    return float(llh_ratio_windows(0, reg_means, reg_ref_means, reg_alt_means, reg_ref_vars, [0],
                                   len(reg_means), alt_vars=reg_alt_vars)[0])


def c_calc_llh_ratio_const_var(reg_means, reg_ref_means, reg_alt_means, const_var):
    n = len(reg_means)
    if n == 0:
        return 0.0   # the reference's empty loop
    return float(llh_ratio_windows(1, reg_means, reg_ref_means, reg_alt_means,
                                   np.full(max(n, 1), float(const_var)), [0], n)[0])


def c_calc_scaled_llh_ratio_const_var(reg_means, reg_ref_means, reg_alt_means, const_var,
                                      scale_factor, density_height_factor, density_height_power):
    n = len(reg_means)
    if n == 0:
        return 0.0
    return float(llh_ratio_windows(2, reg_means, reg_ref_means, reg_alt_means,
                                   np.full(max(n, 1), float(const_var)), [0], n,
                                   scale_factor=scale_factor,
                                   density_height_factor=density_height_factor,
                                   density_height_power=density_height_power)[0])
