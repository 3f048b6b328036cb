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
    MAX_RAW_CPTS, MIN_EVENT_TO_SEQ_RATIO, SIG_MATCH_THRESH, DNA_SAMP_TYPE,
    MAX_POINTS_FOR_THEIL_SEN)

__all__ = ['resquiggle_read', 'resquiggle_batch', 'get_engine']

_ENGINES = {}


def get_engine(device=None):
    """Process-wide engine for `device` (default: $LOCAL_RANK or 0): one process per GPU."""
    if device is None:
        device = int(os.environ.get('LOCAL_RANK', '0'))
        n = _native.lib().tba_device_count()
        if n > 0:
            device %= n
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
                     samp_inds=None, engine=None, return_debug=False):
    """resquiggle_read over a list of `resquiggleResults` (mapping results).

    Returns a list with, per read, either a `resquiggleResults` or a `TomboError` instance
    (same message the reference raises).  `samp_inds[i]`: optional precomputed Theil-Sen
    subsample for read i (1000 indices); when omitted it is drawn from numpy's global RNG in
    read order for every read longer than 1000 bases.
    """
    eng = get_engine() if engine is None else engine
    n = len(map_results)
    if n == 0:
        return []
    if eng.kmer_width != std_ref.kmer_width or getattr(eng, '_model_id', None) != id(std_ref):
        eng.set_model(std_ref.level_means, std_ref.level_sds, std_ref.kmer_width,
                      std_ref.central_pos)
        eng._model_id = id(std_ref)
    K = std_ref.kmer_width
    raws, seqs = [], []
    pre_err = [None] * n
    sv_in = np.zeros((n, 4))
    sv_flags = np.zeros(n, np.int32)
    any_sv = False
    stalls = []
    for i, mr in enumerate(map_results):
        raw = mr.raw_signal if all_raw_signals is None or all_raw_signals[i] is None \
            else all_raw_signals[i]
        if raw is None:
            pre_err[i] = th.TomboError(errors.MESSAGES[21])
            raw = np.zeros(1)
        raws.append(np.ascontiguousarray(raw, dtype=np.float64))
        codes = ts.encode_seq(mr.genome_seq)
        seqs.append(codes)
        if mr.scale_values is not None:
            sv = mr.scale_values
            any_sv = True
            sv_in[i, 0], sv_in[i, 1] = sv.shift, sv.scale
            sv_flags[i] = 1
            if sv.lower_lim is not None and sv.upper_lim is not None:
                sv_in[i, 2], sv_in[i, 3] = sv.lower_lim, sv.upper_lim
                sv_flags[i] |= 2
        stalls.append(mr.stall_ints)
    any_stall = any(s is not None and len(s) for s in stalls)
    si = None
    if not skip_seq_scaling:
        nb = [len(mr.genome_seq) - K + 1 for mr in map_results]
        if any(b > MAX_POINTS_FOR_THEIL_SEN for b in nb):
            si = np.zeros((n, MAX_POINTS_FOR_THEIL_SEN), np.int64)
            for i, b in enumerate(nb):
                if b > MAX_POINTS_FOR_THEIL_SEN:
                    si[i] = _draw_samp_ind(b) if samp_inds is None or samp_inds[i] is None \
                        else samp_inds[i]
    p = _native.make_params(rsqgl_params)
    o = _native.make_opts(
        outlier_thresh=outlier_thresh, const_scale=const_scale,
        skip_seq_scaling=skip_seq_scaling,
        sig_match_thresh=None if seq_samp_type is None else SIG_MATCH_THRESH[seq_samp_type.name],
        max_raw_cpts=max_raw_cpts, min_event_to_seq_ratio=min_event_to_seq_ratio)
    eng.upload(p, o, raws, seqs, sv_in=sv_in if any_sv else None,
               sv_flags=sv_flags if any_sv else None, samp_ind=si,
               stall_ints=stalls if any_stall else None)
    eng.run()
    out = eng.download()
    results = []
    cp = std_ref.central_pos
    dn = K - cp - 1
    for i, mr in enumerate(map_results):
        if pre_err[i] is not None:
            results.append(pre_err[i])
            continue
        st = int(out['status'][i])
        if st != 0:
            if st in errors.MESSAGES:
                results.append(th.TomboError(errors.MESSAGES[st]))
            else:
                results.append(RuntimeError('Unexpected error in resquiggle engine (status %d)' % st))
            continue
        segs = out['segs'][eng.seg_off[i]:eng.seg_off[i + 1]].copy()
        nl = int(out['norm_len'][i])
        norm = out['norm'][eng.raw_off[i]:eng.raw_off[i] + nl].copy()
        sv = out['sv'][i]
        lo = None if np.isnan(sv[2]) else float(sv[2])
        hi = None if np.isnan(sv[3]) else float(sv[3])
        # scaleValues.outlier_thresh: the reference stores the argument after sequence
        # rescaling (resquiggle.py:1187-1188) and normalize_raw_signal's value otherwise
        if skip_seq_scaling:
            ot = None if mr.scale_values is not None else outlier_thresh
            if rsqgl_params.use_t_test_seg and mr.scale_values is None and const_scale is None:
                ot = None
        else:
            ot = outlier_thresh
        results.append(mr._replace(
            read_start_rel_to_raw=int(out['read_start'][i]), segs=segs,
            genome_seq=mr.genome_seq[cp:len(mr.genome_seq) - dn], raw_signal=norm,
            scale_values=th.scaleValues(float(sv[0]), float(sv[1]), lo, hi, ot),
            sig_match_score=float(out['score'][i]),
            norm_params_changed=bool(out['changed'][i])))
    if return_debug:
        return results, out
    return results


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
    rng_state = np.random.get_state()
    res = resquiggle_batch(
        [map_res], std_ref, rsqgl_params, outlier_thresh=outlier_thresh,
        max_raw_cpts=max_raw_cpts, min_event_to_seq_ratio=min_event_to_seq_ratio,
        const_scale=const_scale, skip_seq_scaling=skip_seq_scaling,
        seq_samp_type=seq_samp_type)[0]
    if isinstance(res, Exception):
        # the reference only touches the RNG once the read reaches sequence rescaling
        if not (isinstance(res, th.TomboError) and str(res) in (
                errors.MESSAGES[19], errors.MESSAGES[20])):
            np.random.set_state(rng_state)
        raise res
    return res
