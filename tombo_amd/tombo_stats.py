"""Host-side model / parameter helpers of the resquiggle path (own implementation).

Mirrors the part of /root/reference/tombo/tombo_stats.py the hot path touches:
  TomboModel.get_exp_levels_from_seq   tombo_stats.py:834-862  (P3)
  load_resquiggle_parameters           tombo_stats.py:1505-1556
  compute_num_events                   tombo_stats.py:1558-1574
  get_dynamic_prog_params              tombo_stats.py:2364-2370
  identify_stalls (mean-window method) tombo_stats.py:269-368  (P10, caller-side, RNA only)
  remove_stall_cpts                    tombo_stats.py:1576-1597
The numeric kernels (normalisation, event detection, DP ...) live in csrc/ and are reached
through tombo_amd.resquiggle.
"""
import os
import numpy as np

from . import tombo_helper as th
from ._default_parameters import (
    ALGN_PARAMS_TABLE, SEG_PARAMS_TABLE, RNA_SAMP_TYPE, DNA_SAMP_TYPE, STANDARD_MODELS,
    HALF_NORM_EXPECTED_VAL, MIN_EVENT_TO_SEQ_RATIO, STALL_PARAMS)

_MODEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tombo_models')
_BASE_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _b in enumerate(b'ACGT'):
    _BASE_CODE[_b] = _i


def encode_seq(seq):
    """ACGT string -> uint8 codes 0..3 (255 for anything else)."""
    if isinstance(seq, str):
        seq = seq.encode()
    return _BASE_CODE[np.frombuffer(seq, dtype=np.uint8)]


class TomboModel(object):
    """Canonical k-mer level table: `means[4**K]`, `sds[4**K]` in lexicographic k-mer order.

    Constructed from a sample type (loads the extracted canonical table), or from
    `kmer_ref=[(kmer, mean, sd), ...]` + `central_pos` like the reference
    (tombo_stats.py:746-813).
    """

    def __init__(self, ref_fn=None, kmer_ref=None, central_pos=None, seq_samp_type=None):
        if kmer_ref is not None:
            assert central_pos is not None, 'central_pos must be provided with kmer_ref'
            kmers = [k for k, _, _ in kmer_ref]
            self.kmer_width = len(kmers[0])
            self.central_pos = int(central_pos)
            self.level_means = np.full(4 ** self.kmer_width, np.nan)
            self.level_sds = np.full(4 ** self.kmer_width, np.nan)
            for k, m, s in kmer_ref:
                if isinstance(k, bytes):
                    k = k.decode()
                c = self._kmer_code(k)
                self.level_means[c] = m
                self.level_sds[c] = s
        else:
            if ref_fn is None:
                if seq_samp_type is None:
                    seq_samp_type = th.seqSampleType(DNA_SAMP_TYPE, False)
                ref_fn = os.path.join(_MODEL_DIR, STANDARD_MODELS[seq_samp_type.name] + '.npz')
            tab = np.load(ref_fn)
            self.kmer_width = len(tab['kmer'][0])
            self.central_pos = int(tab['central_pos'])
            self.level_means = np.ascontiguousarray(tab['mean'], dtype=np.float64)
            self.level_sds = np.ascontiguousarray(tab['sd'], dtype=np.float64)
        self.seq_samp_type = seq_samp_type
        self._dicts = None

    @staticmethod
    def _kmer_code(kmer):
        c = 0
        for b in encode_seq(kmer):
            c = c * 4 + int(b)
        return c

    def _kmer_dicts(self):
        if self._dicts is None:
            from itertools import product
            kmers = [''.join(p) for p in product('ACGT', repeat=self.kmer_width)]
            self._dicts = (dict(zip(kmers, self.level_means.tolist())),
                           dict(zip(kmers, self.level_sds.tolist())))
        return self._dicts

    @property
    def means(self):
        return self._kmer_dicts()[0]

    @property
    def sds(self):
        return self._kmer_dicts()[1]

    def kmer_codes(self, seq):
        """int64 table index of every k-mer window of `seq`."""
        codes = encode_seq(seq).astype(np.int64)
        if (codes > 3).any():
            raise th.TomboError('Invalid sequence encountered from genome sequence.')
        n = codes.shape[0] - self.kmer_width + 1
        if n <= 0:
            return np.empty(0, dtype=np.int64)
        idx = np.zeros(n, dtype=np.int64)
        for j in range(self.kmer_width):
            idx = idx * 4 + codes[j:j + n]
        return idx

    def get_exp_levels_from_seq(self, seq, rev_strand=False):
        idx = self.kmer_codes(seq)
        if rev_strand:
            idx = idx[::-1]
        return self.level_means[idx], self.level_sds[idx]


def get_dynamic_prog_params(match_evalue):
    return HALF_NORM_EXPECTED_VAL + match_evalue, match_evalue


def load_resquiggle_parameters(seq_samp_type, sig_aln_params=None, seg_params=None,
                               use_save_bandwidth=False):
    if sig_aln_params is None:
        (match_evalue, skip_pen, bandwidth, save_bandwidth, max_half_z_score,
         band_bound_thresh, start_bw, start_save_bw,
         start_n_bases) = ALGN_PARAMS_TABLE[seq_samp_type.name]
    else:
        (match_evalue, skip_pen, bandwidth, save_bandwidth, max_half_z_score,
         band_bound_thresh, start_bw, start_save_bw, start_n_bases) = sig_aln_params
        bandwidth, save_bandwidth, band_bound_thresh = (
            int(bandwidth), int(save_bandwidth), int(band_bound_thresh))
        start_bw, start_save_bw, start_n_bases = (
            int(start_bw), int(start_save_bw), int(start_n_bases))
    if use_save_bandwidth:
        bandwidth = save_bandwidth
    if seg_params is None:
        seg_params = SEG_PARAMS_TABLE[seq_samp_type.name]
    running_stat_width, min_obs_per_base, raw_min_obs_per_base, mean_obs_per_event = seg_params
    z_shift, stay_pen = get_dynamic_prog_params(match_evalue)
    return th.resquiggleParams(
        match_evalue, skip_pen, bandwidth, max_half_z_score, running_stat_width,
        min_obs_per_base, raw_min_obs_per_base, mean_obs_per_event, z_shift, stay_pen,
        seq_samp_type.name == RNA_SAMP_TYPE, band_bound_thresh, start_bw, start_save_bw,
        start_n_bases)


def compute_num_events(signal_len, seq_len, mean_obs_per_event,
                       min_event_to_seq_ratio=MIN_EVENT_TO_SEQ_RATIO):
    return max(signal_len // mean_obs_per_event, int(seq_len * min_event_to_seq_ratio))


def identify_stalls(all_raw_signal, stall_params=None, return_metric=False):
    """Stall intervals of one read (caller-side RNA preparation, tombo_stats.py:269-368, the
    running-window-mean method): computed on the device (`tba_identify_stalls`; kernels in
    csrc/k_prep_raw.h, the same ones a batch runs under `tba_opts.detect_stalls`).  Returns a list
    of [start, end] int64 pairs like the reference.  int16 DAC, float32 and float64 samples are
    taken as they are (DAC sums are exact, float sums keep np.cumsum's order)."""
    if return_metric:
        raise NotImplementedError('the per-sample stall metric stays on the device')
    from . import _native, resquiggle as rq
    sp = th.stallParams(**STALL_PARAMS) if stall_params is None else stall_params
    if sp.lower_pctl is not None and sp.upper_pctl is not None:
        raise NotImplementedError('the percentile stall detector (PCTL_STALL_PARAMS) is not the '
                                  'reference default and not part of this engine')
    return list(_native.identify_stalls(rq.get_engine(), all_raw_signal, sp))


def remove_stall_cpts(stall_ints, valid_cpts):
    """Drop change points strictly inside a stall interval (same walk as the reference,
    including its behaviour once the interval iterator is exhausted)."""
    if len(stall_ints) == 0:
        return valid_cpts
    it = iter(stall_ints)
    cur = next(it)
    keep = []
    for i, cpt in enumerate(valid_cpts):
        while cpt > cur[1]:
            try:
                cur = next(it)
            except StopIteration:
                break
        if not (cur[0] < cpt < cur[1]):
            keep.append(i)
    return valid_cpts[keep]


# ---------------------------------------------------------------------------------------------
# Row N4: per-read test statistics over resquiggled reads -- the three
# `compute_*_read_stats` functions of tombo_stats.py:3675-4083 after their file access.  The
# reference loads `norm_mean` / `base` of one read from its FAST5 file and computes with numpy /
# scipy / Cython, one read per call; here the read is a `th.resquiggledRead` (the same columns in
# memory, e.g. straight from `resquiggle_batch_events`), the `*_batch` forms take a list of reads
# and run ONE kernel launch for all tested positions of all reads (`tba_read_pvals`,
# `tba_llh_ratio_windows`); the single-read functions are batches of one and keep the reference's
# return shape ({name: stats}, {name: positions}, read_id).  Host code only does what the
# reference does with slices: clipping to the region, strand flips, motif search.
from ._default_parameters import (   # noqa: E402
    SMALLEST_PVAL, FM_OFFSET_DEFAULT, SAMP_COMP_TXT, DE_NOVO_TXT, ALT_MODEL_TXT, CONST_SD_MODEL,
    OCLLHR_SCALE, OCLLHR_HEIGHT, OCLLHR_POWER)


class AltModel(object):
    """Alternate-base k-mer model (tombo_stats.py:922-1125), from
    `kmer_ref=[(kmer, pos, mean, sd), ...]`: expected level of a k-mer when the base at `pos` is
    the alternate base.  `motif`: `th.TomboMotif` with a modified position (default: the bare
    alternate base)."""

    def __init__(self, kmer_ref, central_pos, alt_base, name=None, motif=None):
        self.means, self.sds = {}, {}
        for kmer, pos, m, s in kmer_ref:
            if isinstance(kmer, bytes):
                kmer = kmer.decode()
            self.means[(kmer, int(pos))] = float(m)
            self.sds[(kmer, int(pos))] = float(s)
        self.central_pos, self.alt_base, self.name = int(central_pos), alt_base, name
        self.motif = th.TomboMotif(alt_base, 1) if motif is None else motif
        assert self.motif.mod_pos is not None
        self.kmer_width = len(next(iter(self.means))[0])

    def get_exp_levels_from_kmers(self, seq_kmers, rev_strand=False):
        """levels across a central base: the alternate base is the last base of the first
        k-mer and the first base of the last one (tombo_stats.py:1096-1125)"""
        K = self.kmer_width
        pos_range = range(K) if rev_strand else range(K - 1, -1, -1)
        nan = float('nan')
        return (np.array([self.means.get((k, p), nan) for k, p in zip(seq_kmers, pos_range)]),
                np.array([self.sds.get((k, p), nan) for k, p in zip(seq_kmers, pos_range)]))


def _read_pvals(means, ref_means, ref_sds, off, fm_offset, floor_out, engine=None):
    import ctypes as C
    from . import _native, resquiggle as rq
    eng = rq.get_engine() if engine is None else engine
    m, r, s = (np.ascontiguousarray(a, dtype=np.float64) for a in (means, ref_means, ref_sds))
    off = np.ascontiguousarray(off, dtype=np.int64)
    if not (m.shape[0] == r.shape[0] == s.shape[0] == int(off[-1])):
        raise ValueError('per-base arrays and offsets disagree')
    out = np.empty(m.shape[0], dtype=np.float64)
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    eng._check(eng._L.tba_read_pvals(
        eng._h, m.ctypes.data_as(pd), r.ctypes.data_as(pd), s.ctypes.data_as(pd),
        off.ctypes.data_as(pi), C.c_int64(off.shape[0] - 1), C.c_int64(int(fm_offset)),
        C.c_int(int(floor_out)), C.c_double(SMALLEST_PVAL), out.ctypes.data_as(pd)), 'tba_read_pvals')
    return out


def _fm_guard(n, fm_offset):
    if fm_offset > 0 and n < 2 * fm_offset + 1:   # calc_window_fishers_method, :2257-2259
        raise th.TomboError("P-values vector too short for Fisher's Method window compuation.")


def compute_de_novo_read_stats_batch(reads, std_ref, fm_offset=FM_OFFSET_DEFAULT, reg_data=None,
                                     engine=None):
    """compute_de_novo_read_stats (tombo_stats.py:3771-3873) for a list of `th.resquiggledRead`;
    per read (pvals, positions) or the TomboError the reference raises."""
    K, cp = std_ref.kmer_width, std_ref.central_pos
    dn = K - cp - 1
    prep, out = [], [None] * len(reads)
    for i, rd in enumerate(reads):
        try:
            if rd.means is None or rd.seq is None:
                raise th.TomboError('Read does not contain valid re-squiggled data.')
            reg_start = reg_data.start if reg_data is not None else rd.start
            reg_size = (reg_data.end - reg_data.start) if reg_data is not None else rd.end - rd.start
            lag_b, lag_e = (cp, dn) if rd.strand == '+' else (dn, cp)
            means, seq = np.asarray(rd.means, dtype=np.float64), rd.seq
            read_start, read_end = rd.start, rd.end
            # clip to the region (positions outside are not tested), :3815-3836
            if read_start + lag_b + fm_offset < reg_start:
                c = reg_start - (read_start + lag_b + fm_offset)
                read_start = reg_start - lag_b - fm_offset
                means, seq = (means[c:], seq[c:]) if rd.strand == '+' else (means[:-c], seq[:-c])
            if read_end - lag_e - fm_offset > reg_start + reg_size:
                c = (read_end - lag_e - fm_offset) - (reg_start + reg_size)
                read_end = reg_start + reg_size + lag_e + fm_offset
                means, seq = (means[:-c], seq[:-c]) if rd.strand == '+' else (means[c:], seq[c:])
            if len(seq) < K:
                raise th.TomboError('Read does not contain information in this region.')
            ref_m, ref_s = std_ref.get_exp_levels_from_seq(seq, rd.strand == '-')
            if rd.strand == '-':
                means = means[::-1]
            means = means[lag_b:means.shape[0] - lag_e]
            read_start += lag_b
            read_end -= lag_e
            _fm_guard(means.shape[0], fm_offset)
            prep.append((i, means, ref_m, ref_s, read_start, read_end))
        except th.TomboError as e:
            out[i] = e
    if prep:
        off = np.concatenate([[0], np.cumsum([p[1].shape[0] for p in prep])])
        pv = _read_pvals(np.concatenate([p[1] for p in prep]), np.concatenate([p[2] for p in prep]),
                         np.concatenate([p[3] for p in prep]), off, fm_offset, True, engine)
        for k, (i, _, _, _, rs, re_) in enumerate(prep):
            out[i] = (pv[off[k]:off[k + 1]].copy(), np.arange(rs, re_))
    return out


def compute_de_novo_read_stats(r_data, std_ref, fm_offset=FM_OFFSET_DEFAULT, reg_data=None):
    res = compute_de_novo_read_stats_batch([r_data], std_ref, fm_offset, reg_data)[0]
    if isinstance(res, Exception):
        raise res
    return {DE_NOVO_TXT: res[0]}, {DE_NOVO_TXT: res[1]}, r_data.read_id


def compute_sample_compare_read_stats_batch(reads, ctrl_means, ctrl_sds,
                                            fm_offset=FM_OFFSET_DEFAULT, reg_data=None,
                                            engine=None):
    """compute_sample_compare_read_stats (tombo_stats.py:3675-3769) for a list of reads.
    `ctrl_means` / `ctrl_sds`: control-sample levels over the region extended by fm_offset on
    both sides (NaN where the control has no coverage); with reg_data=None every read is its own
    region, so they are per-read lists."""
    ctrl_means_l = ctrl_means if reg_data is None else None
    prep, out = [], [None] * len(reads)
    for i, rd in enumerate(reads):
        try:
            if rd.means is None:
                raise th.TomboError('Read does not contain re-squiggled level means.')
            cm = np.asarray(ctrl_means_l[i] if ctrl_means_l is not None else ctrl_means, dtype=np.float64)
            cs = np.asarray(ctrl_sds[i] if ctrl_means_l is not None else ctrl_sds, dtype=np.float64)
            reg_start = reg_data.start if reg_data is not None else rd.start
            reg_size = (reg_data.end - reg_data.start) if reg_data is not None else rd.end - rd.start
            means = np.asarray(rd.means, dtype=np.float64)
            read_start, read_end = rd.start, rd.end
            if read_start + fm_offset < reg_start:
                c = reg_start - (read_start + fm_offset)
                read_start = reg_start - fm_offset
                means = means[c:] if rd.strand == '+' else means[:-c]
            if read_end - fm_offset > reg_start + reg_size:
                c = (read_end - fm_offset) - (reg_start + reg_size)
                read_end = reg_start + reg_size + fm_offset
                means = means[:-c] if rd.strand == '+' else means[c:]
            if rd.strand == '-':
                means = means[::-1]
            a, b = read_start - reg_start + fm_offset, read_end - reg_start + fm_offset
            if a < 0 or b > cm.shape[0] or b - a != means.shape[0]:
                raise ValueError('control levels do not cover the read inside the region')
            zvalid = ~(np.isnan(means) | np.isnan(cm[a:b]) | np.isnan(cs[a:b]))
            if not zvalid.any():
                raise th.TomboError('No valid z-scores in read.')
            _fm_guard(means.shape[0], fm_offset)
            prep.append((i, means, cm[a:b], cs[a:b], read_start))
        except th.TomboError as e:
            out[i] = e
    if prep:
        off = np.concatenate([[0], np.cumsum([p[1].shape[0] for p in prep])])
        pv = _read_pvals(np.concatenate([p[1] for p in prep]), np.concatenate([p[2] for p in prep]),
                         np.concatenate([p[3] for p in prep]), off, fm_offset, False, engine)
        for k, (i, _, _, _, rs) in enumerate(prep):
            p = pv[off[k]:off[k + 1]]
            poss = np.where(~np.isnan(p))[0]
            out[i] = (p[poss].copy(), poss + rs)
    return out


def compute_sample_compare_read_stats(r_data, ctrl_means, ctrl_sds, fm_offset=FM_OFFSET_DEFAULT,
                                      reg_data=None):
    res = compute_sample_compare_read_stats_batch(
        [r_data], [ctrl_means] if reg_data is None else ctrl_means,
        [ctrl_sds] if reg_data is None else ctrl_sds, fm_offset, reg_data)[0]
    if isinstance(res, Exception):
        raise res
    return {SAMP_COMP_TXT: res[0]}, {SAMP_COMP_TXT: res[1]}, r_data.read_id


def trim_seq_and_means(seq, means, r_start, reg_start, reg_end, strand, kmer_width, central_pos,
                       max_motif_bb, max_motif_ab):
    """tombo_stats.py:3889-3970: read-centric k-mers, model-able means, first alt-testable
    genomic position, motif search sequence"""
    r_end = r_start + means.shape[0]
    motif_search_seq = seq
    n_start_clip = n_end_clip = 0
    if r_start + kmer_width - 1 < reg_start:
        if strand == '+':
            n_start_clip = reg_start - (r_start + kmer_width - 1)
        else:
            n_end_clip = reg_start - (r_start + kmer_width - 1)
        r_start = reg_start - (kmer_width - 1)
    if r_end - kmer_width + 1 > reg_end:
        if strand == '+':
            n_end_clip = r_end - kmer_width + 1 - reg_end
        else:
            n_start_clip = r_end - kmer_width + 1 - reg_end
    seq = seq[n_start_clip:]
    if n_end_clip > 0:
        seq = seq[:-n_end_clip]
    means = means[n_start_clip + central_pos:]
    means = means[:-(n_end_clip + kmer_width - central_pos - 1)]
    if means.shape[0] < kmer_width:
        raise th.TomboError('Read sequence too short in this region.')
    kmers = th.get_seq_kmers(seq, kmer_width)
    if len(kmers) != means.shape[0]:
        raise th.TomboError('Mismatching k-mer and mean levels.')
    r_start += kmer_width - 1
    lead = n_start_clip + kmer_width - 1 - max_motif_bb
    motif_search_seq = motif_search_seq[lead:] if lead >= 0 else 'N' * -lead + motif_search_seq
    tail = n_end_clip + kmer_width - 1 - max_motif_ab
    # (tail == 0 slices with [:-0], i.e. to the empty string, as the reference's own line does)
    motif_search_seq = motif_search_seq[:-tail] if tail >= 0 else motif_search_seq + 'N' * -tail
    return kmers, means, r_start, motif_search_seq


def compute_alt_model_read_stats_batch(reads, std_ref, alt_refs, use_standard_llhr=False,
                                       reg_data=None, engine=None):
    """compute_alt_model_read_stats (tombo_stats.py:3972-4083) for a list of reads: per read
    ({alt_name: llhrs}, {alt_name: positions}) or the TomboError.  Every motif hit of every read
    and model becomes one window of ONE `tba_llh_ratio_windows` launch."""
    from ._c_helper import llh_ratio_windows
    K = std_ref.kmer_width
    max_bb = max(ar.motif.mod_pos - 1 for _, ar in alt_refs)
    max_ab = max(ar.motif.motif_len - ar.motif.mod_pos for _, ar in alt_refs)
    out = [None] * len(reads)
    wins = []   # (read, alt name, genomic position, means[K], ref_means[K], alt_means[K], vars)
    for i, rd in enumerate(reads):
        try:
            if rd.means is None or rd.seq is None:
                raise th.TomboError('Read does not contain valid re-squiggled data.')
            reg_start = reg_data.start if reg_data is not None else rd.start
            reg_end = reg_data.end if reg_data is not None else rd.end
            kmers, means, r_start, msseq = trim_seq_and_means(
                rd.seq, np.asarray(rd.means, dtype=np.float64), rd.start, reg_start, reg_end,
                rd.strand, K, std_ref.central_pos, max_bb, max_ab)
            testable_len = means.shape[0] - K + 1
            idx = np.array([std_ref._kmer_code(k) for k in kmers], dtype=np.int64)
            ref_m, ref_v = std_ref.level_means[idx], np.square(std_ref.level_sds[idx])
            out[i] = ({}, {})
            for name, ar in alt_refs:
                s_seq = msseq[max_bb - (ar.motif.mod_pos - 1):]
                cut = max_ab - (ar.motif.motif_len - ar.motif.mod_pos)
                if cut > 0:
                    s_seq = s_seq[:-cut]
                poss = []
                for m in ar.motif.motif_pat.finditer(s_seq):
                    ap = m.start()
                    poss.append(r_start + ap if rd.strand == '+' else r_start + testable_len - ap - 1)
                    am, asd = ar.get_exp_levels_from_kmers(kmers[ap:ap + ar.kmer_width])
                    if not CONST_SD_MODEL and not use_standard_llhr:
                        raise th.TomboError('Variable SD scaled likelihood ratio not implemented.')
                    wins.append((i, name, means[ap:ap + K], ref_m[ap:ap + K], am,
                                 ref_v[ap:ap + K], np.square(asd)))
                out[i][1][name] = np.array(poss)
                out[i][0][name] = np.empty(len(poss))
        except th.TomboError as e:
            out[i] = e
    if wins:
        kind = (1 if use_standard_llhr else 2) if CONST_SD_MODEL else 0
        cat = lambda k: np.concatenate([w[k] for w in wins])
        vals = llh_ratio_windows(
            kind, cat(2), cat(3), cat(4), cat(5), np.arange(len(wins), dtype=np.int64) * K, K,
            alt_vars=cat(6) if kind == 0 else None, scale_factor=OCLLHR_SCALE,
            density_height_factor=OCLLHR_HEIGHT, density_height_power=OCLLHR_POWER)
        fill = {}
        for (i, name, *_), v in zip(wins, vals):
            k = fill.get((i, name), 0)
            out[i][0][name][k] = v
            fill[(i, name)] = k + 1
    return out


def compute_alt_model_read_stats(r_data, std_ref, alt_refs, use_standard_llhr=False,
                                 reg_data=None):
    res = compute_alt_model_read_stats_batch([r_data], std_ref, alt_refs, use_standard_llhr,
                                             reg_data)[0]
    if isinstance(res, Exception):
        raise res
    return res[0], res[1], r_data.read_id
