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


def identify_stalls(all_raw_signal, stall_params=None):
    """Mean-window stall detector (caller-side RNA preparation).

    7 offsets of a 50-sample moving average; metric = (sum of the 21 pairwise absolute
    differences + the first one once more) / 21, centred at window_size//2; runs of
    metric <= threshold longer than min_consecutive_obs, widened and merged.
    """
    sp = th.stallParams(**STALL_PARAMS) if stall_params is None else stall_params
    x = np.asarray(all_raw_signal)
    n = x.shape[0]
    if n < sp.window_size:
        return []
    mw, nw = sp.mini_window_size, sp.n_windows
    assert sp.window_size == mw * nw
    csum = np.cumsum(x)
    csum[mw:] = csum[mw:] - csum[:-mw]
    mov = csum[mw - 1:] / mw
    n_pos = n - sp.window_size + 1
    offs = [mov[mw * k: mw * k + n_pos] for k in range(nw)]
    diffs = [np.abs(offs[i] - offs[j]) for i in range(nw) for j in range(i + 1, nw)]
    acc = diffs[0].copy()
    for d in diffs:
        acc += d
    metric = np.full(n, np.nan)
    start_offset = int(sp.window_size * 0.5)
    metric[start_offset:start_offset + n_pos] = acc / len(diffs)
    with np.errstate(invalid='ignore'):
        below = metric <= sp.threshold
    edges = np.where(np.diff(np.concatenate([[False], below])))[0]
    if below[-1]:
        edges = np.concatenate([edges, [n]])
    ivals = edges.reshape(-1, 2)
    ivals = ivals[(ivals[:, 1] - ivals[:, 0]) > sp.min_consecutive_obs]
    if ivals.shape[0] == 0:
        return []
    expand = (sp.window_size // 2) - sp.edge_buffer
    if expand <= 0:
        return ivals
    ivals = ivals.copy()
    ivals[:, 0] -= expand
    ivals[:, 1] += expand
    merged = [ivals[0].copy()]
    for cur in ivals:
        if cur[0] > merged[-1][1]:
            merged.append(cur.copy())
        else:
            merged[-1][1] = cur[1]
    return merged


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
