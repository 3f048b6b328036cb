"""Interface types of the resquiggle hot path.

Own declarations of the namedtuples that make up the data contract of
`tombo.resquiggle.resquiggle_read()` -- field names and order follow the reference so a caller
can switch packages without touching its code:

  TomboError        /root/reference/tombo/tombo_helper.py:67
  alignInfo         tombo_helper.py:109      scaleValues       tombo_helper.py:160
  resquiggleParams  tombo_helper.py:173      stallParams       tombo_helper.py:207
  startClipParams   tombo_helper.py:217      resquiggleResults tombo_helper.py:229
  dpResults         tombo_helper.py:255      genomeLocation    tombo_helper.py:268
  channelInfo       tombo_helper.py:286      seqSampleType     tombo_helper.py:330
  get_seq_kmers     tombo_helper.py:526
"""
from collections import namedtuple


class TomboError(Exception):
    """Expected (per-read) failure; the message string is the failure taxonomy."""


alignInfo = namedtuple('alignInfo', (
    'ID', 'Subgroup', 'ClipStart', 'ClipEnd', 'Insertions', 'Deletions', 'Matches',
    'Mismatches'))

scaleValues = namedtuple('scaleValues', (
    'shift', 'scale', 'lower_lim', 'upper_lim', 'outlier_thresh'))

resquiggleParams = namedtuple('resquiggleParams', (
    'match_evalue', 'skip_pen', 'bandwidth', 'max_half_z_score', 'running_stat_width',
    'min_obs_per_base', 'raw_min_obs_per_base', 'mean_obs_per_event', 'z_shift', 'stay_pen',
    'use_t_test_seg', 'band_bound_thresh', 'start_bw', 'start_save_bw', 'start_n_bases'))
resquiggleParams.__new__.__defaults__ = (None, None, None)

stallParams = namedtuple('stallParams', (
    'window_size', 'threshold', 'min_consecutive_obs', 'edge_buffer', 'lower_pctl',
    'upper_pctl', 'mini_window_size', 'n_windows'))
stallParams.__new__.__defaults__ = (None,) * 4

startClipParams = namedtuple('startClipParams', ('bandwidth', 'num_genome_bases'))

resquiggleResults = namedtuple('resquiggleResults', (
    'align_info', 'genome_loc', 'genome_seq', 'mean_q_score', 'raw_signal', 'channel_info',
    'read_start_rel_to_raw', 'segs', 'scale_values', 'sig_match_score', 'norm_params_changed',
    'start_clip_bases', 'stall_ints'))
resquiggleResults.__new__.__defaults__ = (None,) * 9

dpResults = namedtuple('dpResults', (
    'read_start_rel_to_raw', 'segs', 'ref_means', 'ref_sds', 'genome_seq'))

genomeLocation = namedtuple('genomeLocation', ('Start', 'Strand', 'Chrom'))

channelInfo = namedtuple('channelInfo', (
    'offset', 'range', 'digitisation', 'number', 'sampling_rate'))

seqSampleType = namedtuple('seqSampleType', ('name', 'rev_sig'))


def get_seq_kmers(seq, kmer_width, rev_strand=False):
    """All overlapping k-mers of `seq` (reversed order for the reverse strand)."""
    kmers = [seq[i:i + kmer_width] for i in range(len(seq) - kmer_width + 1)]
    return kmers[::-1] if rev_strand else kmers


# ---- the Events table of a resquiggled read (compute part) -----------------
EVENTS_DTYPE = [(str('norm_mean'), 'f8'), (str('norm_stdev'), 'f8'), (str('start'), 'u4'),
                (str('length'), 'u4'), (str('base'), 'S1')]


def events_table(rsqgl_res, norm_means, norm_stds=None):
    """The structured array `write_new_fast5_group` stores as `Events`
    (tombo_helper.py:2341-2362): per base its normalised mean, standard deviation (NaN when
    `compute_sd` is off), start and length in raw samples and the base letter."""
    import numpy as np
    segs = np.asarray(rsqgl_res.segs, dtype=np.int64)
    n = segs.shape[0] - 1
    tab = np.empty(n, dtype=EVENTS_DTYPE)
    tab['norm_mean'] = norm_means
    tab['norm_stdev'] = np.nan if norm_stds is None else norm_stds
    tab['start'] = segs[:-1]
    tab['length'] = np.diff(segs)
    tab['base'] = np.frombuffer(rsqgl_res.genome_seq.encode(), dtype='S1')
    return tab


def get_event_data(rsqgl_res, compute_sd=True):
    """Events table of one finished read; the statistics come from the HIP kernels behind
    c_new_mean_stds / c_new_means."""
    from ._c_helper import c_new_mean_stds, c_new_means
    import numpy as np
    sig = np.ascontiguousarray(rsqgl_res.raw_signal, dtype=np.float64)
    segs = np.ascontiguousarray(rsqgl_res.segs, dtype=np.int64)
    if compute_sd:
        m, s = c_new_mean_stds(sig, segs)
        return events_table(rsqgl_res, m, s)
    return events_table(rsqgl_res, c_new_means(sig, segs))
