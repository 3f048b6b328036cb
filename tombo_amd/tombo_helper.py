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
