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
  sequenceData      tombo_helper.py:277      comp_seq / rev_comp / invalid_seq /
  rev_transcribe / get_mean_q_score          tombo_helper.py:370-394
  get_raw_read_slot tombo_helper.py:1071     get_channel_info  tombo_helper.py:2071
"""
import re
from collections import namedtuple

import numpy as np


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

sequenceData = namedtuple('sequenceData', ('seq', 'id', 'mean_q_score'))

PHRED_BASE = 33                      # _default_parameters.py:184
_COMPLEMENT = str.maketrans('ACGT', 'TGCA')
_NOT_ACGT = re.compile('[^ACGT]')


def comp_seq(seq):
    """complement of a DNA string (other characters pass through)"""
    return seq.translate(_COMPLEMENT)


def rev_comp(seq):
    return seq.translate(_COMPLEMENT)[::-1]


def invalid_seq(seq):
    """True when the sequence holds anything but A, C, G, T"""
    return _NOT_ACGT.search(seq) is not None


def rev_transcribe(seq):
    """RNA basecalls to the DNA alphabet (U -> T)"""
    return seq.replace('U', 'T')


def get_mean_q_score(read_q):
    """mean Phred score of a FASTQ quality string (np.mean of the per-base integers)"""
    return np.mean([q - PHRED_BASE for q in read_q.encode('ASCII')])


def get_raw_read_slot(fast5_data):
    """the (single) read group under /Raw/Reads of an open single-read FAST5"""
    try:
        return next(iter(fast5_data['/Raw/Reads'].values()))
    except KeyError:
        raise TomboError(
            'Raw data is not found in /Raw/Reads/Read_[read#]. Note that ' +
            'Tombo does not support multi-fast5 format.')


def get_channel_info(fast5_data):
    try:
        attrs = fast5_data['UniqueGlobalKey/channel_id'].attrs
    except KeyError:
        raise TomboError("No channel_id group in HDF5 file. " +
                         "Probably mux scan HDF5 file.")
    try:
        return channelInfo(attrs.get('offset'), attrs.get('range'), attrs.get('digitisation'),
                           attrs.get('channel_number'),
                           attrs.get('sampling_rate').astype(np.int64))
    except KeyError:
        raise TomboError("Channel info parameters not available.")


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


# ---- what downstream Tombo commands read back: readData + index ------------
readData = namedtuple('readData', (
    'start', 'end', 'filtered', 'read_start_rel_to_raw', 'strand', 'fn', 'corr_group', 'rna',
    'sig_match_score', 'mean_q_score', 'read_id'))   # tombo_helper.py:127-158
readData.__new__.__defaults__ = (None, None, None)

# A resquiggled read as the statistics functions need it: the index record plus the two Events
# columns they load from the FAST5 file (`norm_mean`, `base`; tombo_helper.py:1593-1647), held in
# memory -- `means` / `seq` are read-centric (5'->3' of the read), like the table.
resquiggledRead = namedtuple('resquiggledRead', readData._fields + ('means', 'seq'))
resquiggledRead.__new__.__defaults__ = (None,) * 5

SINGLE_LETTER_CODE = {
    'A': 'A', 'C': 'C', 'G': 'G', 'T': 'T', 'B': '[CGT]', 'D': '[AGT]', 'H': '[ACT]',
    'K': '[GT]', 'M': '[AC]', 'N': '[ACGT]', 'R': '[AG]', 'S': '[CG]', 'V': '[ACG]',
    'W': '[AT]', 'Y': '[CT]'}   # tombo_helper.py:54-58


class TomboMotif(object):
    """Sequence motif with the (1-based) modified position (tombo_helper.py:542-626): the part
    the alternative-model statistic uses -- `motif_pat`, `motif_len`, `mod_pos`, `mod_base`."""

    def __init__(self, raw_motif, mod_pos=None):
        import re
        bad = [c for c in raw_motif if c not in SINGLE_LETTER_CODE]
        if bad:
            raise ValueError('Invalid characters in motif: ' + ', '.join(bad))
        self.raw_motif = raw_motif
        self.motif_len = len(raw_motif)
        self.motif_pat = re.compile(''.join(SINGLE_LETTER_CODE[c] for c in raw_motif))
        self.mod_pos = mod_pos
        self.mod_base = None if mod_pos is None else raw_motif[mod_pos - 1]
        if mod_pos is not None:
            assert 0 < mod_pos <= self.motif_len


def read_from_results(rsqgl_res, norm_means, fn=None, corr_group='RawGenomeCorrected_000',
                      rna=False, read_id=None, filtered=False):
    """`resquiggledRead` of a finished read: the index fields `_resquiggle_worker` records
    (resquiggle.py:1591-1600 -> tombo_helper.py:1169-1176) plus the Events columns the statistics
    read back.  `norm_means`: per-base means (tba_batch_base_stats / get_event_data)."""
    start = rsqgl_res.genome_loc.Start
    return resquiggledRead(
        start=start, end=start + len(rsqgl_res.segs) - 1, filtered=filtered,
        read_start_rel_to_raw=rsqgl_res.read_start_rel_to_raw,
        strand=rsqgl_res.genome_loc.Strand, fn=fn,
        corr_group=corr_group + '/' + rsqgl_res.align_info.Subgroup, rna=rna,
        sig_match_score=rsqgl_res.sig_match_score, mean_q_score=rsqgl_res.mean_q_score,
        read_id=read_id if read_id is not None else rsqgl_res.align_info.ID,
        means=norm_means, seq=rsqgl_res.genome_seq)


def index_entry(rd, basedir=''):
    """The tuple `TomboReads._write_index` pickles per read (tombo_helper.py:1169-1183):
    (fn relative to the base directory, start, end, read_start_rel_to_raw, corrected group,
    basecall subgroup, filtered, rna, sig_match_score, mean_q_score, read_id)."""
    fn = rd.fn or ''
    if basedir and fn.startswith(basedir):
        fn = fn[len(basedir):]
    grp = rd.corr_group.split('/')
    return (fn, rd.start, rd.end, rd.read_start_rel_to_raw, grp[0], grp[-1], rd.filtered, rd.rna,
            rd.sig_match_score, rd.mean_q_score, rd.read_id)


def write_index(reads_by_chrm_strand, index_fn, basedir=''):
    """Tombo index file: pickle (protocol 2) of {(chrm, strand): [index_entry, ...]}
    (tombo_helper.py:1160-1187)."""
    import pickle
    data = dict((cs, [index_entry(rd, basedir) for rd in rds])
                for cs, rds in reads_by_chrm_strand.items())
    with open(index_fn, 'wb') as fp:
        pickle.dump(data, fp, protocol=2)
    return data


def read_index(index_fn, basedir=''):
    """{(chrm, strand): [readData, ...]} from a Tombo index file (inverse of `write_index`;
    the reference's parser: tombo_helper.py:1226-1260)."""
    import pickle
    with open(index_fn, 'rb') as fp:
        data = pickle.load(fp)
    out = {}
    for (chrm, strand), recs in data.items():
        out[(chrm, strand)] = [
            readData(start=s, end=e, filtered=flt, read_start_rel_to_raw=rsr, strand=strand,
                     fn=basedir + fn, corr_group=cg + '/' + sub, rna=rna, sig_match_score=sms,
                     mean_q_score=mq, read_id=rid)
            for fn, s, e, rsr, cg, sub, flt, rna, sms, mq, rid in recs]
    return out


# The on-disk layout of a resquiggled read (tombo_helper.py:2341-2460; docs/resquiggle.rst:172-194)
# as DATA: (attribute name, getter, written only when not None).  Names and order are the format.
_SUBGROUP_ATTRS = (
    ('status', lambda r, x: 'success', False),
    ('rna', lambda r, x: x['rna'], False),
    ('signal_match_score', lambda r, x: r.sig_match_score, True),
    ('shift', lambda r, x: r.scale_values.shift, False),
    ('scale', lambda r, x: r.scale_values.scale, False),
    ('norm_type', lambda r, x: x['norm_type'], False),
    ('lower_lim', lambda r, x: r.scale_values.lower_lim, True),
    ('upper_lim', lambda r, x: r.scale_values.upper_lim, True),
    ('outlier_threshold', lambda r, x: r.scale_values.outlier_thresh, True),
)
_ALIGNMENT_ATTRS = (
    ('mapped_start', lambda r: r.genome_loc.Start),
    ('mapped_end', lambda r: r.genome_loc.Start + len(r.segs) - 1),
    ('mapped_strand', lambda r: r.genome_loc.Strand),
    ('mapped_chrom', lambda r: r.genome_loc.Chrom),
)
_ALIGN_INFO_ATTRS = (   # only when the read carries an alignInfo
    ('clipped_bases_start', 'ClipStart'), ('clipped_bases_end', 'ClipEnd'),
    ('num_insertions', 'Insertions'), ('num_deletions', 'Deletions'),
    ('num_matches', 'Matches'), ('num_mismatches', 'Mismatches'),
)


# the Tombo release whose FAST5 layout this writer follows (the reference stamps its own version
# into the corrected group, tombo_helper.py:2311)
TOMBO_VERSION = '1.5.1'


def prep_fast5_data(fast5_data, corr_grp, overwrite, bc_grp=None):
    """`prep_fast5` (tombo_helper.py:2259-2324) on an OPEN, writable FAST5 object: the checks the
    reference makes before it spends any compute on a read.  The basecalls must be there; an
    existing corrected group is an error unless `overwrite` (then it is deleted); the corrected
    group is created with its `tombo_version` / `basecall_group` attributes.  Returns None, or
    the reference's message for the failed-reads list (always a "Tombo error")."""
    try:
        try:
            analyses = fast5_data['/Analyses']
            if bc_grp is not None:
                analyses[bc_grp]
        except Exception:
            return 'Base calls not found in FAST5 (see `tombo preprocess`)'
        try:
            analyses[corr_grp]
            exists = True
        except Exception:
            exists = False
        if exists:
            if not overwrite:
                return 'Tombo data exists in [--corrected-group] and [--overwrite] is not set'
            del analyses[corr_grp]
        grp = analyses.create_group(corr_grp)
        grp.attrs['tombo_version'] = TOMBO_VERSION
        grp.attrs['basecall_group'] = bc_grp
    except Exception:
        return 'Error opening or writing to fast5 file'
    return None


def write_error_status_data(fast5_data, corr_grp, bc_subgrp, error_text):
    """`write_error_status` (tombo_helper.py:2326-2339) on an open FAST5 object: the message of a
    failed read as the `status` attribute of its corrected (sub)group."""
    grp = fast5_data['/Analyses'][corr_grp]
    if bc_subgrp is not None:
        grp = grp.create_group(bc_subgrp)
    grp.attrs['status'] = error_text


def write_new_fast5_group(fast5_data, corr_grp_slot, rsqgl_res, norm_type, compute_sd,
                          alignVals=None, old_segs=None, rna=False, event_data=None):
    """Write a resquiggled read into an open FAST5 file: the groups, attributes and `Events`
    dataset of the tables above (the reference's writer: tombo_helper.py:2341-2460).
    `fast5_data` is anything with the h5py group interface (`__getitem__`, `create_group`,
    `create_dataset`, `.attrs`): an `h5py.File` where h5py is installed, or an in-memory stand-in
    (the image this engine is built in has no HDF5 library; tests use a dict-backed group and
    compare the tree with the one the reference's own writer leaves on it).
    `event_data`: the Events table if already computed on the device
    (`resquiggle_batch_events`), else it is computed here through the HIP kernels."""
    import numpy as np
    try:
        if event_data is None:
            event_data = get_event_data(rsqgl_res, compute_sd)
        datasets = []
        if alignVals is not None:
            for name, col in zip(('read_alignment', 'genome_alignment'), zip(*alignVals)):
                datasets.append((name, np.array(col, dtype='S1')))
        if old_segs is not None:
            datasets.append(('read_segments', old_segs))
    except Exception:
        raise TomboError('Error computing new events')
    try:
        sub = fast5_data['/Analyses'][corr_grp_slot].create_group(rsqgl_res.align_info.Subgroup)
        ctx = dict(rna=rna, norm_type=norm_type)
        for name, get, optional in _SUBGROUP_ATTRS:
            value = get(rsqgl_res, ctx)
            if not (optional and value is None):
                sub.attrs[name] = value
        aln = sub.create_group('Alignment')
        for name, get in _ALIGNMENT_ATTRS:
            aln.attrs[name] = get(rsqgl_res)
        if rsqgl_res.align_info is not None:
            for name, field in _ALIGN_INFO_ATTRS:
                aln.attrs[name] = getattr(rsqgl_res.align_info, field)
        for name, data in datasets:
            aln.create_dataset(name, data=data, compression='gzip')
        ev = sub.create_dataset('Events', data=event_data, compression='gzip')
        ev.attrs['read_start_rel_to_raw'] = rsqgl_res.read_start_rel_to_raw
    except Exception:
        raise TomboError('Error writing resquiggle information back into fast5 file.')
    return event_data
