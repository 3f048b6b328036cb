"""Row N3 of SURVEY.md 8(f): the host glue around the read-to-reference mapping step.

The alignment itself is minimap2 through its python binding `mappy` -- a third-party aligner that
is not part of the reference repository and not on the resquiggle hot path; nothing here
reimplements it.  What the reference does *around* the aligner call is restated here so that a
worker can go FAST5 -> mapped read -> `resquiggle_batch` -> FAST5 record entirely inside this
package, with any object that offers the two `mappy.Aligner` methods the reference uses:

    aligner.map(seq, buf=...)   -> iterable of hits with .ctg .r_st .r_en .strand .mlen .cigar
                                   .q_st .q_en          (mappy.Alignment)
    aligner.seq(ctg, start, end) -> reference substring (or None / '' outside the record)

Reference:  get_read_seq          tombo/resquiggle.py:1221-1276
            map_read              tombo/resquiggle.py:1278-1383
            _io_and_map_read      tombo/resquiggle.py:1385-1413 (signal + channel extraction),
                                  :1438-1465 (filters and the index record)
"""
import traceback

import numpy as np

from . import tombo_helper as th
from ._default_parameters import DNA_SAMP_TYPE, RNA_SAMP_TYPE

__all__ = ['get_read_seq', 'map_read', 'read_fast5_for_mapping', 'filter_and_index_record',
           'process_fast5_batch', 'USE_START_CLIP_BASES']

USE_START_CLIP_BASES = False     # resquiggle.py:76 (development switch, off in the release)

# cigar operations as mappy reports them (minimap2's BAM numbering)
_CIG_MATCH, _CIG_INS, _CIG_DEL, _CIG_SKIP, _CIG_PAD, _CIG_EQ, _CIG_X = 0, 1, 2, 3, 6, 7, 8


def get_read_seq(fast5_data, bc_grp='Basecall_1D_000', bc_subgrp='BaseCalled_template',
                 seq_samp_type=th.seqSampleType(DNA_SAMP_TYPE, False), q_score_thresh=0):
    """Basecalled sequence, read id and mean q-score from the Fastq slot of an open FAST5
    (`th.sequenceData`); RNA basecalls come back in the DNA alphabet.  Error strings are the
    reference's (they are the failure taxonomy of the run summary)."""
    try:
        fastq = fast5_data['/Analyses/' + bc_grp + '/' + bc_subgrp + '/Fastq'][()]
    except KeyError:
        raise th.TomboError('Fastq slot not present in --basecall-group')
    if isinstance(fastq, np.ndarray):      # a 0-d bytes / str dataset
        fastq = fastq.item()
    if isinstance(fastq, bytes):
        fastq = fastq.decode()
    lines = fastq.split('\n')
    read_seq, read_q = lines[1], lines[3]
    mean_q_score = th.get_mean_q_score(read_q)
    if q_score_thresh is not None and mean_q_score < q_score_thresh:
        raise th.TomboError('Read filtered by q-score.')
    raw_attrs = th.get_raw_read_slot(fast5_data).attrs
    # newer files dropped read_id; the reference falls back to read_num, then to a random number
    read_id = raw_attrs.get('read_id')
    if read_id is None and 'read_id' not in raw_attrs:
        read_id = str(raw_attrs['read_num']) if 'read_num' in raw_attrs else \
            str(np.random.randint(1000000000))
    if seq_samp_type.name == RNA_SAMP_TYPE:
        read_seq = th.rev_transcribe(read_seq)
    return th.sequenceData(seq=read_seq, id=read_id, mean_q_score=mean_q_score)


def _cigar_counts(cigar):
    """(insertions, deletions, aligned columns) of a mappy cigar [(length, op), ...]"""
    n_ins = n_del = n_aln = 0
    for op_len, op in cigar:
        if op == _CIG_INS:
            n_ins += op_len
        elif op in (_CIG_DEL, _CIG_SKIP):
            n_del += op_len
        elif op in (_CIG_MATCH, _CIG_EQ, _CIG_X):
            n_aln += op_len
        elif op != _CIG_PAD:
            # soft / hard clips never appear in a mappy cigar
            raise th.TomboError('Invalid cigar operation')
    return n_ins, n_del, n_aln


def _levels_upstream_of_start(seq_samp_type, strand):
    """Does the k-mer model need `central_pos` extra reference bases *before* the mapped start
    (and kmer_width - central_pos - 1 after the end), or the other way round?  (RNA is
    sequenced 3'->5', the minus strand reads the reference backwards.)"""
    if seq_samp_type.name == RNA_SAMP_TYPE:
        return strand == '+'
    return (strand == '-') if USE_START_CLIP_BASES else (strand == '+')


def map_read(fast5_data, aligner, std_ref, seq_samp_type=th.seqSampleType(DNA_SAMP_TYPE, False),
             bc_grp='Basecall_1D_000', bc_subgrp='BaseCalled_template', map_thr_buf=None,
             q_score_thresh=0, seq_len_rng=None):
    """Map the basecalls of an open FAST5 and return the `th.resquiggleResults` the resquiggle
    step starts from: alignment summary, genome location, and the reference sequence of the
    mapped stretch extended by the model's k-mer context (signal fields stay None)."""
    seq_data = get_read_seq(fast5_data, bc_grp, bc_subgrp, seq_samp_type, q_score_thresh)
    hit = None
    for h in aligner.map(str(seq_data.seq), buf=map_thr_buf):   # drain the iterator (mappy leaks
        if hit is None:                                          # otherwise), keep the first hit
            hit = h
    if hit is None:
        raise th.TomboError('Alignment not produced')
    ref_start, ref_end = hit.r_st, hit.r_en
    if seq_len_rng is not None and not (seq_len_rng[0] < ref_end - ref_start < seq_len_rng[1]):
        raise th.TomboError('Mapped location not within --sequence-length-range')
    strand = '+' if hit.strand == 1 else '-'
    n_ins, n_del, n_aln = _cigar_counts(hit.cigar)
    read_len = len(seq_data.seq)
    clip_5p, clip_3p = hit.q_st, read_len - hit.q_en      # in read orientation
    if strand == '-':
        clip_5p, clip_3p = clip_3p, clip_5p
    align_info = th.alignInfo(seq_data.id.decode() if isinstance(seq_data.id, bytes) else seq_data.id,
                              bc_subgrp, clip_5p, clip_3p, n_ins, n_del, hit.mlen,
                              n_aln - hit.mlen)
    # reference window = mapped stretch + the k-mer context of its first and last base; a mapping
    # that starts closer to the record start than that context is shortened instead
    before = std_ref.central_pos
    after = std_ref.kmer_width - std_ref.central_pos - 1
    if not _levels_upstream_of_start(seq_samp_type, strand):
        before, after = after, before
    ref_start = max(ref_start, before)
    genome_seq = aligner.seq(hit.ctg, ref_start - before, ref_end + after)
    if genome_seq is None or genome_seq == '':
        raise th.TomboError('Invalid mapping location')
    if isinstance(genome_seq, bytes):
        genome_seq = genome_seq.decode()
    if strand == '-':
        genome_seq = th.rev_comp(genome_seq)
    # (a mapping that runs to the end of a record yields a shorter window; the reference accepts
    # that here and lets the resquiggle step deal with the lengths)
    start_clip_bases = seq_data.seq[hit.q_en:][::-1] if USE_START_CLIP_BASES else None
    return th.resquiggleResults(
        align_info=align_info, genome_loc=th.genomeLocation(ref_start, strand, hit.ctg),
        genome_seq=genome_seq, mean_q_score=seq_data.mean_q_score,
        start_clip_bases=start_clip_bases)


def read_fast5_for_mapping(fast5_data, aligner, std_ref, seq_samp_type, bc_grp='Basecall_1D_000',
                           bc_subgrp='BaseCalled_template', map_thr_buf=None, q_score_thresh=0,
                           sig_len_rng=None, seq_len_rng=None):
    """The first half of the reference's `_io_and_map_read`: raw DAC signal and channel
    information out of the file, the read mapped, the non-canonical-base check, and the signal
    attached -- the `map_res` a worker passes through `rq.adjust_map_res` (RNA signal flip,
    stall detection) and on to `resquiggle_batch` / `resquiggle_batch_iters`."""
    try:
        channel_info = th.get_channel_info(fast5_data)
    except th.TomboError:
        channel_info = None            # not needed downstream
    try:
        all_raw_signal = th.get_raw_read_slot(fast5_data)['Signal'][:]
    except OSError:
        raise th.TomboError('Cannot read raw signal data (inflate() probably failed)')
    if sig_len_rng is not None and not (sig_len_rng[0] < all_raw_signal.shape[0] < sig_len_rng[1]):
        raise th.TomboError('Raw signal not within --signal-length-range')
    map_res = map_read(fast5_data, aligner, std_ref, seq_samp_type, bc_grp, bc_subgrp,
                       map_thr_buf, q_score_thresh, seq_len_rng)
    if th.invalid_seq(map_res.genome_seq):
        raise th.TomboError('Reference mapping contains non-canonical bases ' +
                            '(transcriptome reference cannot contain U bases)')
    return map_res._replace(raw_signal=all_raw_signal, channel_info=channel_info)


def filter_and_index_record(rsqgl_res, fast5_fn, corr_grp, bc_subgrp, seq_samp_type,
                            sig_match_thresh, obs_filter=None):
    """What `_io_and_map_read` puts on the index queue for a resquiggled read
    (resquiggle.py:1438-1465): (chrom, strand, th.readData) with the reversible filters applied --
    signal-matching score above the threshold, or observations-per-base percentiles above
    `obs_filter` [(percentile, threshold), ...]."""
    is_filtered = False
    if rsqgl_res.sig_match_score > sig_match_thresh:
        is_filtered = True
    elif obs_filter is not None:
        base_lens = np.diff(rsqgl_res.segs)
        is_filtered = any(np.percentile(base_lens, pctl) > thresh for pctl, thresh in obs_filter)
    loc = rsqgl_res.genome_loc
    mapped_end = loc.Start + len(rsqgl_res.segs) - 1
    return loc.Chrom, loc.Strand, th.readData(
        loc.Start, mapped_end, is_filtered, rsqgl_res.read_start_rel_to_raw, loc.Strand, fast5_fn,
        corr_grp + '/' + bc_subgrp, seq_samp_type.rev_sig, rsqgl_res.sig_match_score,
        rsqgl_res.mean_q_score, rsqgl_res.align_info.ID)


def process_fast5_batch(fast5s, aligner, std_ref, rsqgl_params, seq_samp_type, save_params=None,
                        outlier_thresh=None, bc_grp='Basecall_1D_000',
                        bc_subgrp='BaseCalled_template', corr_grp='RawGenomeCorrected_000',
                        compute_sd=False, obs_filter=None, q_score_thresh=0, sig_match_thresh=None,
                        sig_len_rng=None, seq_len_rng=None, map_thr_buf=None, write=True,
                        engine=None, overwrite=False):
    """One worker's share of `tombo resquiggle` for a list of reads, batch-shaped: what the
    reference does per read across `_io_and_map_read` and `_resquiggle_worker`
    (resquiggle.py:1385-1602) -- read the FAST5, map, resquiggle with the scale-iteration and
    save-bandwidth retries (`resquiggle_batch_iters`: every pass of every read on the GPU), write
    the corrected group, produce the index record.

    fast5s: [(open FAST5 object, file name), ...] (h5py.File opened 'r+', or any stand-in with
    the same group interface).  Returns (index_records, failures): index_records =
    [(chrom, strand, th.readData), ...] for the reads that succeeded (what the reference puts on
    its index queue, filters applied), failures = [(message, 'subgroup:::file', is_tombo_error)]
    in the reference's failed-reads format -- including, as there, an entry for every read that a
    reversible filter flagged (it is still written and indexed).

    With `write`, every file is first prepared like `th.prep_fast5` does (resquiggle.py:1656:
    basecalls present, corrected group absent or `overwrite`, group created with its version
    attributes) -- before any mapping or GPU work is spent on it; the failures of that step carry
    the bare file name, as the reference's do.  A read that fails with a Tombo error afterwards
    gets the message as the `status` of its corrected subgroup (`th.write_error_status`).
    """
    from . import resquiggle as rq
    from ._default_parameters import SIG_MATCH_THRESH, OUTLIER_THRESH
    if outlier_thresh is None:
        outlier_thresh = OUTLIER_THRESH
    if sig_match_thresh is None:
        sig_match_thresh = SIG_MATCH_THRESH[seq_samp_type.name]
    failures, mapped, owners = [], [], []

    def tombo_failure(f5, fn, msg):
        if write:
            try:
                th.write_error_status_data(f5, corr_grp, bc_subgrp, msg)
            except Exception:
                pass
        failures.append((msg, bc_subgrp + ':::' + fn, True))
    for k, (f5, fn) in enumerate(fast5s):
        if write:
            err = th.prep_fast5_data(f5, corr_grp, overwrite, bc_grp)
            if err is not None:
                failures.append((err, fn, True))
                continue
        try:
            mr = read_fast5_for_mapping(f5, aligner, std_ref, seq_samp_type, bc_grp, bc_subgrp,
                                        map_thr_buf, q_score_thresh, sig_len_rng, seq_len_rng)
        except th.TomboError as e:
            tombo_failure(f5, fn, str(e))
            continue
        except Exception:
            # any other per-read failure (truncated Fastq, missing dataset, aligner error) is
            # that read's failure, not the batch's: _io_and_map_read, resquiggle.py:1476-1479
            failures.append((traceback.format_exc(), bc_subgrp + ':::' + fn, False))
            continue
        mapped.append(mr)
        owners.append(k)
    # adjust_map_res (RNA flip + stall detection, resquiggle.py:1506-1530) runs on the device as
    # part of every pass
    results = rq.resquiggle_batch_iters(
        mapped, std_ref, rsqgl_params, save_params=save_params, outlier_thresh=outlier_thresh,
        seq_samp_type=seq_samp_type, engine=engine, device_prep=True) if mapped else []
    index_records = []
    for k, res in zip(owners, results):
        f5, fn = fast5s[k]
        if isinstance(res, Exception):
            if isinstance(res, th.TomboError):
                tombo_failure(f5, fn, str(res))
            else:
                failures.append((str(res), bc_subgrp + ':::' + fn, False))
            continue
        if write:
            try:
                th.write_new_fast5_group(f5, corr_grp, res, 'median', compute_sd,
                                         rna=seq_samp_type.rev_sig)
            except th.TomboError as e:
                tombo_failure(f5, fn, str(e))
                continue
            except Exception:
                failures.append((traceback.format_exc(), bc_subgrp + ':::' + fn, False))
                continue
        rec = filter_and_index_record(res, fn, corr_grp, bc_subgrp, seq_samp_type, sig_match_thresh,
                                      obs_filter)
        # the failed-reads entries of the reversible filters (resquiggle.py:1442-1455); like the
        # reference, the observations-per-base message goes out for every read that passed the
        # score filter whenever an `obs_filter` is set, filtered or not
        if res.sig_match_score > sig_match_thresh:
            failures.append(('Poor raw to expected signal matching ' +
                             '(revert with `tombo filter clear_filters`)', bc_subgrp + ':::' + fn, True))
        elif obs_filter is not None:
            failures.append(('Read filtered by observation per base ' +
                             'thresholds (revert with `tombo filter clear_filters`)',
                             bc_subgrp + ':::' + fn, True))
        index_records.append(rec)
    return index_records, failures
