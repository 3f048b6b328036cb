"""Deterministic synthetic nanopore reads.

Read i of a set uses numpy.random.default_rng(base_seed + i): sequence = B+K-1 uniform ACGT;
per-base level = canonical model mean; dwell = max(min_dwell, Geometric(1/mean_dwell));
signal = level + N(0, 0.25^2), with 200 leading N(0.5, 1) and 100 trailing N(-0.5, 1) samples;
delivered as continuous float64 pA = x * scale + offset (no rounding: tie-free event
detection).  DNA: 4 kHz / 450 b/s (mean dwell 9, min 2, scale 12, offset 90).  RNA: 3 kHz /
70 b/s (mean dwell 43, min 6, scale 80), already in 5'->3' order.
"""
import numpy as np

from . import tombo_helper as th

DNA_SYNTH = dict(mean_dwell=9, min_dwell=2, scale=12.0, offset=90.0)
RNA_SYNTH = dict(mean_dwell=43, min_dwell=6, scale=80.0, offset=500.0)


def synth_read(std_ref, n_bases, seed, mean_dwell=9, min_dwell=2, scale=12.0, offset=90.0,
               noise_sd=0.25, n_lead=200, n_trail=100, lead=None):
    """Returns (genome_seq:str of n_bases+K-1, raw:float64[S], true_starts:int64[n_bases+1])."""
    rng = np.random.default_rng(seed)
    k = std_ref.kmer_width
    codes = rng.integers(0, 4, size=n_bases + k - 1)
    seq = ''.join('ACGT'[c] for c in codes)
    idx = np.zeros(n_bases, dtype=np.int64)
    for j in range(k):
        idx = idx * 4 + codes[j:j + n_bases]
    levels = std_ref.level_means[idx]
    dwell = np.maximum(min_dwell, rng.geometric(1.0 / mean_dwell, size=n_bases))
    body = np.repeat(levels, dwell) + rng.normal(0.0, noise_sd, size=int(dwell.sum()))
    if lead is not None:
        n_lead = lead
    head = rng.normal(0.5, 1.0, size=n_lead)
    tail = rng.normal(-0.5, 1.0, size=n_trail)
    x = np.concatenate([head, body, tail])
    starts = n_lead + np.concatenate([[0], np.cumsum(dwell)])
    return seq, x * scale + offset, starts.astype(np.int64)


def synth_map_res(std_ref, n_bases, seed, **kw):
    """A `resquiggleResults` holding only what mapping would provide."""
    seq, raw, _ = synth_read(std_ref, n_bases, seed, **kw)
    return th.resquiggleResults(
        align_info=th.alignInfo('read_%d' % seed, 'BaseCalled_template', 0, 0, 0, 0,
                                n_bases, 0),
        genome_loc=th.genomeLocation(0, '+', 'synth'), genome_seq=seq, mean_q_score=10.0,
        raw_signal=raw)


def stalled_signal(rng, n, n_stalls, scale=90.0):
    """RNA-like level steps of ~43 samples (70 bases/s at 3 kHz), in raw units, with `n_stalls`
    stalled stretches (flat level + a few units of noise, 150-2500 samples: around the stall
    detector's min_consecutive_obs + window on both sides) inserted at random places."""
    n_lv = n // 20 + 2
    lv = rng.normal(0.0, 1.0, n_lv)
    dwell = np.maximum(6, rng.geometric(1.0 / 43.0, n_lv))
    x = np.repeat(lv, dwell)[:n]
    x = x + rng.normal(0.0, 0.25, x.shape[0])
    raw = x * scale + 500.0
    for _ in range(n_stalls):
        a = int(rng.integers(0, max(1, raw.shape[0] - 3000)))
        ln = int(rng.integers(150, 2500))
        raw[a:a + ln] = raw[a] + rng.normal(0.0, rng.uniform(1.0, 12.0), raw[a:a + ln].shape[0])
    return raw


def edit_read(seq, raw, true_starts, edit):
    """Disagreements between the mapped sequence and the signal, for test reads:
    dict(kind='truncate', frac=f): the signal ends after the first f of the bases;
    dict(kind='cut', n=k): the signal of k bases in the middle is missing (a deletion in the read);
    dict(kind='insert', n=k, seed=s): k extra bases in the middle of the sequence (an insertion
    in the reference)."""
    if not edit:
        return seq, raw
    kind = edit['kind']
    if kind == 'truncate':
        return seq, np.ascontiguousarray(raw[:true_starts[int(len(true_starts) * edit['frac'])]])
    if kind == 'cut':
        a = len(true_starts) // 2
        return seq, np.concatenate([raw[:true_starts[a]], raw[true_starts[a + edit['n']]:]])
    if kind == 'insert':
        rng = np.random.default_rng(edit.get('seed', 5))
        ins = ''.join('ACGT'[c] for c in rng.integers(0, 4, edit['n']))
        a = len(seq) // 2
        return seq[:a] + ins + seq[a:], raw
    raise ValueError('unknown edit %r' % (kind,))
