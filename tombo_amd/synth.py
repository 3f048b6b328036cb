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
