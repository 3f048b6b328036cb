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


# ---- the device generator (csrc/k_synth.h), restated in numpy ----------------------------------
# Test infrastructure for tba_synth_generate / _native.Synth: same draws, same IEEE operations in the
# same order, so the arrays are equal bit for bit.  (The product path never calls this.)
_U64 = np.uint64


def _synth_hash(x):
    with np.errstate(over='ignore'):
        x = x + _U64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> _U64(27))) * _U64(0x94D049BB133111EB)
        return x ^ (x >> _U64(31))


def _synth_draw(key, stream, idx):
    with np.errstate(over='ignore'):
        return _synth_hash(key + _U64(stream << 40) + idx.astype(np.uint64))


def device_synth_tables(mean_dwell):
    """(dwell thresholds uint32[256], noise constant): thr[k] = floor(2^32 (1 - q^(k+1))) by repeated
    multiplication, c = 1 / sqrt((65536^2 - 1) / 3)"""
    q = np.float64(1.0) - np.float64(1.0) / np.float64(mean_dwell)
    thr = np.zeros(256, np.uint32)
    t = np.float64(1.0)
    for k in range(256):
        t = t * q
        v = np.floor(np.float64(4294967296.0) * (np.float64(1.0) - t))
        thr[k] = 0xffffffff if v >= 4294967295.0 else int(v)
    return thr, float(np.float64(1.0) / np.sqrt((np.float64(65536.0) * np.float64(65536.0) - np.float64(1.0)) / np.float64(3.0)))


def device_reads_reference(std_ref, seed, n_bases, raw_dtype=np.int16, first_read=0, mean_dwell=9,
                           min_dwell=2, scale=12.0, offset=90.0, noise_sd=0.25, n_lead=200, n_trail=100,
                           dac_per_pa=1.0 / 0.1709, dac_offset=10.0, reverse=False):
    """The batch tba_synth_generate makes for these arguments: (list of raw arrays, list of code arrays)."""
    thr, c = device_synth_tables(mean_dwell)
    k = std_ref.kmer_width
    means = np.asarray(std_ref.level_means, dtype=np.float64)
    raws, codes_out = [], []
    with np.errstate(over='ignore'):
        seed_h = _synth_hash(_U64(int(seed) & 0xffffffffffffffff))
    for i, nb in enumerate(n_bases):
        nb = int(nb)
        with np.errstate(over='ignore'):
            key = _synth_hash(seed_h + _U64(first_read + i))
        codes = ((_synth_draw(key, 0, np.arange(nb + k - 1)) >> _U64(11)) & _U64(3)).astype(np.int64)
        u = (_synth_draw(key, 1, np.arange(nb)) >> _U64(32)).astype(np.uint32)
        dwell = np.maximum(1 + np.searchsorted(thr, u, side='right'), min_dwell).astype(np.int64)
        idx = np.zeros(nb, dtype=np.int64)
        for j in range(k):
            idx = idx * 4 + codes[j:j + nb]
        body = int(dwell.sum())
        S = n_lead + body + n_trail
        h = _synth_draw(key, 2, np.arange(S))
        m16 = _U64(0xffff)
        tot = ((h & m16) + ((h >> _U64(16)) & m16) + ((h >> _U64(32)) & m16) + (h >> _U64(48))).astype(np.int64)
        noise = (tot - 131070).astype(np.float64) * np.float64(c)
        x = np.empty(S, np.float64)
        x[:n_lead] = np.float64(0.5) + noise[:n_lead]
        x[n_lead:n_lead + body] = np.repeat(means[idx], dwell) + noise[n_lead:n_lead + body] * np.float64(noise_sd)
        x[n_lead + body:] = np.float64(-0.5) + noise[n_lead + body:]
        pa = x * np.float64(scale) + np.float64(offset)
        if np.dtype(raw_dtype) == np.int16:
            out = np.clip(np.rint(pa * np.float64(dac_per_pa) + np.float64(dac_offset)), -32768.0, 32767.0).astype(np.int16)
        else:
            out = pa
        raws.append(np.ascontiguousarray(out[::-1]) if reverse else out)
        codes_out.append(codes.astype(np.uint8))
    return raws, codes_out
