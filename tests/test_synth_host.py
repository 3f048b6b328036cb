"""Host side of the device read generator (csrc/k_synth.h): the tables the library builds equal the
numpy restatement's, and the restated reads behave like the reads of synth.synth_read (the GPU test
test_gpu_synth.py holds the device to this restatement bit for bit)."""
import numpy as np


def test_tables_of_the_library_and_of_the_restatement_agree():
    from tombo_amd import _native, synth
    for md in (9, 43, 2, 200):
        thr, c = _native.synth_tables(_native.make_synth_params(mean_dwell=md))
        thr2, c2 = synth.device_synth_tables(md)
        assert np.array_equal(thr, thr2) and c == c2
        assert np.all(np.diff(thr.astype(np.int64)) >= 0)


def test_restated_reads_look_like_synth_reads():
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    model = ts.TomboModel(seq_samp_type=th.seqSampleType('DNA', False))
    raws, codes = synth.device_reads_reference(model, 3, [4000, 4000], raw_dtype=np.float64)
    raws_b, codes_b = synth.device_reads_reference(model, 3, [4000], first_read=1, raw_dtype=np.float64)
    assert np.array_equal(raws[1], raws_b[0]) and np.array_equal(codes[1], codes_b[0])   # f(seed, index)
    assert not np.array_equal(codes[0][:200], codes[1][:200])
    other, _ = synth.device_reads_reference(model, 4, [4000], raw_dtype=np.float64)
    assert len(other[0]) != len(raws[0]) or not np.array_equal(other[0], raws[0])
    for r, c in zip(raws, codes):
        dwell = (len(r) - 300) / 4000.0
        assert 8.3 < dwell < 10.3                          # max(2, Geometric(1/9)): mean 9.2
        assert np.bincount(c, minlength=4).min() > 800     # uniform bases
        x = (r - 90.0) / 12.0
        assert abs(x[:200].mean() - 0.5) < 0.3 and abs(x[-100:].mean() + 0.5) < 0.4
        assert 0.7 < x[:200].std() < 1.3
    # the noise around the levels: unit-variance Irwin-Hall scaled by noise_sd
    k = model.kmer_width
    idx = np.zeros(4000, np.int64)
    for j in range(k):
        idx = idx * 4 + codes[0][j:j + 4000].astype(np.int64)
    quiet, _ = synth.device_reads_reference(model, 3, [4000], raw_dtype=np.float64, noise_sd=0.0)
    resid = (raws[0] - quiet[0])[200:-100] / 12.0
    assert abs(resid.std() - 0.25) < 0.01 and abs(resid.mean()) < 0.01 and np.abs(resid).max() <= 0.25 * 3.47


def test_restatement_is_pinned():
    """the draws of csrc/k_synth.h as recorded when the device generator was held to this restatement on
    an MI355X (tests/test_gpu_synth.py): a change of either side shows here without a GPU"""
    import hashlib
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    model = ts.TomboModel(seq_samp_type=th.seqSampleType('DNA', False))
    raws, codes = synth.device_reads_reference(model, 20260927, [700, 1300], first_read=41)
    h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    assert [len(r) for r in raws] == [6916, 12763]
    assert [h(r) for r in raws] == ['ff36502a6f096453', 'dbf06206530e2e7f']
    assert [h(c) for c in codes] == ['4fb2e7d2adf0ba47', '144199660e1e925c']
    assert raws[0][:6].tolist() == [671, 589, 551, 486, 659, 656] and codes[0][:8].tolist() == [2, 2, 0, 0, 3, 3, 3, 1]
