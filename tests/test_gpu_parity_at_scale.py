"""Parity with the oracle where the handful-of-reads tests do not reach (round 6).

1. k_theil_sen's one-pass median (sorted points, blocks of 64 rows against chunks of 64 partners, mirrored pairs of row
   blocks) at point counts that exercise every shape of its enumeration: fewer than TSW_MIN_POINTS (the generic two-pass
   select), an odd number of row blocks (the middle block has no mirror), a last block of one row, exactly 1000 and the
   subsampled case -- every read against the oracle (status, boundaries, scale values, score).
2. EVERY read of a 4 096-read RNA and a 2 048-read DNA batch against the oracle (the oracle on the host's threads)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('samp_name,n_bases', [
    ('DNA', 200), ('DNA', 256), ('DNA', 257), ('DNA', 300), ('DNA', 320), ('DNA', 321), ('DNA', 449), ('DNA', 512),
    ('DNA', 577), ('DNA', 640), ('DNA', 705), ('DNA', 961), ('DNA', 999), ('DNA', 1000), ('DNA', 1001), ('RNA', 385), ('RNA', 833)])
def test_reads_of_n_bases_equal_the_oracle(samp_name, n_bases):
    import oracle
    from tombo_amd import _native as N
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    from test_gpu_determinism import _device_batch
    n = 6
    eng, gen, model, params, raw_off, seq_off = _device_batch(samp_name, n, n_bases, 1000 + n_bases)
    eng.run()
    out = eng.download(want_norm=False)
    si = eng.get(N.GET_SAMP_IND)
    h_raw, h_seq = gen.download()
    segs, seg_off = out['segs'], np.asarray(eng.seg_off)
    checked = 0
    for i in range(n):
        raw = h_raw[raw_off[i]:raw_off[i + 1]].astype(np.float64)
        want = oracle.resquiggle_read(
            raw, h_seq[seq_off[i]:seq_off[i + 1]], model.level_means, model.level_sds, oracle.make_params(params),
            oracle.make_opts(model.kmer_width, model.central_pos, outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH[samp_name]),
            stall_ints=oracle.identify_stalls(raw) if samp_name == 'RNA' else None, samp_ind=si[i] if n_bases > 1000 else None)
        assert want['status'] == int(out['status'][i]), (i, want['status'], int(out['status'][i]))
        if want['status'] == 0:
            np.testing.assert_array_equal(segs[seg_off[i]:seg_off[i + 1]], want['segs'], err_msg='read %d' % i)
            # shift, scale (the Theil-Sen fit's output), lower / upper limit: the same bits
            np.testing.assert_array_equal(np.asarray(out['sv'][i], np.float64).view(np.int64),
                                          np.asarray(want['scale_values'], np.float64).view(np.int64), err_msg='read %d' % i)
            assert int(out['read_start'][i]) == want['read_start_rel_to_raw']
            assert float(out['score'][i]) == want['sig_match_score']
            checked += 1
    eng.close(), gen.close()
    assert checked >= 1


@pytest.mark.parametrize('samp_name,n,n_bases,bandwidth', [
    ('RNA', 4096, 3000, 500), ('DNA', 2048, 10000, 500),
    ('DNA', 1024, 10000, 300),      # Tombo's default band: k_dp<5>
    ('DNA', 4096, 2000, 100)])      # BASELINE config 1's shape: k_dp_multi<4, 2>, two reads per wavefront
def test_every_read_of_a_batch_equals_the_oracle(samp_name, n, n_bases, bandwidth):
    """4 096 RNA reads of 3 kb / 2 048 DNA reads of 10 kb (and the benchmark's two narrower bands), EVERY one against the oracle
    (status, boundaries, scale values).
    About 6 % of such RNA reads have a skip window of the largest wave class (k_skip_dp_wave<1792>: its signal from global
    memory, round 6), a quarter one of the middle class -- the handful of reads of the other RNA parity tests may have none"""
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    from tombo_amd import _native as N
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    from test_gpu_determinism import _device_batch
    rna = samp_name == 'RNA'
    eng, gen, model, params, raw_off, seq_off = _device_batch(samp_name, n, n_bases, 4242, bandwidth)
    eng.run()
    out = eng.download(want_norm=False)
    si = eng.get(N.GET_SAMP_IND)
    h_raw, h_seq = gen.download()
    segs, seg_off = out['segs'], np.asarray(eng.seg_off)
    p_, o_ = oracle.make_params(params), oracle.make_opts(model.kmer_width, model.central_pos, outlier_thresh=5.0,
                                                          sig_match_thresh=SIG_MATCH_THRESH[samp_name])

    def one(i):   # (the C restatement runs with the GIL released)
        raw = h_raw[raw_off[i]:raw_off[i + 1]].astype(np.float64)
        return oracle.resquiggle_read(raw, h_seq[seq_off[i]:seq_off[i + 1]], model.level_means, model.level_sds, p_, o_,
                                      stall_ints=oracle.identify_stalls(raw) if rna else None, samp_ind=si[i])
    with ThreadPoolExecutor(min(32, os.cpu_count() or 8)) as ex:
        wants = list(ex.map(one, range(n)))
    bad = []
    for i, want in enumerate(wants):
        if want['status'] != int(out['status'][i]):
            bad.append('read %d: status %d, oracle %d' % (i, int(out['status'][i]), want['status']))
        elif want['status'] == 0 and not np.array_equal(segs[seg_off[i]:seg_off[i + 1]], want['segs']):
            bad.append('read %d: boundaries differ' % i)
        elif want['status'] == 0 and not np.array_equal(np.asarray(out['sv'][i]).view(np.int64), np.asarray(want['scale_values']).view(np.int64)):
            bad.append('read %d: scale values differ' % i)
    eng.close(), gen.close()
    assert not bad, '\n'.join(bad[:20])
    assert sum(w['status'] == 0 for w in wants) > 0.9 * n, sum(w['status'] == 0 for w in wants)
