"""One resident batch, run again and again, must give the same bytes -- at the sizes where the machine holds
several rounds of wavefronts.  Round 5's chunk-parallel traceback did not (a few wavefronts per 10 000-read RNA
batch, differently from run to run, every 48-read parity test green); round 6 found what it hung on
(profiles/r06_traceback_rootcause.txt: the kernel's VGPR allocation) and put a verifier behind the kernel whose
count has to stay at zero.  These tests hold both: the bytes, and the count.

    TBA_LIB_PATH=<a build> python -m pytest tests/test_gpu_determinism.py -m gpu -q

runs them against another build of the library: profiles/r06_traceback_determinism_alloc224.txt is this file
FAILING on a -DTBA_TB_ALLOC224 -DTBA_NO_TB_VERIFY build (the round-5 kernel's allocation, no verifier)."""
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
pytestmark = pytest.mark.gpu


def _device_batch(samp_name, n_reads, n_bases, seed, bandwidth=500):
    """n_reads synthetic reads made on the device (tba_synth_*), resident in one engine"""
    from tombo_amd import _native as N, tombo_stats as ts, tombo_helper as th, synth
    from tombo_amd._default_parameters import SIG_MATCH_THRESH, STALL_PARAMS
    rna = samp_name == 'RNA'
    samp = th.seqSampleType(samp_name, rna)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=bandwidth)
    if bandwidth <= 100:
        params = params._replace(band_bound_thresh=10)   # (the default 40 fails every read at W = 100: bench.py does the same)
    sp = N.make_synth_params(**(synth.RNA_SYNTH if rna else synth.DNA_SYNTH))
    gen = N.Synth(model, 0)
    raw, raw_off, seq, seq_off = gen.generate(sp, seed, np.full(n_reads, n_bases, np.int64))
    eng = N.Engine(0)
    eng.ensure_model(model)
    eng.upload_packed(N.make_params(params),
                      N.make_opts(outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH[samp_name],
                                  stall_params=th.stallParams(**STALL_PARAMS) if rna else None, subsample_seed=seed),
                      raw, raw_off, seq, seq_off)
    return eng, gen, model, params, raw_off, seq_off


@pytest.mark.parametrize('samp_name,n_reads,n_bases,n_runs,bandwidth', [
    ('RNA', 10240, 3000, 32, 500), ('DNA', 8192, 10000, 8, 500),
    ('DNA', 8192, 10000, 4, 300),       # Tombo's default band: k_dp<5>
    ('DNA', 16384, 2000, 8, 100)])      # BASELINE config 1's shape: k_dp_multi<4, 2>, two reads per wavefront
def test_a_resident_batch_gives_the_same_bytes_every_run(samp_name, n_reads, n_bases, n_runs, bandwidth):
    """>= 10 000 RNA / >= 8 192 DNA reads from the device generator, 32 / 8 runs (the round-5 fault showed in one RNA
    run in five), and the two narrower bands of the benchmark's other configurations: read_tb, boundaries, status
    equal in every run, the verifier's count zero in every run, and nearly every read walked chunk-parallel"""
    from tombo_amd import _native as N
    eng, gen, model, params, raw_off, seq_off = _device_batch(samp_name, n_reads, n_bases, 20261001, bandwidth)
    runs, vfail = [], []
    for _ in range(n_runs):
        eng.run()
        out = eng.download(want_norm=False)
        tb = eng.get(N.GET_READ_TB)
        vfail.append(int(eng.get(N.GET_TB_VERIFY_FAIL).sum()))
        runs.append((zlib.crc32(tb.tobytes()), zlib.crc32(out['segs'].tobytes()), out['status'].tobytes()))
    form = eng.get(N.GET_TB_FORM)
    ok = out['status'] == 0
    eng.close(), gen.close()
    assert ok.sum() > 0.95 * n_reads, int(ok.sum())
    if bandwidth > 128:     # (narrower bands: k_dp_multi's reads have no centre strip and cpl_class 0 leaves them to the lane walk)
        assert (form[ok] == N.TB_FORM_PAR16).sum() > 0.9 * ok.sum(), np.bincount(form)
    assert vfail == [0] * n_runs, 'the verifier of the chunk-parallel traceback disagreed: %s rows per run' % vfail
    assert len(set(runs)) == 1, 'run-dependent results: %s' % [r[:2] for r in runs]


def test_late_reads_of_a_big_rna_batch_equal_the_oracle():
    """the reads whose wavefronts start on an already busy SIMD (the ones round 5 got wrong), against the oracle"""
    import oracle
    from tombo_amd import _native as N
    from tombo_amd._default_parameters import SIG_MATCH_THRESH
    n = 6144
    eng, gen, model, params, raw_off, seq_off = _device_batch('RNA', n, 1500, 99)
    eng.run()
    out = eng.download(want_norm=False)
    si = eng.get(N.GET_SAMP_IND)
    h_raw, h_seq = gen.download()
    segs, seg_off = out['segs'], np.asarray(eng.seg_off)
    assert int(eng.get(N.GET_TB_VERIFY_FAIL).sum()) == 0
    checked = 0
    for i in list(range(4100, n, 157)) + [0, 1, 4095, 4096]:
        raw = h_raw[raw_off[i]:raw_off[i + 1]].astype(np.float64)
        want = oracle.resquiggle_read(
            raw, h_seq[seq_off[i]:seq_off[i + 1]], model.level_means, model.level_sds, oracle.make_params(params),
            oracle.make_opts(model.kmer_width, model.central_pos, outlier_thresh=5.0, sig_match_thresh=SIG_MATCH_THRESH['RNA']),
            stall_ints=oracle.identify_stalls(raw), samp_ind=si[i])
        assert want['status'] == int(out['status'][i]), (i, want['status'], int(out['status'][i]))
        if want['status'] == 0:
            np.testing.assert_array_equal(segs[seg_off[i]:seg_off[i + 1]], want['segs'], err_msg='read %d' % i)
            checked += 1
    eng.close(), gen.close()
    assert checked >= 10


_INJECT_CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + '/tests')
import oracle
from tombo_amd import _native as N
from tombo_amd._default_parameters import SIG_MATCH_THRESH
from test_gpu_determinism import _device_batch
n = 1400
eng, gen, model, params, raw_off, seq_off = _device_batch('DNA', n, 2500, 7)
eng.run()
out = eng.download(want_norm=False)
vf, form, path = eng.get(N.GET_TB_VERIFY_FAIL), eng.get(N.GET_TB_FORM), eng.get(N.GET_PATH)[:, 0]
ok = out['status'] == 0
hit = np.flatnonzero(vf)
# the build injects its fault into every 5th read (index %% 5 == 3) that was walked chunk-parallel
cand = np.flatnonzero(ok & (path == 1) & (np.arange(n) %% 5 == 3))
assert hit.size >= 0.9 * cand.size and set(hit.tolist()) <= set(cand.tolist()), (hit.size, cand.size)
assert np.all(vf[hit] == 1) and np.all(form[hit] == N.TB_FORM_LANE), (np.unique(vf[hit]), np.unique(form[hit]))
assert np.all(form[ok & (path == 1) & (vf == 0) & (np.arange(n) %% 5 != 3)] == N.TB_FORM_PAR16)
h_raw, h_seq = gen.download()
si = eng.get(N.GET_SAMP_IND)
seg_off = np.asarray(eng.seg_off)
for i in hit[:6].tolist() + [0, 1, 2]:
    want = oracle.resquiggle_read(
        h_raw[raw_off[i]:raw_off[i + 1]].astype(np.float64), h_seq[seq_off[i]:seq_off[i + 1]], model.level_means, model.level_sds,
        oracle.make_params(params), oracle.make_opts(model.kmer_width, model.central_pos, outlier_thresh=5.0,
                                                     sig_match_thresh=SIG_MATCH_THRESH['DNA']), samp_ind=si[i])
    assert want['status'] == int(out['status'][i]) == 0, i
    assert np.array_equal(out['segs'][seg_off[i]:seg_off[i + 1]], want['segs']), i
print('INJECT_OK %%d reads caught of %%d injected' %% (hit.size, cand.size))
'''


def test_the_verifier_catches_an_injected_fault_and_the_read_is_walked_again():
    """libtombo_amd_inject.so (built by build() beside the library; -DTBA_TB_INJECT=5) puts the round-5 fault -- the
    first row under a chunk top one event too high -- into every fifth read.  In a child process on that build: the
    verifier's count names exactly those reads, they are walked again by the lane-per-read kernel, and every result
    is the oracle's."""
    from tombo_amd import _native
    if not os.path.exists(_native.INJECT_LIB_PATH):
        pytest.fail('%s is missing: run `python -c "import __graft_entry__ as g; g.build()"`' % _native.INJECT_LIB_PATH)
    env = dict(os.environ, TBA_LIB_PATH=_native.INJECT_LIB_PATH)
    p = subprocess.run([sys.executable, '-c', _INJECT_CHILD % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'INJECT_OK' in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
