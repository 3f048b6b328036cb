"""The scheme of the chunk-parallel traceback kernel (tombo_amd/csrc/k_tb_par.h), checked on CPU:
tools/tb_par_model.py restates its phases and its chain of agreements over the move matrix the
oracle's forward pass leaves; the stitched walk must equal c_banded_traceback's serial walk, the
walks started in the middle of the band must merge into the true path within a few rows (else the
kernel would be correct but no faster than the lane-per-read walk), a failing read must get the
serial walk's status, and the verifier behind the kernel must catch a phase B that went wrong.  The kernel itself is compared with the oracle's read_tb by the -m gpu parity
tests."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tools'))
import tb_par_model as M  # noqa: E402


@pytest.mark.parametrize('n_bases,bw,seed', [(1500, 200, 0), (1200, 100, 3), (900, 500, 5)])
def test_chunk_parallel_walk_equals_serial(n_bases, bw, seed):
    import oracle
    tb, st, top = M.forward(n_bases, bw, seed)
    rc_o, want_o = oracle.banded_traceback(tb, st, top)          # the oracle's own walk
    rc_s, want = M.serial(tb, st, top)
    assert rc_o == 0 and rc_s == 0
    np.testing.assert_array_equal(want, want_o)
    for lanes in (4, 16, 64):
        rc, got, info = M.chunk_parallel(tb, st, top, lanes)
        assert rc == 0, 'chain broken or error: %r' % (rc,)
        np.testing.assert_array_equal(got, want)
        # the overlap is a few rows (the first adaptive rows after the static start take longer):
        # what makes the scheme pay
        assert len(info['merge_rows']) == info['n_chunks'] - 1, info
        if info['merge_rows']:
            assert np.median(info['merge_rows']) <= 4 and max(info['merge_rows']) < info['chunk'], info


@pytest.mark.parametrize('start_cell', [0, 3, 60, 99])
def test_any_start_cell_gives_the_serial_walk_or_a_broken_chain(start_cell):
    """the start cell of a chunk is a guess: a bad one costs rows (or breaks the chain, which sends
    the read to the serial kernel), never a wrong result"""
    tb, st, top = M.forward(1200, 100, 7)
    rc_s, want = M.serial(tb, st, top)
    assert rc_s == 0
    for lanes in (4, 16):
        rc, got, info = M.chunk_parallel(tb, st, top, lanes, start_cell=start_cell)
        if rc is not None:
            assert rc == 0
            np.testing.assert_array_equal(got, want)


def test_chunk_parallel_walk_reports_the_first_error():
    """band-edge threshold: the status is the one the serial walk stops with -- from the parallel walk where the
    error is a phase A's on the true path, from the serial walk it hands the read to where it is a phase B's"""
    tb, st, top = M.forward(1500, 100, 1)
    seen, how_seen = set(), set()
    for thresh in (0, 1, 5, 20, 40):
        rc_s, want = M.serial(tb, st, top, thresh)
        for lanes in (4, 16, 64):
            rc_p, _, _ = M.chunk_parallel(tb, st, top, lanes, thresh)
            assert rc_p is None or rc_p == rc_s, (thresh, lanes)
            rc, got, how = M.traceback(tb, st, top, lanes, thresh)
            assert rc == rc_s, (thresh, lanes)
            if rc == 0:
                np.testing.assert_array_equal(got, want)
            how_seen.add(how)
        seen.add(rc_s)
    assert seen == {0, 2}   # both outcomes were exercised
    assert 'parallel' in how_seen


@pytest.mark.parametrize('n_bases,bw,seed', [(1500, 200, 0), (900, 500, 5)])
def test_verifier_sends_a_faulty_walk_to_the_serial_kernel(n_bases, bw, seed):
    """k_tb_par_verify (round 6).  Two faults of a phase B, each at one, at several and at all chunk tops: the walk
    coming out of its first row one event too high (what the MI355X did at a 224-VGPR allocation), and a false
    agreement on the first compare (what round 5 took it for).  The chain believes both; the verifier's count is
    non-zero exactly when the result is not the serial walk's, the read then gets the serial walk, and an intact
    read costs nothing"""
    tb, st, top = M.forward(n_bases, bw, seed)
    rc_s, want = M.serial(tb, st, top)
    assert rc_s == 0
    caught = 0
    for lanes in (4, 16):
        rc, good, info = M.chunk_parallel(tb, st, top, lanes)
        assert rc == 0 and M.verify(tb, st, good, lanes) == 0
        np.testing.assert_array_equal(good, want)
        assert M.traceback(tb, st, top, lanes)[2] == 'parallel'
        n = info['n_chunks']
        for kind in ('first_row_plus_one', 'fail_phase_b'):
            for failing in ({0}, {n - 2}, set(range(n - 1)), set(range(0, n - 1, 2))):
                rc, got, _ = M.chunk_parallel(tb, st, top, lanes, **{kind: failing})
                if rc is None:
                    continue                                  # (the faulty walk found no agreement: serial anyway)
                assert rc == 0                                # the chain believes it
                wrong = bool((got != want).any())
                n_diff = M.verify(tb, st, got, lanes)
                assert (n_diff > 0) == wrong, (kind, failing, n_diff)
                rc_t, out, how = M.traceback(tb, st, top, lanes, **{kind: failing})
                assert rc_t == 0 and how == ('serial: verifier' if wrong else 'parallel')
                np.testing.assert_array_equal(out, want)
                caught += wrong
    assert caught > 0   # (the injected faults do leave wrong rows somewhere)
