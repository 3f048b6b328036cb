"""The scheme of the chunk-parallel traceback kernel (tombo_amd/csrc/k_tb_par.h), checked on CPU:
tools/tb_par_model.py restates its phases and its chain of agreements over the move matrix the
oracle's forward pass leaves; the stitched walk must equal c_banded_traceback's serial walk, the
walks started in the middle of the band must merge into the true path within a few rows (else the
kernel would be correct but no faster than the lane-per-read walk), and a failing read must get the
serial walk's status.  The kernel itself is compared with the oracle's read_tb by the -m gpu parity
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
    """band-edge threshold: the status is the one the serial walk stops with"""
    tb, st, top = M.forward(1500, 100, 1)
    seen = set()
    for thresh in (0, 1, 5, 20, 40):
        rc_s, _ = M.serial(tb, st, top, thresh)
        for lanes in (4, 16, 64):
            rc, _, _ = M.chunk_parallel(tb, st, top, lanes, thresh)
            assert rc == rc_s, (thresh, lanes)
        seen.add(rc_s)
    assert seen == {0, 2}   # both outcomes were exercised


@pytest.mark.parametrize('n_bases,bw,seed', [(1500, 200, 0), (900, 500, 5)])
def test_repair_pass_restores_the_serial_walk(n_bases, bw, seed):
    """k_tb_par_repair (round 5): with the phase B of some -- or all -- lanes ending on its first
    compare, the speculative rows under the chunk tops stay and the walk is wrong; a second phase B
    over the finished array (state from the entry above each chunk top) overwrites exactly those rows,
    and a pass over an intact array changes nothing"""
    tb, st, top = M.forward(n_bases, bw, seed)
    rc_s, want = M.serial(tb, st, top)
    assert rc_s == 0
    shown = 0
    for lanes in (4, 16):
        rc, good, info = M.chunk_parallel(tb, st, top, lanes)
        assert rc == 0
        rc_r, n_over = M.repair(tb, st, good, lanes)
        assert (rc_r, n_over) == (0, 0)
        np.testing.assert_array_equal(good, want)
        n = info['n_chunks']
        for failing in ({0}, {n - 2}, set(range(n - 1)), set(range(0, n - 1, 2))):
            rc, got, _ = M.chunk_parallel(tb, st, top, lanes, fail_phase_b=failing)
            assert rc == 0                                   # (the chain believes the false agreements)
            speculative = int((got != want).sum())
            shown += speculative
            rc_r, n_over = M.repair(tb, st, got, lanes)
            assert rc_r == 0
            np.testing.assert_array_equal(got, want)
            assert n_over == speculative                     # only the rows that were wrong are written
            assert M.repair(tb, st, got, lanes) == (0, 0)    # idempotent
    assert shown > 0   # (the injected failure does leave wrong rows somewhere)
