"""CPU model of the chunk-parallel traceback (tombo_amd/csrc/k_tb_par.h): the same phases and the
same chain of agreements over a plain move matrix tb[row, band cell] (0 stay / 1 skip / 2 diagonal)
and the band starts, as the oracle's forward pass leaves them.  Used by
tests/test_tb_par_model.py to check the scheme (walks started in the middle of the band merge into
the true path within a few rows, the stitched walk equals the serial traceback, errors are the
serial walk's) without a GPU, and from the command line to print how many rows the merges take:
python tools/tb_par_model.py [n_bases] [bandwidth]"""
import numpy as np

NONE = 1 << 62
OK, INTERNAL, BEYOND = 0, 1, 2


def _step(tb, starts, bw, rr, cur_ev, thresh):
    """one row of c_banded_traceback (pyx:293-308); returns (rc, new cur_ev, band-edge violation)"""
    st = int(starts[rr - 1])
    bp = cur_ev - st
    if bp >= bw or bp < -bw:
        return INTERNAL, cur_ev, False
    while tb[rr, bp] == 0:        # (numpy wraps a negative index like Python)
        bp -= 1
        if bp < -bw:
            return INTERNAL, cur_ev, False
    if tb[rr, bp] == 2:
        bp -= 1
    viol = thresh >= 0 and min(bp, bw - bp - 1) < thresh
    return OK, st + bp, viol


def serial(tb, starts, top_pos, thresh=-1):
    n, bw = tb.shape[0] - 1, tb.shape[1]
    out = np.zeros(n + 1, np.int64)
    cur = top_pos + int(starts[n - 1])
    out[n] = cur + 1
    for rr in range(n, 0, -1):
        rc, cur, viol = _step(tb, starts, bw, rr, cur, thresh)
        if rc:
            return rc, out
        if viol:
            return BEYOND, out
        out[rr - 1] = cur + 1
    return OK, out


def chunk_parallel(tb, starts, top_pos, lanes=16, thresh=-1, min_chunk=64, start_cell=None, n_static=100,
                   fail_phase_b=(), first_row_plus_one=()):
    """k_main_tb_par: returns (rc, read_tb, info); rc None = the read is left to the serial walk (the chain
    broke, or its status would rest on a phase B: an error of one).  n_static: rows with a static band at
    the start of the read (the path is anywhere in those bands, so they all go to the lowest chunk).
    Faults, for the verifier's test (`verify`): fail_phase_b -- chunks whose lane's phase B ends on its first
    compare as if it had found agreement there (what round 5 took the GPU's failure for);
    first_row_plus_one -- chunks whose lane's phase B comes out of its first row one event too high and walks
    on from there (what the failure was: profiles/r06_traceback_rootcause.txt)"""
    B, bw = tb.shape[0] - 1, tb.shape[1]
    top_rows = max(B - (n_static + 16), 1)
    L = max((top_rows + lanes - 1) // lanes, min_chunk)
    n_chunks = (top_rows + L - 1) // L
    hi = [B - c * L for c in range(n_chunks)]
    lo = [max(h - L, 0) for h in hi]
    lo[-1] = 0
    out = np.full(B + 1, -(1 << 40), np.int64)    # (stale values never equal a state)
    start = [top_pos + int(starts[B - 1])] + [int(starts[hi[c] - 1]) + (bw // 2 if start_cell is None else start_cell)
                                               for c in range(1, n_chunks)]
    out[B] = start[0] + 1
    # phase A
    cur, rcA, viol_lo, wrote_lo = list(start), [OK] * n_chunks, [NONE] * n_chunks, list(lo)
    for c in range(n_chunks):
        for rr in range(hi[c], lo[c], -1):
            rc, nxt, viol = _step(tb, starts, bw, rr, cur[c], thresh)
            if rc:
                rcA[c], wrote_lo[c] = rc, hi[c]
                break
            if viol:
                viol_lo[c] = rr
            cur[c] = nxt
            out[rr - 1] = nxt + 1
    # phase B
    rcB, merged, merge_rows = [OK] * n_chunks, [NONE] * n_chunks, []
    rec = out.copy()                               # (every lane compares with the phase-A record)
    for c in range(n_chunks - 1):
        if rcA[c]:
            continue
        if cur[c] == start[c + 1] or c in fail_phase_b:
            merged[c] = lo[c]
            merge_rows.append(0)
            continue
        for rr in range(lo[c], lo[c + 1], -1):
            rc, nxt, viol = _step(tb, starts, bw, rr, cur[c], thresh)
            if rc or viol:
                rcB[c] = rc if rc else BEYOND
                break
            if rr == lo[c] and c in first_row_plus_one:
                nxt += 1
            cur[c] = nxt
            if rr - 1 >= wrote_lo[c + 1] and rec[rr - 1] == nxt + 1:
                merged[c] = rr - 1
                merge_rows.append(lo[c] - (rr - 1))
                break
            out[rr - 1] = nxt + 1
    # the chain
    status, true_from = OK, B + 1
    for j in range(n_chunks):
        if viol_lo[j] != NONE and viol_lo[j] <= true_from:
            status = BEYOND
            break
        if rcA[j]:
            status = rcA[j]
            break
        if j == n_chunks - 1:
            break
        if rcB[j] or merged[j] == NONE:
            return None, out, dict(merge_rows=merge_rows)
        true_from = merged[j]
    return status, out, dict(merge_rows=merge_rows, chunk=L, n_chunks=n_chunks)


def verify(tb, starts, out, lanes=16, min_chunk=64, n_static=100, rows=16):
    """k_tb_par_verify: under every chunk top the first `rows` rows are walked again from the entry above the
    top (compare only).  Returns the number of rows where `out` holds something else; the kernel sends a read
    with a non-zero count to the serial walk."""
    B, bw = tb.shape[0] - 1, tb.shape[1]
    top_rows = max(B - (n_static + 16), 1)
    L = max((top_rows + lanes - 1) // lanes, min_chunk)
    n_chunks = (top_rows + L - 1) // L
    hi = [B - c * L for c in range(n_chunks)]
    lo = [max(h - L, 0) for h in hi]
    lo[-1] = 0
    n_diff = 0
    for c in range(n_chunks - 1):
        if lo[c] < 1:
            continue
        cur = int(out[lo[c]]) - 1
        for rr in range(lo[c], max(lo[c] - rows, lo[c + 1]), -1):
            rc, cur, _ = _step(tb, starts, bw, rr, cur, -1)
            if rc:
                n_diff += 1
                break
            if out[rr - 1] != cur + 1:
                n_diff += 1
    return n_diff


def traceback(tb, starts, top_pos, lanes=16, thresh=-1, **faults):
    """what the engine does with a read: chunk-parallel walk, verifier, serial walk where either gives the read
    back.  Returns (rc, read_tb, how) with how in 'parallel', 'serial: left by the parallel walk',
    'serial: verifier'."""
    rc, out, _ = chunk_parallel(tb, starts, top_pos, lanes, thresh, **faults)
    if rc is None:
        return serial(tb, starts, top_pos, thresh) + ('serial: left by the parallel walk',)
    if rc == OK and verify(tb, starts, out, lanes):
        return serial(tb, starts, top_pos, thresh) + ('serial: verifier',)
    return rc, out, 'parallel'


def forward(n_bases=1500, bw=200, seed=0, static_rows=100):
    """an adaptive banded forward pass of the oracle over a synthetic DNA read: (moves, starts, top)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    import oracle
    from tombo_amd import tombo_stats as ts, tombo_helper as th, synth
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)
    mr = synth.synth_map_res(model, n_bases, seed, **synth.DNA_SYNTH)
    o = oracle.resquiggle_read(
        mr.raw_signal, ts.encode_seq(mr.genome_seq), model.level_means, model.level_sds,
        oracle.make_params(params), oracle.make_opts(model.kmer_width, model.central_pos, outlier_thresh=5.0),
        samp_ind=None if n_bases <= 1000 else np.random.default_rng(seed).choice(n_bases, 1000, replace=False),
        debug=True)
    assert o['status'] == 0, o['status']
    ev = o['dbg']['event_means']
    codes = ts.encode_seq(mr.genome_seq)
    K = model.kmer_width
    kidx = np.zeros(n_bases, np.int64)
    for j in range(K):
        kidx = kidx * 4 + codes[j:j + n_bases]
    mu, sd = model.level_means[kidx], model.level_sds[kidx]
    p = params
    nb = static_rows
    z = np.empty((nb, bw))
    for r in range(nb):
        z[r] = p.z_shift - np.minimum(p.max_half_z_score, np.abs(ev[r:r + bw] - mu[r]) / sd[r])
    starts0 = np.arange(nb, dtype=np.int64)
    fwd, tb = oracle.banded_forward_pass(z, starts0, p.skip_pen, p.stay_pen)
    fwd_a = np.zeros((n_bases + 1, bw))
    tb_a = np.zeros((n_bases + 1, bw), dtype=np.int8)
    st_a = np.zeros(n_bases, dtype=np.int64)
    fwd_a[:nb + 1], tb_a[:nb + 1], st_a[:nb] = fwd, tb, starts0
    rc = oracle.adaptive_banded_forward_pass(fwd_a, tb_a, st_a, ev, mu, sd, p.z_shift, p.skip_pen,
                                             p.stay_pen, nb, -15.0, True, p.max_half_z_score)
    assert rc == 0, rc
    return tb_a.astype(np.int64), st_a, int(np.argmax(fwd_a[-1]))


if __name__ == '__main__':
    import sys
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    bw = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    tb, st, top = forward(nb, bw)
    rc0, want = serial(tb, st, top, 5)
    for lanes in (8, 16, 64):
        rc, got, info = chunk_parallel(tb, st, top, lanes, 5)
        print('lanes', lanes, 'rc', rc, 'serial rc', rc0, 'equal', rc0 != 0 or bool(np.array_equal(got, want)), info)
