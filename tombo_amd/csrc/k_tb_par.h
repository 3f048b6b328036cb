// k_tb_par.h -- the main traceback (c_banded_traceback, pyx:281-310), rows of one read walked by
// several lanes at once.
//
// The walk is a pointer chase: the state entering row rr is the event position cur_ev, and what it
// is on leaving depends on the move codes of that row only.  k_main_tb walks the B rows of a read
// on one lane (10 000 dependent steps: 4.8 ms for a batch of 10 kb reads however few reads there
// are; 49 ms for a 200 kb read).  Here a read is cut into chunks of rows with a lane each:
//   phase A: the lane of chunk c walks the rows of its chunk, top to bottom, writing read_tb.
//     Chunk 0 starts from the end of the forward pass (the true walk); a chunk c >= 1 starts at
//     the MIDDLE of its top row's band (cell W / 2): the adaptive band is placed so that the best
//     cell of the previous row sits there, and traceback paths that start near each other merge
//     within a few rows (measured on synthetic 10 kb reads: median 1 row, maximum 12 from W / 2;
//     from an end of the band 200-300 rows) -- so all but the first few rows of what the lane
//     writes are the true path already, only nobody knows yet.
//   phase B: the lane of chunk c walks on into chunk c + 1, compares its state after every row
//     with what the lane below recorded for that row, overwrites where they differ and stops where
//     they agree.  A walk is a deterministic function of its state, so from an agreement on the
//     recorded path IS the continuation of this lane's path.
//   chain (one lane per read): chunk 0 is the true walk; if its lane merged into chunk 1's
//     record, that record is true from the merge row down, in particular where chunk 1's lane
//     started ITS phase B, and so on down the read.  The result is the serial walk row for row,
//     by induction, whatever the start cells were worth; a read whose chain breaks (a lane that
//     finds no agreement inside the next chunk) is left to k_main_tb, untouched (never seen on
//     synthetic reads: tests/test_gpu_parity.py asserts it through TBA_GET_TB_PARALLEL).
// Errors of the serial walk are errors on the TRUE path only: those of chunk 0's phase A, of any
// phase B, and of a chunk's phase A at or below the row where the lane above merged into it; phase
// A therefore walks on through band-edge violations and records the lowest row that had one.  The
// first error in walk order gives the status (a violation comes before the walk's death).
//
// What holds the result to the serial walk's, row for row:
//   * the scheme: tools/tb_par_model.py restates it over the oracle's move matrices
//     (tests/test_tb_par_model.py, CPU); on the GPU every parity test compares read_tb with the oracle's;
//   * k_tb_par_verify, behind the kernel boundary: under every chunk top it walks the first TBR rows
//     again from the entry above (compare only) and hands any read with a disagreement to the serial
//     kernels, counted (ReadState.tb_verify_fail, TBA_GET_TB_VERIFY_FAIL: asserted zero by the GPU tests
//     and by bench.py).  A status that rests on a phase B (its error, or a broken chain) is never final
//     either: such a read is the serial kernels' too.  _trim_traceback and top_pos are written by the
//     verifier, after it has agreed -- until then the serial walk can still start from top_pos;
//   * the register allocation.  Round 5 found a few wavefronts per 10 000-read RNA batch whose result
//     depended on the run and "fixed" them with two repair passes; round 6 found what it was
//     (profiles/r06_traceback_rootcause.txt).  Not the handoff through memory: with read_tb poisoned
//     before the kernel, and with phase B's stores diverted into a second array, phase B entered with
//     the same state every run and still computed a different FIRST ROW (path one event too high in
//     every lane of the wavefront where the move was a diagonal) -- a computation, not a stale load.
//     The same machine code (hand-assembled, tools/asm_variant.py) fails in 8-16 of 24 runs when the
//     kernel descriptor allocates 224 VGPRs (28 granules: what the compiler had chosen, two wavefronts
//     per SIMD) and in 0 of 24 with 225-256, the instructions untouched; s_nop / s_waitcnt padding
//     anywhere in the failing binary changes nothing.  The moment is known too: in every failing
//     wavefront the SIMD's other wavefront terminated inside the failing one's phase B (time stamps,
//     -DTBA_TB_TIMES) -- that is what "only second residents" was.  Idle and read-verify probes of a 224-register
//     allocation (tools/vgpr_probe) see no register change, so the trigger needs this kernel's
//     activity and is not understood further; the kernels here keep away from that allocation
//     (TBP_NOT_224_VGPRS, and tests/test_kernel_resources.py holds every kernel of the library to it).
#pragma once
#include "k_dp.h"

#define TBP_NONE ((i64)0x3fffffffffffffffll)
#define TBP_WAVE_BELOW 1024 // batches up to this many reads: a wavefront per read (latency form)
#define TBP_MIN_CHUNK 64    // rows; the overlap (phase B) is a block of 16 or two

// (tbp_fence: lanes of ONE wavefront hand read_tb entries to each other through memory -- wave_mem_fence,
// tba_common.h, and why __threadfence_block() is not the fence for that)
#define tbp_fence wave_mem_fence

// A kernel of this file must not be allocated exactly 224 VGPRs (see above): naming v231 as clobbered
// makes the compiler report 232.  -DTBA_TB_ALLOC224 takes it away (the build the determinism test is
// shown failing on, profiles/r06_traceback_determinism_alloc224.txt).
#ifdef TBA_TB_ALLOC224
#define TBP_NOT_224_VGPRS() ((void)0)
#else
#define TBP_NOT_224_VGPRS() asm volatile("" ::: "v231")
#endif
// Experiment builds of the round-6 hunt (tools/tb_hunt.py): -DTBA_TB_B2 diverts phase B's stores into a
// second array behind read_tb and records the state every lane enters phase B with in a third
// (read_tb is allocated three arrays long), so that what phase B computed can be compared from run to
// run without phase A's values in the way; -DTBA_TB_WAVES1 pads the workgroup with LDS until only one
// wavefront fits on a SIMD (never failed).  The results of such a build are NOT the traceback.
#ifdef TBA_TB_B2
__device__ i64 TBA_TB_B2_OFF;
#define TBP_B_STORE(tb_, i_, v_) ((tb_)[(i_) + TBA_TB_B2_OFF] = (v_))
#else
#define TBP_B_STORE(tb_, i_, v_) ((tb_)[(i_)] = (v_))
#endif
enum { TBP_A = 0, TBP_B = 1, TBP_V = 2 };

// One block of TBR rows of one walker: rows r0, r0 - 1, ... > stop, with the reference's rules
// (negative band positions wrap like Python indices).  The fetch / bit-mask / clz scheme is
// k_main_tb's (k_dp.h).
//   TBP_A (phase A): every row writes tb; a band-edge violation is recorded (viol_lo = its
//     row; rows descend, so the last one recorded is the lowest) and the walk goes on; leaving the
//     band ends it (rc).
//   TBP_B (phase B): before a row's value is written it is compared with what tb holds
//     (rows >= cmp_lo only: below that the lane underneath wrote nothing); equal -> merged_row = that
//     row, the walk ends; a violation ends it as well (rc), as in the serial walk.
//   TBP_V (the verifier): nothing is written; every row is compared with what tb holds and the
//     rows that differ are counted in *dbg_stores (a band-edge violation is not its business: phase B
//     has seen the same row).
//   IDENT: the band of row r starts at event r - 1 (the static band of start discovery, resquiggle.py:710-716):
//     no band-start array to read.
template <int MODE, bool IDENT = false>
__device__ __forceinline__ void tbp_block(const unsigned char *mv, int rowb, int roww, const i64 *st,
    int Wi, int thresh, i64 r0, i64 stop, i64 &cur_ev, int &bp_guess, int &rc, i64 *tb,
    i64 &viol_lo, i64 cmp_lo, i64 &merged_row, const unsigned char *strip, int strip_s0, i64 n_static,
    int *dbg_stores = nullptr)
{
    constexpr bool EXT = MODE != TBP_A, VER = MODE == TBP_V;
    i64 stv[TBR];
    uint4 win[TBR];
    i64 oldv[TBR];
    int wb = (bp_guess >> 4) - 2;                   // first dword of the window
    wb = wb < 0 ? 0 : (wb > roww - 4 ? roww - 4 : wb);
    // The centre strip (k_dp.h) holds cells [strip_s0, strip_s0 + 64) of every adaptive row, 16 bytes
    // per row: while the walk is in its middle -- it nearly always is -- the block's windows come from
    // there, eight rows to a cache line instead of one.  Same bits as the window of the full row at
    // dword strip_s0 / 16, so everything below is unchanged; a position outside takes the slow path of
    // its row (full row) and the next block looks again.
    const bool use_strip = strip_s0 >= 0 && r0 - (TBR - 1) > n_static && bp_guess >= strip_s0 + 24 &&
                           bp_guess < strip_s0 + 60;
    if (use_strip) wb = strip_s0 >> 4;
    // band starts (and, phase B / verifier, the recorded path) of two rows per 16-byte access: the lanes of a wavefront
    // walk 64 different parts of the reads, so every memory instruction is 64 separate requests whatever its width --
    // the kernel is bound by their number (48 per block of 16 rows with one row per access)
#pragma unroll
    for (int k = 0; k < TBR; k += 2) {
        const i64 rb = r0 - k - 1;                      // the lower row of the pair (rows r0 - k and r0 - k - 1)
        const i64 rbc = rb >= 1 ? rb : 1;
        union { f64x2_u d; i64 q[2]; } u;
        if (IDENT) { u.q[0] = rbc - 1; u.q[1] = rbc; }
        else u.d = *(const f64x2_u *)(st + rbc - 1);    // (st[rb - 1], st[rb]) = the entries of rows rb and rb + 1
        stv[k + 1] = u.q[0];
        stv[k] = rb >= 1 ? u.q[1] : u.q[0];             // (below row 1 both clamp to row 1's entry, as one row per access did)
        if (EXT) {
            u.d = *(const f64x2_u *)(tb + rbc - 1);
            oldv[k + 1] = u.q[0];
            oldv[k] = rb >= 1 ? u.q[1] : u.q[0];
        }
    }
#pragma unroll
    for (int k = 0; k < TBR; k++) {
        const i64 rr = r0 - k;
        const i64 rc_ = rr >= 1 ? rr : 1;
        win[k] = use_strip ? *(const uint4 *)(strip + rc_ * MV_STRIP_BYTES) : *(const uint4 *)(mv + rc_ * rowb + 4 * wb);
    }
    u64 nzl[TBR], nzh[TBR], d2l[TBR], d2h[TBR];
    int fl_full[TBR], m2_full[TBR];
#pragma unroll
    for (int k = 0; k < TBR; k++) {
        const u64 lo = ((u64)win[k].y << 32) | win[k].x, hi = ((u64)win[k].w << 32) | win[k].z;
        const u64 E = 0x5555555555555555ull;
        nzl[k] = (lo | (lo >> 1)) & E; nzh[k] = (hi | (hi >> 1)) & E;
        d2l[k] = (lo >> 1) & ~lo & E;  d2h[k] = (hi >> 1) & ~hi & E;
        const int cf = __clzll((long long)(nzl[k] | 1ull));
        fl_full[k] = nzl[k] ? 31 - (cf >> 1) : -1;
        m2_full[k] = (int)((d2l[k] >> (2 * (fl_full[k] & 31))) & 1ull);
    }
    static_assert(TBR % 2 == 0, "rows are loaded and stored in pairs");
    bool pend_a = false;                            // (phase A: the even row's value, waiting for its pair)
    i64 pend_v = 0;
#pragma unroll
    for (int k = 0; k < TBR; k++) {
        const i64 rr = r0 - k;
        bool sa = false;                            // phase A: this row has a value to record (sv), stored with its pair row
        i64 sv = 0;
        {
        const bool act = rr > stop && rr >= 1 && rc == TBA_OK && (!EXT || VER || merged_row == TBP_NONE);
        const i64 bp64 = cur_ev - stv[k];
        int bp = (int)bp64;
        const int lc = bp - 16 * wb;                // position inside the window
        const bool in_win = bp64 < Wi && bp64 >= 0 && (unsigned)lc < 64u;
        const bool up = (lc & 32) != 0;
        const int amt = 62 - 2 * (lc & 31);
        const u64 sn = (up ? nzh[k] : nzl[k]) << amt, s2 = (up ? d2h[k] : d2l[k]) << amt;
        const bool hit = sn != 0;
        const int c = __clzll((long long)(sn | 1ull)); // 1 + 2 (lc - f); 63 without a hit
        const bool low = !hit && up && fl_full[k] >= 0;
        const int m2_hit = (int)((s2 >> (63 - c)) & 1ull);
        const int f = hit ? lc - (c >> 1) : fl_full[k];
        const int m2 = hit ? m2_hit : m2_full[k];
        const bool fast = in_win && (hit || low);
        int m = m2 ? 2 : 1;
        if (fast) bp = 16 * wb + f;
        if (__builtin_expect(act && !fast, 0)) {
            // outside the window, or nothing but stays down to its start
            if (bp64 >= Wi || bp64 < -Wi) { rc = TBA_INTERNAL; goto row_end; }
            if (in_win) bp = wb > 0 ? 16 * wb - 1 : -1; // everything in the window was a stay
            const unsigned char *row = mv + rr * rowb;
            // the highest non-stay cell at or below bp, 32 cells (one aligned 8-byte load) at a time
            m = 0;
            while (bp >= 0) {
                const u64 wd = *(const u64 *)(row + 8 * (bp >> 5));
                const int top = bp & 31;
                const u64 ms = top == 31 ? wd : (wd & ((1ull << (2 * top + 2)) - 1ull));
                const u64 nz = (ms | (ms >> 1)) & 0x5555555555555555ull;
                if (nz) {
                    const int ff = (63 - __clzll((long long)nz)) >> 1;
                    bp = 32 * (bp >> 5) + ff;
                    m = (int)((wd >> (2 * ff)) & 3ull);
                    break;
                }
                bp = 32 * (bp >> 5) - 1;
            }
            if (bp < 0) {
                // the reference keeps walking through Python's wrap-around of a negative index
#define MVG(b_) ({ int bb_ = (b_) < 0 ? (b_) + Wi : (b_); (int)((row[bb_ >> 2] >> (2 * (bb_ & 3))) & 3); })
                m = MVG(bp);
                while (m == 0) {
                    bp--;
                    if (bp < -Wi) { rc = TBA_INTERNAL; break; }
                    m = MVG(bp);
                }
#undef MVG
                if (rc != TBA_OK) goto row_end;
            }
        }
        if (m == 2) bp--;
        const int edge = bp < Wi - bp - 1 ? bp : Wi - bp - 1;
        const bool beyond = thresh >= 0 && edge < thresh;
        if (act) {
            if (EXT && !VER && beyond) rc = TBA_BEYOND_BANDWIDTH;
            else {
                if (!EXT && beyond) viol_lo = rr;
                cur_ev = stv[k] + bp;
                bp_guess = bp;
                if (VER) { if (oldv[k] != cur_ev + 1) ++*dbg_stores; }
                else if (EXT && rr - 1 >= cmp_lo && oldv[k] == cur_ev + 1) merged_row = rr - 1;
                else {
                    if (EXT) TBP_B_STORE(tb, rr - 1, cur_ev + 1); else { sa = true; sv = cur_ev + 1; }
                    if (EXT && dbg_stores) ++*dbg_stores;
                }
            }
        }
        }
row_end:
        if (MODE == TBP_A) {
            // rows r0 - k (even k) and r0 - k - 1 are neighbours in read_tb: one 16-byte store for the pair
            if ((k & 1) == 0) { pend_a = sa; pend_v = sv; }
            else if (pend_a && sa) {
                union { f64x2_u d; i64 q[2]; } u;
                u.q[0] = sv; u.q[1] = pend_v;
                *(f64x2_u *)(tb + rr - 1) = u.d;
            } else {
                if (pend_a) tb[rr] = pend_v;
                if (sa) tb[rr - 1] = sv;
            }
        }
    }
}

// _trim_traceback (resquiggle.py:754-764) and the first base's change point, as k_main_tb (one lane)
__device__ __forceinline__ void tbp_trim(ReadState &r, i64 *tb, i64 B)
{
    const i64 n_ev = r.n_ev - r.clip;
    volatile i64 *vtb = tb;
    {
        i64 i = 0;
        while (vtb[i] < 0) { vtb[i] = 0; i++; if (i > B) { r.status = TBA_INTERNAL; return; } }
        i64 j = 1;
        while (vtb[B + 1 - j] > n_ev) { vtb[B + 1 - j] = n_ev; j++; if (j > B + 1) { r.status = TBA_INTERNAL; return; } }
    }
    i64 t0 = vtb[0];
    if (t0 < 0) t0 += n_ev + 1;
    r.top_pos = t0;
}

// LPR lanes per read (a power of two <= 64), 64 / LPR reads per wavefront.  idx != nullptr: the
// reads are idx[0 .. n_reads) (the long reads: one wavefront each).  Reads this kernel finishes
// are marked (ReadState.tb_done) and skipped by k_main_tb / k_main_tb_long.
template <int LPR>
__global__ __launch_bounds__(64) void k_main_tb_par(ReadState *rs, i64 n_reads, const i32 *idx,
    const DevParams *dp, const unsigned char *moves, const i64 *band_starts, i64 *read_tb)
{
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x, g = lane / LPR, c = lane % LPR, gbase = g * LPR;
    TBP_NOT_224_VGPRS();
#ifdef TBA_TB_WAVES1
    __shared__ volatile int occ_pad[40 * 256];      // (40 KB per one-wavefront workgroup: four wavefronts on a CU)
    if (n_reads < 0) occ_pad[lane] = lane;
#endif
#ifdef TBA_TB_TIMES
    // (experiment: when and where the wavefront ran -- start / end on the 100 MHz counter and HW_ID into the read's dbg[],
    // to see which wavefronts shared a SIMD with one that failed; profiles/r06_traceback_rootcause.txt)
    const i64 tt0 = (i64)__builtin_amdgcn_s_memrealtime();
#endif
    const i64 slot = (i64)blockIdx.x * RPW + g;
    const bool have = slot < n_reads;
    const i64 ri = have ? (idx ? (i64)idx[slot] : slot) : 0;
    ReadState &r = rs[ri];
    // every lane of the group takes the same decision here (same read)
    const bool on = have && r.status == TBA_OK && r.path == PATH_ADAPTIVE && r.tb_done == 0 &&
                    (idx != nullptr || !r.is_long) && r.B >= 2 && cpl_class(r.W) != 0;
    const i64 B = on ? r.B : 2;
    const int Wi = on ? (int)r.W : 64;
    const int rowb = (int)mv_row_bytes(Wi), roww = rowb / 4;
    const unsigned char *mv = moves + (on ? r.moves_off : 0);
    const i64 *st = band_starts + (on ? r.ref_off : 0);
    i64 *tb = read_tb + (on ? r.seg_off : 0);
    const int thresh = (int)dp->p.band_bound_thresh;
    const int strip_s0 = on ? r.strip_s0 : -1;
    const unsigned char *strip = mv + (B + 1) * (i64)rowb;   // (behind the move rows; valid when strip_s0 >= 0)
    const i64 n_stat = on ? r.n_static : 0;

    // chunk c: rows (lo, hi], hi = B - c L; fewer chunks than lanes for short reads.  The rows of
    // the static bands at the start of the read (masked start, resquiggle.py:607-683: the path is
    // anywhere in those bands, not near their middle) all belong to the lowest chunk.
    i64 top_rows = B - ((on ? r.n_static : 0) + 16);
    top_rows = top_rows < 1 ? 1 : top_rows;
    i64 L = (top_rows + LPR - 1) / LPR;
    L = L < TBP_MIN_CHUNK ? TBP_MIN_CHUNK : L;
    const int n_chunks = (int)((top_rows + L - 1) / L);
    const i64 hi = B - (i64)c * L, lo = c >= n_chunks - 1 || hi - L < 0 ? 0 : hi - L;
    const bool mine = on && c < n_chunks;
    // ---- phase A
    i64 cur = 0, viol_lo = TBP_NONE, none = TBP_NONE;
    int guess = Wi / 2, rcA = TBA_OK;
    if (mine) {
        if (c == 0) { guess = (int)r.top_pos; cur = r.top_pos + st[B - 1]; tb[B] = cur + 1; }
        else cur = st[hi - 1] + Wi / 2;
    }
    const i64 start_ev = cur;                       // my state entering row hi
    {
        i64 r0 = hi;
        bool walking = mine;
        while (__any(walking)) {
            if (walking) {
                tbp_block<TBP_A>(mv, rowb, roww, st, Wi, thresh, r0, lo, cur, guess, rcA, tb, viol_lo, 0, none, strip, strip_s0, n_stat);
                r0 -= TBR;
                if (rcA != TBA_OK || r0 <= lo) walking = false;
            }
        }
    }
    // the lowest tb index this lane wrote a value into: rows are written top-down and a walk that
    // dies (rcA) stops writing, so without a death it is lo; after one, nothing below is known --
    // the lane above then compares nothing in this chunk (cmp_lo above every row of it)
    const i64 wrote_lo = rcA == TBA_OK ? lo : hi;
    tbp_fence(); // phase B reads what the lane below wrote
    // ---- phase B: into chunk c + 1
    const bool ext = mine && c + 1 < n_chunks && rcA == TBA_OK;
    const i64 nxt_start = shfl_i64(start_ev, (lane + 1) & 63), nxt_wrote_lo = shfl_i64(wrote_lo, (lane + 1) & 63);
    const i64 lo2 = c + 1 >= n_chunks - 1 || lo - L < 0 ? 0 : lo - L; // lo of chunk c + 1
    i64 merged_row = TBP_NONE;
    int rcB = TBA_OK;
#ifdef TBA_TB_TIMES
    const i64 ttB0 = (i64)__builtin_amdgcn_s_memrealtime();
#endif
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 12
    int dbg_st = 0;
    int *dbg_stp = &dbg_st;
    const bool dbg_imm = ext && cur == nxt_start;
#else
    int *dbg_stp = nullptr;
#endif
    {
        bool walking = ext;
        if (ext && cur == nxt_start) { merged_row = lo; walking = false; } // (entering row lo = its hi)
        i64 r0 = lo;
#ifdef TBA_TB_B2
        if (ext) tb[lo + 2 * TBA_TB_B2_OFF] = cur | ((i64)guess << 40);   // (third array: the state phase B starts from)
#endif
        while (__any(walking)) {
            if (walking) {
                tbp_block<TBP_B>(mv, rowb, roww, st, Wi, thresh, r0, lo2, cur, guess, rcB, tb, viol_lo, nxt_wrote_lo, merged_row, strip, strip_s0, n_stat, dbg_stp);
                r0 -= TBR;
                if (rcB != TBA_OK || merged_row != TBP_NONE || r0 <= lo2) walking = false;
            }
        }
    }
#ifdef TBA_TB_TIMES
    const i64 ttB1 = (i64)__builtin_amdgcn_s_memrealtime();
#endif
#ifdef TBA_TB_INJECT
    // (test build, libtombo_amd_inject.so: the round-5 fault made deterministic -- the first row under the second
    // chunk top of every TBA_TB_INJECT-th read comes out one event too high; tests/test_gpu_determinism.py
    // expects the verifier to catch exactly those reads and the results to be the oracle's all the same)
    if (ext && c == 1 && lo >= 1 && ri % TBA_TB_INJECT == 3) tb[lo - 1] += 1;
#endif
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 12
    // per read: 0 lanes that extended, 1 merged at once (state equal on entry), 2 merged later, 3 rows
    // overwritten in phase B, 4 chunks, 5 sum of the rows merged at relative to the chunk top, 6 lanes
    // whose phase A died, 7 sum of phase-A start states
    if (on) {
        auto add = [&](int k, i64 v) { atomicAdd((unsigned long long *)&r.dbg[k], (unsigned long long)v); };
        if (ext) add(0, 1);
        if (dbg_imm) add(1, 1);
        if (ext && !dbg_imm && merged_row != TBP_NONE) { add(2, 1); add(5, lo - merged_row); }
        add(3, dbg_st);
        if (c == 0) add(4, n_chunks);
        if (mine && rcA != TBA_OK) add(6, 1);
        if (mine) add(7, start_ev);
    }
#endif
    // ---- the chain, top down (uniform over the group: every lane runs the same loop)
    int status = TBA_OK;
    bool broken = false, from_b = false;            // from_b: the status is a phase B's
    i64 true_from = B + 1;                          // chunk 0: all of phase A is the true walk
    for (int j = 0; j < LPR; j++) {
        const i64 vj = shfl_i64(viol_lo, gbase + j), mj = shfl_i64(merged_row, gbase + j);
        const int aj = __shfl(rcA, gbase + j, 64), bj = __shfl(rcB, gbase + j, 64);
        if (j >= n_chunks || status != TBA_OK || broken) continue;
        if (vj != TBP_NONE && vj <= true_from) { status = TBA_BEYOND_BANDWIDTH; continue; }
        if (aj != TBA_OK) { status = aj; continue; }
        if (j == n_chunks - 1) continue;            // walked down to row 1: done
        if (bj != TBA_OK) { status = bj; from_b = true; continue; }
        if (mj == TBP_NONE) { broken = true; continue; }
        true_from = mj;
    }
    // What rests on a phase B is not final here: an error of one, or a chain without agreement, leaves
    // the read to the serial kernels (tb_done stays 0, top_pos untouched); a finished chain goes to
    // k_tb_par_verify (tb_done = 2), which trims.  An error of a phase A on the true path is the
    // serial walk's own error at that row.
#ifdef TBA_TB_DRAIN
    // (experiment: no load of this wavefront is still in flight when it terminates -- the rows a phase B prefetched and
    // never looked at are otherwise pending at s_endpgm)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
#ifdef TBA_TB_TIMES
    if (have && c == 0) {
        r.dbg[0] = tt0; r.dbg[1] = (i64)__builtin_amdgcn_s_memrealtime();
        r.dbg[2] = (i64)(u32)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |   // HW_ID
                   ((i64)(u32)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); // XCC_ID
        r.dbg[3] = (i64)blockIdx.x; r.dbg[4] = ttB0; r.dbg[5] = ttB1;
    }
#endif
    if (!on || c != 0 || broken || (status != TBA_OK && from_b)) return;
    r.tb_form = LPR; // TBA_TB_FORM_PAR16 / TBA_TB_FORM_PAR64
    if (status != TBA_OK) { r.tb_done = 1; r.status = status; return; }
#ifndef TBA_NO_TB_VERIFY
    r.tb_done = 2;
#else
    tbp_fence(); // the lanes' read_tb entries, before lane 0 of the group reads them back
    r.tb_done = 1;
    tbp_trim(r, tb, B);
#endif
}

// start discovery epilogue: traceback, score_valid_bases (tombo_stats.py:2340-2362), events per
// base (resquiggle.py:740-752) and the retry / fallback decision (resquiggle.py:992-1006).
// One thread per read.
__global__ __launch_bounds__(64) void k_start_tb(ReadState *rs, i64 n_reads, const DevParams *dp, int mode,
    const double *event_means, const double *ref_means, const double *ref_sds,
    const unsigned char *moves, i64 start_moves_stride, i64 *read_tb, double *start_vals)
{
    i64 ri = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (ri >= n_reads) return;
    ReadState &r = rs[ri];
    if (r.status != TBA_OK) return;
    if (r.start_state != (mode == DP_START_TRY ? ST_TRY : ST_RETRY)) return;
    const tba_params &P = dp->p;
    const i64 nb = P.start_n_bases;
    const i64 bw = mode == DP_START_TRY ? P.start_bw : P.start_save_bw;
    i64 *tb = read_tb + r.seg_off;
    // c_banded_traceback over the start band (identity band starts, no band-edge test) in blocks of TBR rows: their
    // 64-cell windows fetched together (tbp_block; a cell-by-cell walk is a dependent memory round trip per cell --
    // 0.5 ms per launch for 250 rows, whatever the batch)
    int rc = TBA_OK;
    {
        const unsigned char *mv = moves + ri * start_moves_stride;
        const int rowb = (int)mv_row_bytes(bw), roww = rowb / 4;
        i64 cur = r.top_pos + (nb - 1), viol = TBP_NONE, none = TBP_NONE;
        int guess = (int)r.top_pos;
        tb[nb] = cur + 1;
        for (i64 r0 = nb; r0 >= 1 && rc == TBA_OK; r0 -= TBR)
            tbp_block<TBP_A, true>(mv, rowb, roww, nullptr, (int)bw, -1, r0, 0, cur, guess, rc, tb, viol, 0, none, nullptr, -1, 0);
    }
    const double *ev = event_means + r.ev_off;
    if (rc == TBA_OK && mode == DP_START_TRY && dp->o.check_start_score) {
        const double *mu = ref_means + r.ref_off, *sd = ref_sds + r.ref_off;
        double *vals = start_vals + ri * nb;
        i64 nv = 0;
        for (i64 i = 0; i < nb; i++) {
            if (tb[i] == tb[i + 1]) continue;
            double m = np_sum(ev + tb[i], tb[i + 1] - tb[i]) / (double)(tb[i + 1] - tb[i]);
            vals[nv++] = fabs((m - mu[i]) / sd[i]);
        }
        if (nv == 0) rc = TBA_INVALID_START_PATH;
        else if (np_sum(vals, nv) / (double)nv > dp->o.sig_match_thresh) rc = TBA_POOR_START;
    }
    if (rc == TBA_OK) {
        r.epb = (double)(tb[nb] - tb[0]) / (double)(nb + 1);
        r.mapped_start = tb[0];
        r.start_state = ST_OK;
        r.start_res[2 * mode] = (double)tb[0];
        r.start_res[2 * mode + 1] = r.epb;
        r.n_start_calls = mode + 1;
    } else if (mode == DP_START_TRY && rc != TBA_INTERNAL) {
        // except th.TomboError: retry with the save bandwidth or fall back to the static DP
        r.pad0 = rc; // why the first try failed (stand-alone find_seq_start_in_events reports it)
        r.start_state = r.n_ev < P.start_save_bw + nb ? ST_STATIC : ST_RETRY;
    } else {
        r.status = rc;
    }
}

// The verifier, behind the kernel boundary (what the result rests on: top of this file).  Same lanes,
// same chunk geometry as k_main_tb_par<LPR>, over the reads it finished (tb_done == 2): the lane of
// boundary c takes the state entering the top row of chunk c + 1 from the entry above it -- the bottom
// of chunk c, true by the chain -- walks TBR rows down without writing and counts the rows where
// read_tb holds something else.  Phase B merged within two rows at nearly every boundary and its
// faults of round 5 sat in its first row, so this covers what phase B wrote; rows it walked beyond
// the first block (one boundary in ~300) rest on phase B alone.  A read with any disagreement goes
// back to the serial kernels (tb_done = 0, its count in tb_verify_fail); the others are trimmed here.
template <int LPR>
__global__ __launch_bounds__(64) void k_tb_par_verify(ReadState *rs, i64 n_reads, const i32 *idx,
    const DevParams *dp, const unsigned char *moves, const i64 *band_starts, i64 *read_tb)
{
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x, g = lane / LPR, c = lane % LPR, gbase = g * LPR;
    TBP_NOT_224_VGPRS();
    const i64 slot = (i64)blockIdx.x * RPW + g;
    const bool have = slot < n_reads;
    const i64 ri = have ? (idx ? (i64)idx[slot] : slot) : 0;
    ReadState &r = rs[ri];
    const bool on = have && r.status == TBA_OK && r.path == PATH_ADAPTIVE && r.tb_done == 2 && r.tb_form == LPR &&
                    (idx != nullptr || !r.is_long);
    const i64 B = on ? r.B : 2;
    const int Wi = on ? (int)r.W : 64;
    const int rowb = (int)mv_row_bytes(Wi), roww = rowb / 4;
    const unsigned char *mv = moves + (on ? r.moves_off : 0);
    const i64 *st = band_starts + (on ? r.ref_off : 0);
    i64 *tb = read_tb + (on ? r.seg_off : 0);
    const int strip_s0 = on ? r.strip_s0 : -1;
    const unsigned char *strip = mv + (B + 1) * (i64)rowb;
    const i64 n_stat = on ? r.n_static : 0;
    i64 top_rows = B - ((on ? r.n_static : 0) + 16);   // (the chunk geometry of k_main_tb_par)
    top_rows = top_rows < 1 ? 1 : top_rows;
    i64 L = (top_rows + LPR - 1) / LPR;
    L = L < TBP_MIN_CHUNK ? TBP_MIN_CHUNK : L;
    const int n_chunks = (int)((top_rows + L - 1) / L);
    const i64 hi = B - (i64)c * L, lo = c >= n_chunks - 1 || hi - L < 0 ? 0 : hi - L;
    const i64 lo2 = c + 1 >= n_chunks - 1 || lo - L < 0 ? 0 : lo - L;
    const bool ext = on && c + 1 < n_chunks && lo >= 1;
    i64 cur = ext ? tb[lo] - 1 : 0;                 // recorded after row lo + 1: the state entering row lo, + 1
    int guess = Wi / 2, rc = TBA_OK, n_diff = 0;
    if (ext) {
        const i64 g0 = cur - st[lo - 1];
        guess = g0 < 0 ? 0 : (g0 >= Wi ? Wi - 1 : (int)g0);
    }
    i64 none = TBP_NONE, viol = TBP_NONE;
    if (ext) tbp_block<TBP_V>(mv, rowb, roww, st, Wi, -1, lo, lo2, cur, guess, rc, tb, viol, 0, none, strip, strip_s0, n_stat, &n_diff);
    if (ext && rc != TBA_OK) n_diff++;              // (a walk that died where phase B's did not)
    int total = 0;
    for (int j = 0; j < LPR; j++) total += __shfl(n_diff, gbase + j, 64);
    if (!on || c != 0) return;
    if (total != 0) {                               // the serial kernels walk this read from top_pos, untouched so far
        r.tb_verify_fail = total;
        r.tb_done = 0;
        r.tb_form = TBA_TB_FORM_NONE;
        return;
    }
    r.tb_done = 1;
    tbp_trim(r, tb, B);
}
