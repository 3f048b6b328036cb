// k_dp_wgm.h -- the MAIN forward pass (c_adaptive_banded_forward_pass, _c_dynamic_programming.pyx:
// 314-412; static rows: c_banded_forward_pass, pyx:240-279) with one WORKGROUP per read: the
// "latency form" VERDICT r3 #4 asked for.  Bit for bit k_dp's results (tests/test_gpu_dp_workgroup.py)
// -- and SLOWER than k_dp, so it is compiled in but switched off (TBA_WG_BATCH, k_dp.h).
//
// The idea: k_dp gives a read one wavefront -- 8 cells per lane at W = 500, 1.29 us per row whatever
// else the machine does, 12.9 ms for a 10 kb read (a batch of one: resquiggle_read), 240 ms for a
// 200 kb one.  Spread the band over the 256 lanes of a workgroup (4 wavefronts on the 4 SIMDs of a
// CU), two cells per lane:
//   events   : LDS ring as in k_dp, refilled 256 events at a time one row ahead
//   prev row : LDS, -inf around the band: a row's candidates are plain indexed reads at the band offset
//   levels   : mu, sd, 1 / sd of 256 rows staged in LDS
//   stay chain: k_dp's exact fixed point of the exit values (the least fixed point of a monotone
//              system, whatever the order of the updates): inside a wavefront the values move by
//              DPP, four hops per convergence test, no barrier; across wavefronts a round publishes
//              the exit of lane 63 and whether it moved -- two rounds (barriers) per row
//   arg-max  : per wavefront as in k_dp, the four (max, index) pairs combined by every lane after the
//              barrier that ends the row
// Arithmetic, tie rules, move codes and the packed move layout (linear in the band cell, the row
// stride of the band's k_dp class) are k_dp's, so the traceback kernels do not know which kernel
// wrote the rows.
//
// What was measured (MI355X, one 10 kb read, W = 500; tools/wgm_probe.py with -DTBA_WGM_STATS,
// tools/latency_stages.py; profiles/r04_dp_workgroup_form.txt):
//   __syncthreads barriers, one per sweep              46.9 ms  (4.7 us per row)
//   + LDS-only barriers (s_waitcnt lgkmcnt(0); s_barrier)  49.0  (the barriers were not waiting for memory)
//   + wavefront-local sweeps, 2 barriers per row        31.3
//   + static / adaptive rows split at compile time      29.9
//   + four hops per convergence test                    20.3     k_dp: 12.9
// Why it loses: a row is one dependent chain -- arg-max of the previous row -> band start -> events
// -> z (a division) -> candidates -> stay chain -> cells -> arg-max -- of ~300 instructions, and a
// lone wavefront issues a DEPENDENT float64 instruction every 10-20 cycles whatever its width.
// Splitting the band shrinks the per-cell work, which was not on that chain (k_dp's 8 cells per lane
// are independent and fill the issue slots between the chain's steps), and adds to it: per row 965
// cycles up to the sweeps, 912 from the cells to the arg-max, and in between the stay chains of the
// band's right half (the cells ahead of the path are reached by staying), which are ~30 cells long
// -- 4 hops at 8 cells per lane, 15-20 at two (wavefronts 2 and 3: 19.8 hops per row against 8.4 for
// 0 and 1), ~140 cycles a hop, while the other wavefronts wait at the round's barrier.
#pragma once
#include "k_dp.h"

// Workgroup barrier that waits for the LDS traffic only (__syncthreads() is a full fence: the
// compiler puts s_waitcnt vmcnt(0) in front of the s_barrier, and a row has global accesses in flight)
#define WGM_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define WGM_RING 2048
#define WGM_LVROWS 256

__global__ __launch_bounds__(WGM_NT) void k_dp_wgm(ReadState *rs, const i32 *list, const DevParams *dp,
    const double *event_means, const double *ref_means, const double *ref_sds,
    i64 *band_starts, const i32 *lo_arr, const i32 *hi_arr, unsigned char *moves, double *last_row)
{
    constexpr int NT = WGM_NT, CPL = WGM_CPL, RING = WGM_RING;
    constexpr int PROW = 8 + WGM_MAXW + 264;
    __shared__ double ring[RING + CPL];        // event means around the band (+ mirror of the first slots)
    __shared__ double PR[PROW];                // previous row: 8 x -inf, cells, -inf behind
    __shared__ double LV[3 * WGM_LVROWS];      // level, sd, 1 / sd of 256 rows
    __shared__ double xch[2][4];               // exit value of each wavefront's last lane, per sweep parity
    __shared__ int chg[2][4];                  // "this wavefront saw an incoming value change" of the sweep before
    __shared__ double s_max[2][4];             // per row parity: wavefront maxima ...
    __shared__ int s_idx[2][4];                // ... and the band cell of each
    const i64 ri = list ? list[blockIdx.x] : blockIdx.x;
    ReadState &r = rs[ri];
    if (r.status != TBA_OK || r.path == PATH_NONE || !dp_by_workgroup(dp, r)) return;
    const tba_params &P = dp->p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = (int)r.W, n_rows = (int)r.B, n_static = (int)r.n_static;
    const int n_ev = (int)(r.n_ev - r.clip);
    const double *ev = event_means + r.ev_off + r.clip;
    const double *rmu = ref_means + r.ref_off, *rsd = ref_sds + r.ref_off;
    i64 *bst = band_starts + r.ref_off;
    const i32 *lo_a = lo_arr + r.ref_off, *hi_a = hi_arr + r.ref_off;
    unsigned char *mv = moves + r.moves_off;
    const i64 mv_stride = mv_class_rowb(cpl_class(W));
    const double stay_pen = P.stay_pen, skip_pen = P.skip_pen, z_shift = P.z_shift;
    const double zcap = P.do_winsorize_z ? P.max_half_z_score : INFINITY;
    const double fill_masked = dp->fill_masked;
    const double NEG_INF = -INFINITY;
    const int half_bw = W / 2;
    const int b0 = tid * CPL;
    int nvalid = W - b0;
    nvalid = nvalid < 0 ? 0 : (nvalid > CPL ? CPL : nvalid);
    double zs[CPL];
#pragma unroll
    for (int j = 0; j < CPL; j++) zs[j] = j < nvalid ? z_shift : NEG_INF;

    auto ring_store = [&](int a, double x) {
        const int sl = a & (RING - 1);
        ring[sl] = x;
        if (sl < CPL) ring[RING + sl] = x;
    };
    auto ev_load = [&](int a) {
        const int ac = a < 0 ? 0 : (a >= n_ev ? n_ev - 1 : a);
        const double x = ev[ac];
        return (a >= 0 && a < n_ev) ? x : 0.0;
    };
    auto stage_levels = [&](int first) {       // rows first .. first + 255 (caller puts a barrier behind)
        int rc = first + tid;
        rc = rc < n_rows ? rc : n_rows - 1;
        const double sd = rsd[rc];
        LV[3 * tid] = rmu[rc]; LV[3 * tid + 1] = sd; LV[3 * tid + 2] = 1.0 / sd;
    };
    // row 0 of the forward pass: zeros inside the band (pyx:253-254)
    for (int k = tid; k < PROW; k += NT) PR[k] = (k >= 8 && k < 8 + W) ? 0.0 : NEG_INF;
    int st_n = 0, lo_n = 0, hi_n = W;
    auto fetch_row = [&](int rr) {             // band geometry of a static row, one row ahead
        const int rc = rr < n_rows ? rr : n_rows - 1;
        if (rc < n_static) { st_n = (int)bst[rc]; lo_n = lo_a[rc]; hi_n = hi_a[rc]; }
    };
    fetch_row(0);
    int filled = n_static > 0 ? st_n : 0;      // first event not in the ring yet
    for (int c = 0; c < RING / NT - 1; c++) { ring_store(filled + tid, ev_load(filled + tid)); filled += NT; }
    stage_levels(0);
    if (tid < 4) { s_max[1][tid] = tid == 0 ? 0.0 : NEG_INF; s_idx[1][tid] = 0; } // "row -1": arg-max 0
    __syncthreads();

    double pf = 0.0;       // one prefetched chunk (event pf_at + tid), in flight; pf_in: inside the read
    int pf_at = 0;
    bool pf_pending = false, pf_in = false;
    int prev_start = 0;
    double v[CPL];
#pragma unroll
    for (int j = 0; j < CPL; j++) v[j] = j < nvalid ? 0.0 : NEG_INF;
    // One row.  ADAPT (compile time): the row is past the static rows -- that instance never touches
    // st_n / lo_n / hi_n, registers with loads in flight: a row that as much as selects on one of
    // them gets an s_waitcnt vmcnt(0), which on gfx9 also waits for the previous row's STORES (a
    // memory round trip per row: 3.1 us per row measured).  Returns true when the read has failed.
#ifdef TBA_WGM_STATS
    i64 st_it = 0, st_rnd = 0, st_wait = 0, st_cyc[4] = {0, 0, 0, 0};  // wave 0: sweeps, rounds, cycles: to sweeps / sweeps / cells.. / row-end barrier
#endif
    auto row_step = [&](const int row, auto adapt_tag) __attribute__((always_inline)) -> bool {
        constexpr bool adapt = decltype(adapt_tag)::value;
#ifdef TBA_WGM_STATS
        const i64 t_row = (i64)__builtin_readcyclecounter();
#endif
        // arg-max of the previous row (first index among equal maxima: first wavefront holding it)
        int am;
        {
            const int par = (row + 1) & 1;
            int best = 0;
#pragma unroll
            for (int w = 1; w < 4; w++) if (s_max[par][w] > s_max[par][best]) best = w;
            am = s_idx[par][best];
        }
        int cur_start, lo, hi;
        double fill;
        if constexpr (!adapt) {
            cur_start = st_n; lo = lo_n; hi = hi_n;
            fill = fill_masked;
            fetch_row(row + 1);
        } else {
            // adaptive band placement, pyx:342-358
            cur_start = prev_start + am - half_bw + 1;
            if (cur_start < prev_start) cur_start = prev_start;
            if (cur_start >= n_ev) {
                if (row < n_rows - 2) { if (tid == 0) r.status = TBA_ADAPT_BEYOND; return true; }
                cur_start = n_ev - 1;
            }
            if (tid == 0) bst[row] = cur_start;
            lo = 0;
            hi = cur_start + W <= n_ev ? W : n_ev - cur_start;
            fill = MASK_FILL_Z_SCORE;          // literal -15, pyx:385-386
        }
        const int diff = row > 0 ? cur_start - prev_start : 0;
        // (a band jump past the prefetched events: static rows only, the adaptive step is bounded)
        while (cur_start + WGM_MAXW > filled) {
            if (pf_pending) { ring_store(pf_at + tid, pf_in ? pf : 0.0); filled = pf_at + NT; pf_pending = false; }
            else { ring_store(filled + tid, ev_load(filled + tid)); filled += NT; }
            WGM_BARRIER();
        }
        const double mu = LV[3 * (row & (WGM_LVROWS - 1))], sd = LV[3 * (row & (WGM_LVROWS - 1)) + 1],
                     y = LV[3 * (row & (WGM_LVROWS - 1)) + 2];
        // shifted half z-scores (pyx:361-372 / resquiggle.py:574-582,712-720)
        double z[CPL];
        {
            const double *er = ring + ((cur_start + b0) & (RING - 1));
#pragma unroll
            for (int j = 0; j < CPL; j++) {
                double pz = fabs(div_by_recip(er[j] - mu, sd, y));
                pz = __builtin_fmin(pz, zcap);
                z[j] = zs[j] - pz;
            }
            if ((!adapt && lo != 0) || hi != W) {      // masked start rows / band past the last event
#pragma unroll
                for (int j = 0; j < CPL; j++) {
                    z[j] = (b0 + j >= lo && b0 + j < hi) ? z[j] : fill;
                    z[j] = j < nvalid ? z[j] : NEG_INF;
                }
            }
        }
        // diag / skip candidates from the previous row (pyx:220-231, first cell pyx:392-401):
        // A[k] = previous-row cell b0 + diff + k - 1
        double cv[CPL];
        bool tk[CPL];
        {
            const int at = 8 + b0 + diff - 1;
            const double *pa = PR + (at < PROW - CPL - 1 ? at : PROW - CPL - 1);   // (all -inf out there)
            double A[CPL + 1];
#pragma unroll
            for (int k = 0; k <= CPL; k++) A[k] = pa[k];
            const bool fs = diff == 0;
#pragma unroll
            for (int j = 0; j < CPL; j++) {
                const double d = A[j] + z[j];
                const double s = A[j + 1] - skip_pen;
                bool take_s = s > d;
                if (j == 0) {
                    take_s = tid == 0 ? fs : take_s;   // band cell 0: skip xor diag
                    cv[j] = take_s ? s : d;
                } else {
                    cv[j] = max_f64_raw(s, d);
                }
                tk[j] = take_s;
            }
        }
        // stay chain
        double exit0;
        {
            double x = NEG_INF;
#pragma unroll
            for (int j = 0; j < CPL; j++) x = max_f64_raw(cv[j], (x - stay_pen) + z[j]);
            exit0 = x;
        }
        // Fixed point of the exit values (k_dp's sweeps; the least fixed point of a monotone system is
        // reached whatever the order of the updates).  Inside a wavefront the values move by DPP
        // until nothing changes -- no barrier; across wavefronts a round publishes the exit of lane
        // 63 and whether it moved since the last round, and the row is done when no exit moved: two
        // rounds when no stay chain crosses a wavefront boundary.
#ifdef TBA_WGM_STATS
        const i64 t_sw = (i64)__builtin_readcyclecounter();
        st_cyc[0] += t_sw - t_row;
#endif
        double in = NEG_INF, ex = exit0, bin = NEG_INF, last_pub = NEG_INF;
        bool ok = false;
        for (int rnd = 0; rnd < 8 && !ok; rnd++) {
            bool conv = false;
            // (four hops per convergence test: a hop is a dependent chain of ~7 instructions, the test
            // -- compare, ballot, scalar branch -- costs as much again; hops past the fixed point
            // change nothing)
            for (int it = 0; it < 20; it++) {
                bool moved = false;
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    double nin = wave_shr1_f64(ex, NEG_INF);
                    nin = lane == 0 ? bin : nin;
                    moved = nin != in;
                    in = nin;
                    double c = in;
#pragma unroll
                    for (int j = 0; j < CPL; j++) c = (c - stay_pen) + z[j];
                    ex = max_f64_raw(exit0, c);
                }
#ifdef TBA_WGM_STATS
                st_it += 4;
#endif
                if (__ballot(moved) == 0) { conv = true; break; }
            }
            if (!conv) break;                                   // NaNs
#ifdef TBA_WGM_STATS
            st_rnd++;
#endif
            if (lane == 63) { xch[rnd & 1][wave] = ex; chg[rnd & 1][wave] = ex != last_pub ? 1 : 0; last_pub = ex; }
#ifdef TBA_WGM_STATS
            const i64 t_w = (i64)__builtin_readcyclecounter();
#endif
            WGM_BARRIER();
#ifdef TBA_WGM_STATS
            st_wait += (i64)__builtin_readcyclecounter() - t_w;
#endif
            // (wavefront 3's exit feeds nobody)
            if ((chg[rnd & 1][0] | chg[rnd & 1][1] | chg[rnd & 1][2]) == 0) ok = true;
            else if (wave > 0) bin = xch[rnd & 1][wave - 1];
        }
        if (!ok) { if (tid == 0) r.status = TBA_INTERNAL; return true; }   // NaNs in the signal
#ifdef TBA_WGM_STATS
        const i64 t_cl = (i64)__builtin_readcyclecounter();
        st_cyc[1] += t_cl - t_sw;
#endif
        // the cells, their move codes (0 stay, 1 skip, 2 diag), lane-local maximum
        u32 mvw = 0;
        double lmax = NEG_INF;
        {
            double x = in;
#pragma unroll
            for (int j = 0; j < CPL; j++) {
                const double s = (x - stay_pen) + z[j];
                const u32 f = cv[j] > s ? (tk[j] ? 1u : 2u) : 0u;
                mvw |= f << (2 * j);
                x = max_f64_raw(cv[j], s);
                v[j] = x;
                lmax = max_f64_raw(lmax, x);
            }
        }
        {   // two lanes to a byte, the row linear in the band cell
            const u32 up = (u32)__shfl_down((int)mvw, 1, 64);
            const int byte = tid >> 1;
            if (!(lane & 1) && byte < mv_stride) mv[(i64)(row + 1) * mv_stride + byte] = (unsigned char)(mvw | (up << 4));
        }
#pragma unroll
        for (int j = 0; j < CPL; j++) if (j < nvalid) PR[8 + b0 + j] = v[j];
        // event ring upkeep: land the chunk that was in flight, ask for the next one
        if (pf_pending) { ring_store(pf_at + tid, pf_in ? pf : 0.0); filled = pf_at + NT; pf_pending = false; }
        if (filled < cur_start + 2 * WGM_MAXW && filled < n_ev + WGM_MAXW) {
            // (the clamped load only: selecting on the value here would wait for it here)
            const int a = filled + tid;
            pf_at = filled; pf_in = a < n_ev; pf = ev[a < n_ev ? a : n_ev - 1]; pf_pending = true;
        }
        if (((row + 1) & (WGM_LVROWS - 1)) == 0) stage_levels(row + 1);
        // wavefront arg-max, first index among equal maxima (c_argmax, pyx:186-197)
        {
            const double wm = wave_max_f64(lmax);
            const u64 eq = __ballot(lmax == wm && nvalid > 0);
            const int wl = eq ? __ffsll((unsigned long long)eq) - 1 : 0;
            int wj = CPL - 1;
#pragma unroll
            for (int j = CPL - 2; j >= 0; j--) wj = ((__ballot(v[j] == wm) >> wl) & 1ull) ? j : wj;
            if (lane == 0) { s_max[row & 1][wave] = eq ? wm : NEG_INF; s_idx[row & 1][wave] = (wave * 64 + wl) * CPL + wj; }
        }
        prev_start = cur_start;
#ifdef TBA_WGM_STATS
        const i64 t_b = (i64)__builtin_readcyclecounter();
        st_cyc[2] += t_b - t_cl;
#endif
        WGM_BARRIER();
#ifdef TBA_WGM_STATS
        st_cyc[3] += (i64)__builtin_readcyclecounter() - t_b;
#endif
        return false;
    };
    {
        int row = 0;
        const int n_stat_rows = n_static < n_rows ? n_static : n_rows;
        for (; row < n_stat_rows; row++) if (row_step(row, BoolTag<false>{})) return;
        for (; row < n_rows; row++) if (row_step(row, BoolTag<true>{})) return;
    }
    // last row + traceback start (np.argmax of the last row, resquiggle.py:728,1032)
    double *lr = last_row + ri * TBA_MAX_BAND;
#pragma unroll
    for (int j = 0; j < CPL; j++) lr[b0 + j] = v[j];
#ifdef TBA_WGM_STATS
    if (lane == 0) r.dbg[wave] = st_it;
#endif
    if (tid == 0) {
        const int par = (n_rows + 1) & 1;
        int best = 0;
        for (int w = 1; w < 4; w++) if (s_max[par][w] > s_max[par][best]) best = w;
        r.top_pos = s_idx[par][best];
        r.dp_wg = 1;
#ifdef TBA_WGM_STATS
        r.dbg[4] = n_rows; r.dbg[5] = st_rnd; r.dbg[6] = st_cyc[1]; r.dbg[7] = st_wait;
#endif
    }
}
