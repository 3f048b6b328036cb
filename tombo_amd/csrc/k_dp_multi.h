// k_dp_multi.h -- the main adaptive banded forward pass for NARROW bands: several reads per
// wavefront (c_adaptive_banded_forward_pass / c_process_band / c_argmax,
// _c_dynamic_programming.pyx:186-236,314-412, after the masked start rows of
// _get_masked_start_fwd_pass, resquiggle.py:607-683).
//
// k_dp (k_dp.h) gives a read the 64 lanes of a wavefront whatever its band: at Tombo's default
// bandwidth (300) a lane owns 5 cells, at 100 it owns 4 with 156 of 256 lane-cells idle, and a stay
// run of a few dozen cells crosses 5-7 lane boundaries, each one more exact sweep of the whole
// wavefront.  Here a wavefront is cut into RPW groups of LPR = 64 / RPW lanes, one read per group,
// CPL cells per lane (W <= 128: 4 reads x 16 lanes x 8 cells; W <= 256: 2 x 32 x 8; W <= 320:
// 2 x 32 x 10): the lanes are full, a stay run crosses half as many boundaries, and every
// per-row overhead (control, reductions, ballots) is paid once for RPW reads.
//
// Same cell arithmetic, same exact fixed-point sweeps, same packed 2-bit move rows (linear in the
// band position, stride mv_row_bytes(W)) as k_dp: the traceback does not know which kernel wrote a
// read.  What changes is the plumbing: everything that is wave-uniform in k_dp (band start,
// arg-max, row geometry, level of the row) is uniform per GROUP here and lives in vector
// registers; DPP shifts stop at group boundaries; the groups of a wavefront run the same row index
// in lock step (reads are taken in order of length, so the groups of a wavefront finish
// together).  The previous row cannot stay in registers as in k_dp -- its shift by the band offset
// is a compile-time register renaming there, and the groups of a wavefront have different offsets
// (running the per-offset code once per distinct offset under exec masks was tried first: the
// divergent 9-way switch doubled the register count to 410) -- so it goes through LDS: every lane
// writes its cells, reads them back at its group's offset (11 ds_read_b64 at one base address);
// cells past the band and the pad around the row hold -inf for good, which is every band-edge
// guard of pyx:220-231 at once.
#pragma once
#include "k_dp.h"

// lane i <- lane i+1 inside a group of LPR lanes; the last lane of a group takes -inf
template <int LPR>
__device__ __forceinline__ double grp_shl1_f64(double x, bool last_of_group)
{
    const double NI = -INFINITY;
    int lo = __double2loint(x), hi = __double2hiint(x);
    if constexpr (LPR == 16) { // row_shl:1 -- DPP rows are 16 lanes: nothing crosses a group
        lo = __builtin_amdgcn_update_dpp(__double2loint(NI), lo, 0x101, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(__double2hiint(NI), hi, 0x101, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    } else {
        lo = __builtin_amdgcn_update_dpp(__double2loint(NI), lo, 0x130, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(__double2hiint(NI), hi, 0x130, 0xf, 0xf, false);
        const double t = __hiloint2double(hi, lo);
        return LPR == 64 ? t : (last_of_group ? NI : t);
    }
}
// lane i <- lane i-1 inside a group; the first lane of a group takes -inf
template <int LPR>
__device__ __forceinline__ double grp_shr1_f64(double x, bool first_of_group)
{
    const double NI = -INFINITY;
    int lo = __double2loint(x), hi = __double2hiint(x);
    if constexpr (LPR == 16) { // row_shr:1
        lo = __builtin_amdgcn_update_dpp(__double2loint(NI), lo, 0x111, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(__double2hiint(NI), hi, 0x111, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    } else {
        lo = __builtin_amdgcn_update_dpp(__double2loint(NI), lo, 0x138, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(__double2hiint(NI), hi, 0x138, 0xf, 0xf, false);
        const double t = __hiloint2double(hi, lo);
        return LPR == 64 ? t : (first_of_group ? NI : t);
    }
}

// `order`: read indices in order of decreasing length (host, plan_batch): the groups of a
// wavefront hold reads of like length.  A group takes its read only if it is on the adaptive path
// with the batch bandwidth (a short read's whole-read static band has its own width: k_dp).
#ifndef TBA_DPM_WAVES
#define TBA_DPM_WAVES 3 // waves per SIMD the register allocation aims at (LDS allows 11 per CU)
#endif
template <int CPL, int RPW>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TBA_DPM_WAVES))) void k_dp_multi(ReadState *rs, i64 n_reads, const i32 *order,
    const DevParams *dp, const double *event_means, const double *ref_means, const double *ref_sds,
    i64 *band_starts, const i32 *lo_arr, const i32 *hi_arr, unsigned char *moves, double *last_row)
{
    constexpr int LPR = 64 / RPW, CELLS = LPR * CPL;
    constexpr int S = CPL < 8 ? CPL : 8;
    constexpr int RING = CELLS + 4 * LPR <= 256 ? 256 : CELLS + 4 * LPR <= 512 ? 512 : 1024;
    constexpr int PROW = 8 + CELLS + 8 + CPL + 1; // previous row of a group: 8 x -inf, cells, -inf pad
    __shared__ double ring_all[RPW * (RING + CPL)];
    __shared__ double prow_all[RPW * PROW];
    const tba_params &P = dp->p;
    const int lane = threadIdx.x, g = lane / LPR, sl = lane % LPR;
    const bool first_lane = sl == 0, last_lane = sl == LPR - 1;
    const i64 slot = (i64)blockIdx.x * RPW + g;
    const i64 ri = order[slot < n_reads ? slot : n_reads - 1];
    ReadState &r = rs[ri];
    const int Wi = (int)uni(P.bandwidth);
    bool alive = slot < n_reads && r.status == TBA_OK && r.path == PATH_ADAPTIVE && r.W == Wi;
    if (__ballot(alive) == 0) return;
    const double stay_pen = uni(P.stay_pen), skip_pen = uni(P.skip_pen), z_shift = uni(P.z_shift);
    const double zcap = P.do_winsorize_z ? uni(P.max_half_z_score) : INFINITY;
    const double fill_masked = uni(dp->fill_masked);
    const double NEG_INF = -INFINITY;
    const int half_bw = Wi / 2; // integer division, pyx:327
    const i64 mv_stride = mv_row_bytes(Wi);

    // per group (identical in all lanes of the group)
    const int n_rows = alive ? (int)r.B : 0, n_static = (int)r.n_static;
    int n_ev = (int)(r.n_ev - r.clip);
    n_ev = n_ev < 1 ? 1 : n_ev;
    const double *ev = event_means + r.ev_off + r.clip;
    // the per-base arrays of the read share one offset (one register pair instead of five pointers)
    const i64 ro = r.ref_off;
    const double *rmu = ref_means, *rsd = ref_sds;
    i64 *bst = band_starts;
    const i32 *lo_a = lo_arr, *hi_a = hi_arr;
    unsigned char *mv = moves + r.moves_off;
    double *ring = ring_all + g * (RING + CPL);
    int max_rows = n_rows;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(max_rows, o, 64); max_rows = t > max_rows ? t : max_rows; }
    max_rows = uni(max_rows);

    const int b0 = sl * CPL;
    int nvalid = Wi - b0;
    nvalid = nvalid < 0 ? 0 : (nvalid > CPL ? CPL : nvalid);
    const bool mixed = uni((Wi % CPL) != 0);        // a lane with cells on both sides of the band end
    const double zs_lane = nvalid > 0 ? z_shift : NEG_INF; // z_shift of my cells (the mixed lane: per cell)
    double v[CPL];
    int prev_start = 0, am = 0;
    // previous row in LDS: row 0 is zeros (pyx:253-254) inside the band, -inf everywhere else
    double *prow = prow_all + g * PROW;
    for (int k = sl; k < PROW; k += LPR) prow[k] = (k >= 8 && k < 8 + Wi) ? 0.0 : NEG_INF;

    // event ring of the group (k_dp: one per wavefront), chunks of LPR events
    auto ring_store = [&](int a, double x) {
        const int q = (a + RING) & (RING - 1);
        ring[q] = x;
        if (q < CPL) ring[RING + q] = x;
    };
    auto ev_load = [&](int a) {
        const int ac = a < 0 ? 0 : (a >= n_ev ? n_ev - 1 : a);
        const double x = ev[ac];
        return (a >= 0 && a < n_ev) ? x : 0.0;
    };
    int filled = n_static > 0 && n_rows > 0 ? (int)bst[ro] : 0;
    for (int c = 0; c < RING / LPR - 2; c++) { ring_store(filled + sl, ev_load(filled + sl)); filled += LPR; }
    double pf = 0.0;
    int pf_at = 0;
    bool pf_pending = false;
    __syncthreads();

    // level, sd and reciprocal of LPR rows at a time: lane sl of the group holds row blk0 + sl
    double mu_v = 0, sd_v = 1, y_v = 1;
    auto load_levels = [&](int first) {
        int rc = first + sl;
        rc = rc < n_rows ? rc : n_rows - 1;
        rc = rc < 0 ? 0 : rc;
        mu_v = rmu[ro + rc]; sd_v = rsd[ro + rc];
        y_v = 1.0 / sd_v;
        asm volatile("" : "+v"(mu_v)); // (the loads end here, not in every row: k_dp.h)
    };
    load_levels(0);
    int st_n = 0, lo_n = 0, hi_n = Wi;
    // band geometry of the static (masked start) rows, one row ahead.  Rows in which no group is
    // static must not touch the three registers at all (k_dp.h: the compiler guards any use with
    // s_waitcnt vmcnt(0), which also waits for the previous row's stores), hence the wave-uniform
    // branches around the loads here and around the selects below.
    auto fetch_row = [&](int rr) {
        int rc = rr < n_rows ? rr : n_rows - 1;
        rc = rc < 0 ? 0 : rc;
        if (__ballot(rc < n_static) != 0) {
            if (rc < n_static) { st_n = (int)bst[ro + rc]; lo_n = lo_a[ro + rc]; hi_n = hi_a[ro + rc]; }
        }
    };
    fetch_row(0);
    const int grp_base = lane & ~(LPR - 1);

    for (int row = 0; row < max_rows; row++) {
        bool act = alive && row < n_rows;
        const int sel = row & (LPR - 1);
        if (sel == 0 && row != 0) load_levels(row);
        const double mu = shfl_f64(mu_v, grp_base | sel), sd = shfl_f64(sd_v, grp_base | sel),
                     y = shfl_f64(y_v, grp_base | sel);
        const bool is_static = row < n_static;
        // adaptive band placement, pyx:342-358
        int cs = prev_start + am - half_bw + 1;
        cs = cs < prev_start ? prev_start : cs;
        if (cs >= n_ev) {
            if (act && !is_static && row < n_rows - 2) {
                if (first_lane) r.status = TBA_ADAPT_BEYOND;
                alive = false; act = false;
            }
            cs = n_ev - 1;
        }
        int cur_start = cs, lo = 0, hi = cs + Wi <= n_ev ? Wi : n_ev - cs;
        if (__ballot(is_static) != 0) {
            cur_start = is_static ? st_n : cur_start;
            lo = is_static ? lo_n : lo;
            hi = is_static ? hi_n : hi;
        }
        if (!act) { cur_start = prev_start; lo = 0; hi = Wi; } // a finished group idles in place
        if (act && !is_static && first_lane) bst[ro + row] = cur_start;
        fetch_row(row + 1);
        const int diff = row > 0 ? cur_start - prev_start : 0;

        // the ring covers [cur_start, cur_start + CELLS) (only a large band jump gets here)
        if (__ballot(act && cur_start + CELLS > filled) != 0) {
            if (pf_pending) { ring_store(pf_at + sl, pf); filled = pf_at + LPR; pf_pending = false; }
            while (__ballot(act && cur_start + CELLS > filled) != 0) {
                if (act && cur_start + CELLS > filled) { ring_store(filled + sl, ev_load(filled + sl)); filled += LPR; }
            }
        }
        // shifted half z-scores of my cells (pyx:361-372)
        double z[CPL];
        {
            const double *er = ring + ((cur_start + RING + b0) & (RING - 1));
#pragma unroll
            for (int j = 0; j < CPL; j++) {
                double pz = fabs(div_by_recip(er[j] - mu, sd, y));
                pz = __builtin_fmin(pz, zcap);
                z[j] = zs_lane - pz; // lanes past the band: -inf - pz = -inf
            }
            if (mixed) {
#pragma unroll
                for (int j = 0; j < CPL; j++) z[j] = j < nvalid ? z[j] : NEG_INF;
            }
            if (__ballot(act && (lo != 0 || hi != Wi)) != 0) { // masked start rows / band past the last event
                const double fill = is_static ? fill_masked : MASK_FILL_Z_SCORE; // literal -15, pyx:385-386
#pragma unroll
                for (int j = 0; j < CPL; j++) {
                    z[j] = (b0 + j >= lo && b0 + j < hi) ? z[j] : fill;
                    z[j] = j < nvalid ? z[j] : NEG_INF;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // diag / skip candidates (pyx:220-231, first cell pyx:392-401) from the previous row in LDS:
        // A[k] = previous-row cell (b0 + diff + k - 1) is cell k's diagonal source and cell k-1's skip
        // source; everything outside the band reads -inf
        double cv[CPL];
        u32 tk = 0; // non-stay move code of every cell (1 skip, 2 diag), 2 bits per cell
        {
            int a0 = b0 + diff - 1;
            a0 = a0 > CELLS + 7 ? CELLS + 7 : a0;
            const double *pa = prow + 8 + a0;
            double A[CPL + 1];
#pragma unroll
            for (int k = 0; k <= CPL; k++) A[k] = pa[k];
            const bool fs = diff == 0;
#pragma unroll
            for (int j = 0; j < CPL; j++) {
                const double d = A[j] + z[j];
                const double sk = A[j + 1] - skip_pen;
                bool take_s = sk > d;
                if (j == 0) {
                    take_s = first_lane ? fs : take_s; // band cell 0: skip xor diag
                    cv[j] = take_s ? sk : d;
                } else {
                    cv[j] = max_f64_raw(sk, d);
                }
                tk |= (take_s ? 1u : 2u) << (2 * j);
            }
        }
        // stay chain: exact fixed-point sweeps inside every group (k_dp.h: sweep 1 from -inf, then
        // only the pure stay chain is carried to the lane exits until no incoming value changes);
        // finished groups do not hold the others up
        double in = NEG_INF;
        bool converged = false;
        double exit0;
        {
            double x = NEG_INF;
#pragma unroll
            for (int j = 0; j < CPL; j++) x = max_f64_raw(cv[j], (x - stay_pen) + z[j]);
            exit0 = x;
        }
        {
            double nin = grp_shr1_f64<LPR>(exit0, first_lane);
            for (int it = 0; it < LPR + 2; it++) {
                if (__ballot(act && nin != in) == 0) { converged = true; break; }
                in = nin;
                double c = in;
#pragma unroll
                for (int j = 0; j < CPL; j++) c = (c - stay_pen) + z[j];
                nin = grp_shr1_f64<LPR>(max_f64_raw(exit0, c), first_lane);
            }
            if (!converged) { // only reachable with NaNs in the signal: the groups still changing fail
                const u64 bm = __ballot(act && nin != in);
                const bool mine_bad = ((bm >> grp_base) & (LPR == 64 ? ~0ull : ((1ull << LPR) - 1ull))) != 0;
                if (mine_bad) { if (first_lane) r.status = TBA_INTERNAL; alive = false; act = false; }
            }
        }
        // the cells, their move codes packed 2 bits per cell, lane-local maximum and its first cell
        // (pyx:186-197)
        static_assert(CPL <= 16, "move codes of a lane in one 32-bit word");
        u32 mvw[1];
        double lmax = NEG_INF;
        int lidx = 0;
        {
            double x = in;
            u32 keep = 0; // 3 << 2j where the cell is not a stay (pyx:216-231: strict >)
#pragma unroll
            for (int j = 0; j < CPL; j++) {
                const double s = (x - stay_pen) + z[j];
                keep |= cv[j] > s ? (3u << (2 * j)) : 0u;
                x = max_f64_raw(cv[j], s);
                v[j] = x;
                lidx = x > lmax ? j : lidx;
                lmax = max_f64_raw(lmax, x);
            }
            mvw[0] = tk & keep;
        }
        // the row becomes the previous row (cells past the band keep their -inf)
        if (act) {
            double *pw = prow + 8 + b0;
            if (nvalid == CPL) {
#pragma unroll
                for (int j = 0; j < CPL; j++) pw[j] = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < CPL; j++) if (j < nvalid) pw[j] = v[j];
            }
        }
        if (act) {
            unsigned char *mrow = mv + (row + 1) * mv_stride;
            if constexpr ((2 * CPL) % 8 != 0) {
                // 10 cells = 20 bits per lane: a pair of lanes is 5 whole bytes, written by the even
                // lane (quad_perm [1,1,3,3] brings the odd lane's bits) -- the row stays linear
                static_assert(CPL == 10, "odd class: 10 cells per lane");
                const u32 q0 = mvw[0];
                const u32 q1 = (u32)__builtin_amdgcn_update_dpp(0, (int)q0, 0xf5, 0xf, 0xf, false); // quad_perm:[1,1,3,3]
                if ((sl & 1) == 0) {
                    const u64 bits = (u64)q0 | ((u64)q1 << 20);
                    unsigned char *p = mrow + (sl >> 1) * 5;
#pragma unroll
                    for (int q = 0; q < 5; q++) p[q] = (unsigned char)(bits >> (8 * q));
                }
            } else if constexpr (CPL == 8) *(unsigned short *)(mrow + sl * 2) = (unsigned short)mvw[0];
            else if constexpr (CPL == 16) *(u32 *)(mrow + sl * 4) = mvw[0];
            else {
#pragma unroll
                for (int q = 0; q < CPL / 4; q++) mrow[sl * (CPL / 4) + q] = (unsigned char)(mvw[q / 4] >> (8 * (q % 4)));
            }
        }
        // event ring upkeep: land the chunk in flight, ask for the next one
        if (pf_pending) { ring_store(pf_at + sl, pf); filled = pf_at + LPR; pf_pending = false; }
        if (act && filled < cur_start + CELLS + 3 * LPR && filled < n_ev + CELLS) {
            pf_at = filled; pf = ev_load(filled + sl); pf_pending = true;
        }
        // arg-max of every group: first lane of the group that holds its maximum, that lane's first
        // maximal cell.  The maximum is located in float32 first (k_dp.h), per group.
        {
            float fm = (float)lmax;
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1"
                         : "+v"(fm));
            if constexpr (LPR == 32) { // the other 16-lane row of the group
                const float o = __shfl_xor(fm, 16, 64);
                fm = o > fm ? o : fm;
            }
            // candidate lanes: float32 image equals the group's; the true maximum is among them
            const bool cand = (float)lmax == fm && nvalid > 0;
            // the largest float64 among the candidates, by a group-wide max of the candidates
            // (one candidate per group is the common case: its value IS the maximum)
            const u64 cm = __ballot(cand);
            const u64 gmask = LPR == 64 ? ~0ull : (((1ull << LPR) - 1ull) << grp_base);
            const u64 mine = cm & gmask;
            double wm;
            if (__ballot(__popcll(mine) != 1) == 0) {
                wm = shfl_f64(lmax, __builtin_ctzll(mine | (1ull << 63)));
            } else { // float32 ties inside a group: exact group maximum through LDS-free shuffles
                double t = cand ? lmax : NEG_INF;
#pragma unroll
                for (int o = LPR / 2; o >= 1; o >>= 1) { const double u = shfl_xor_f64(t, o); t = u > t ? u : t; }
                wm = t;
            }
            const u64 eq = __ballot(lmax == wm && nvalid > 0) & gmask;
            const int wl = __builtin_ctzll(eq | (1ull << 63)); // first lane of my group holding the maximum
            const int widx = __shfl(lidx, wl, 64);
            am = (wl - grp_base) * CPL + widx;
        }
        if (__ballot(act && row == n_rows - 1) != 0) {
            if (act && row == n_rows - 1) { // last row + traceback start of a read that ends here
                double *lr = last_row + ri * TBA_MAX_BAND;
#pragma unroll
                for (int j = 0; j < CPL; j++) lr[b0 + j] = v[j];
                if (first_lane) r.top_pos = am;
            }
        }
        prev_start = cur_start;
    }
}
