// k_tail.h -- after the event-space DP: raw-signal resolution of skipped bases, Theil-Sen
// rescale, final score (resquiggle.py:402-540, 1176-1214; tombo_stats.py:401-450, 2327-2338).
#pragma once
#include "k_select.h"

// ---------------------------------------------------------------------------------------------
// rq.resolve_skipped_bases_with_raw (resquiggle.py:402-540): deletion windows, then per window
// c_reg_z_scores (pyx:34-97, reg_start=0, reg_end=n, max_base_shift=n) -> raw_forward_pass
// (resquiggle.py:345-380, c_base_forward_pass pyx:99-163) -> raw_traceback (:382-400,
// c_base_traceback pyx:165-182).  One thread per read; per-read scratch arenas:
//   win[2*(B+1)] i64, fw/z[cap] f64, ld[cap]... (cap elements each, see engine)
struct Win { i64 s, e; };

__device__ inline i64 merge_windows(Win *w, i64 n)
{
    i64 m = 0;
    for (i64 i = 0; i < n; i++) {
        if (m > 0 && w[i].s < w[m - 1].e) w[m - 1].e = w[i].e;
        else w[m++] = w[i];
    }
    return m;
}
__device__ inline void trim_windows(Win *w, i64 n, i64 n_segs)
{
    if (w[0].s < 0) w[0].s = 0;
    if (w[n - 1].e > n_segs - 1) w[n - 1].e = n_segs - 1;
}
__device__ inline int window_too_small(const i64 *segs, i64 n_segs, Win w, i64 m, double extra_sig_factor)
{
    if (w.e >= n_segs || w.s < -n_segs) return -1;
    i64 n_events = w.e - w.s;
    i64 se = segs[w.e < 0 ? w.e + n_segs : w.e], ss = segs[w.s < 0 ? w.s + n_segs : w.s];
    return (double)(se - ss) <= (double)((n_events + 1) * m) * extra_sig_factor;
}

// raw-signal DP of one window.  scratch (8-byte units): fw[n*len] forward scores of every base
// (the traceback compares adjacent rows), zp/zc[len] z-scores of the previous / current base,
// cum[len] np.cumsum of the previous base's z-scores, ld[2*len] last-diagonal counters.
// raw_window_need() is the matching size.
__device__ __forceinline__ i64 raw_window_need(i64 n, i64 len) { return n * len + 5 * len; }

__device__ inline void base_z_row(const double *x, i64 len, double mu, double sd, bool winsor,
                                  double mh, double *z)
{
    for (i64 k = 0; k < len; k++) { // c_base_z_scores, pyx:17-32
        double v = (x[k] - mu) / sd;
        if (v > 0) v = -v;
        if (winsor && v < -mh) v = -mh;
        z[k] = v;
    }
}

__device__ inline int raw_window_dp(const double *sig, i64 L, const double *means,
    const double *sds, i64 n, i64 m, bool winsor, double mh, double *scratch, i64 *new_segs)
{
    if (n < 2) return TBA_INTERNAL;
    // admissible interval of base i: [i*m, L-(n-1-i)*m)  (pyx:56-81): same length for all
    const i64 len = L - (n - 1) * m;
    if (len <= 0) return TBA_INTERNAL;
    if (m == 1) {
        // DNA (raw_min_obs_per_base = 1): the last-diagonal bookkeeping is vacuous (lag is always
        // 1, no tail), so a row is fwd[k] = z[k] + max(prev_fwd[k], fwd[k-1]) with the running
        // value carried in a register; only the forward rows go to scratch (for the traceback)
        double *__restrict__ fwr = scratch;
        {
            const double mu = means[0], sd = sds[0];
            double acc = 0;
            for (i64 k = 0; k < len; k++) {
                double zv = (sig[k] - mu) / sd;
                if (zv > 0) zv = -zv;
                if (winsor && zv < -mh) zv = -mh;
                acc = k == 0 ? zv : acc + zv;
                fwr[k] = acc;
            }
        }
        for (i64 i = 1; i < n; i++) {
            const double mu = means[i], sd = sds[i];
            const double *__restrict__ x = sig + i;
            const double *__restrict__ pf = fwr + (i - 1) * len;
            double *__restrict__ bf = fwr + i * len;
            double stay = 0;
            // (eight positions' loads in flight together: the few windows left to this path are the longest ones, and
            // a load -> use loop pays a memory round trip per position)
            for (i64 k0 = 0; k0 < len; k0 += 8) {
                double xv[8], dg[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const i64 kk = k0 + u < len ? k0 + u : len - 1;
                    xv[u] = x[kk]; dg[u] = pf[kk];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const i64 k = k0 + u;
                    if (k < len) {
                        double zv = (xv[u] - mu) / sd;
                        if (zv > 0) zv = -zv;
                        if (winsor && zv < -mh) zv = -mh;
                        const double best = (k == 0 || dg[u] > stay) ? dg[u] : stay;
                        stay = zv + best;
                        bf[k] = stay;
                    }
                }
            }
        }
        i64 sig_start = (n - 1) + len - 1; // raw_traceback / c_base_traceback, pyx:165-182
        for (i64 b = n - 1; b >= 1; b--) {
            const double *cf = fwr + b * len, *nf = fwr + (b - 1) * len;
            const i64 cs = b, ns = b - 1, ne = ns + len;
            i64 cnt = 1, found = -1;
            for (i64 sp = sig_start; sp >= 0; sp--) {
                cnt += 1;
                if (cnt <= 1 || sp - 1 >= ne) continue;
                if (sp <= cs) { found = sp; break; }
                if (nf[sp - ns - 1] > cf[sp - cs - 1]) { found = sp; break; }
            }
            if (found < 0) return TBA_INTERNAL;
            new_segs[b - 1] = found;
            sig_start = found - 1;
        }
        return TBA_OK;
    }
    double *fw = scratch, *zp = fw + n * len, *zc = zp + len, *cum = zc + len;
    i64 *pl = (i64 *)(cum + len), *bl = pl + len;
    base_z_row(sig, len, means[0], sds[0], winsor, mh, zp);
    { // first row: np.cumsum, last_diag = m (resquiggle.py:352-361)
        double acc = 0;
        for (i64 k = 0; k < len; k++) { acc = k == 0 ? zp[k] : acc + zp[k]; fw[k] = acc; pl[k] = m; }
    }
    for (i64 i = 1; i < n; i++) { // c_base_forward_pass, pyx:99-163
        base_z_row(sig + i * m, len, means[i], sds[i], winsor, mh, zc);
        const double *pf = fw + (i - 1) * len;
        double *bf = fw + i * len;
        const i64 ps = (i - 1) * m, pe = ps + len, b_s = i * m, b_e = b_s + len;
        { double acc = 0; for (i64 k = 0; k < len; k++) { acc = k == 0 ? zp[k] : acc + zp[k]; cum[k] = acc; } }
        if (b_s - ps - 1 < 0 || b_s - ps - 1 >= len) return TBA_INTERNAL;
        bf[0] = zc[0] + pf[b_s - ps - 1];
        bl[0] = 1;
        double stay_run = bf[0]; // bf / bl of the previous position, carried in registers
        i64 bl_run = 1;
        for (i64 pos = b_s + 1; pos < pe + 1; pos++) {
            if (pos - b_s >= len) break;
            i64 lag = 1;
            for (;;) {
                i64 idx = pos - ps - lag;
                if (idx < 0) idx += len;
                if (idx < 0 || idx >= len) return TBA_INTERNAL;
                if (pl[idx] + lag <= m) lag++;
                else break;
            }
            i64 di = pos - ps - lag;
            if (di < 0) di += len;
            double diag = pf[di];
            if (lag > 1) diag += cum[pos - ps - 1] - cum[di];
            const double stay = stay_run;
            double best;
            i64 dv;
            if (diag > stay) { best = diag; dv = 1; }
            else { best = stay; dv = bl_run + 1; }
            stay_run = zc[pos - b_s] + best;
            bl_run = dv;
            bf[pos - b_s] = stay_run;
            bl[pos - b_s] = dv;
        }
        if (b_e > pe + 1) {
            double fv = bf[pe - b_s];
            i64 cl = bl[pe - b_s];
            for (i64 k = 0; k < b_e - pe - 1; k++) {
                fv += zc[k + pe - b_s + 1];
                cl += 1;
                bf[k + pe - b_s + 1] = fv;
                bl[k + pe - b_s + 1] = cl;
            }
        }
        i64 *t = pl; pl = bl; bl = t;
        double *tz = zp; zp = zc; zc = tz;
    }
    i64 sig_start = (n - 1) * m + len - 1; // raw_traceback / c_base_traceback, pyx:165-182
    for (i64 b = n - 1; b >= 1; b--) {
        const double *cf = fw + b * len, *nf = fw + (b - 1) * len;
        const i64 cs = b * m, ns = (b - 1) * m, ne = ns + len;
        i64 cnt = 1, found = -1;
        for (i64 sp = sig_start; sp >= 0; sp--) {
            cnt += 1;
            if (cnt <= m || sp - 1 >= ne) continue;
            if (sp <= cs) { found = sp; break; }
            if (nf[sp - ns - 1] > cf[sp - cs - 1]) { found = sp; break; }
        }
        if (found < 0) return TBA_INTERNAL;
        new_segs[b - 1] = found;
        sig_start = found - 1;
    }
    return TBA_OK;
}

// The DNA windows (raw_min_obs_per_base = 1) of the lane-per-window kernel out of LDS.  A window keeps ONE forward
// row, updated in place, and one bit per (base, position) for the traceback: row i at position k needs fwd[i-1][k]
// (what the slot holds) and its own previous position (a register), and the traceback's test
// "fwd[b-1][j+1] > fwd[b][j]" (pyx:176-180) is the very compare that picked `diag` over `stay` at position j + 1, so
// it is kept as bit j while the row is made.  Same arithmetic in the same order as raw_window_dp's m == 1 path
// (the division by the base's sd through div_by_recip: IEEE, tba_common.h); what it no longer does is write every
// forward row to global scratch and read it back (5.3 GB per cfg2 batch, three 64-line memory instructions per
// position).  Two layouts of the same function (rp / rs: row slot k at rp[k * rs]; fp / fs / words: flag word w of
// base i at fp[(i * words + w) * fs]):
//   column -- the windows of at most SKL_N bases over an admissible interval of at most SKL_LEN samples (48 and 40
//     measured slower: more windows in the flat phase), 97 % of the
//     ~57 windows of a 10 kb read (mean 5 bases x 25 samples): all lanes at once, lane l in column l of
//     row[k][l] (conflict-free), one flag word per base;
//   flat -- the few larger ones, afterwards, up to SKL_FLAT_N at a time: a lane takes a quarter of the same LDS as
//     a flat row of up to SKL_FLAT_LEN slots and a quarter of the flag words.
// Anything larger still goes through global scratch (raw_window_dp).
#ifndef SKL_LEN
#define SKL_LEN 56   // (+ 8 slots a batch of eight positions may run past the interval's end: 32 KB, four wavefronts per CU)
#endif
#define SKL_N 14     // (14 flag rows + 64 row slots + the kernel's own words = 40 192 bytes: four workgroups on a CU, not three)
#define SKL_FLAT_N 4
#define SKL_FLAT_LEN ((SKL_LEN + 8) * 64 / SKL_FLAT_N - 8)   // 1016 samples
#define SKL_FLAT_WORDS (SKL_N * 64 / SKL_FLAT_N)            // 224 flag words: bases x ceil(len / 64)
struct SkipLaneSmem {
    double row[SKL_LEN + 8][64];
    u64 flag[SKL_N][64];
};
__device__ __forceinline__ int raw_window_dp_lane_lds(const double *sig, int len, const double *means,
    const double *sds, int n, bool winsor, double mh, double *rp, int rs, u64 *fp, int fs, int words, i64 *new_segs)
{
    auto zscore = [&](double x, double mu, double sd, double y) { // c_base_z_scores, pyx:17-32
        double zv = div_by_recip(x - mu, sd, y);
        if (zv > 0) zv = -zv;
        if (winsor && zv < -mh) zv = -mh;
        return zv;
    };
    // Eight positions at a time, the samples of the NEXT eight already on their way: a lane's window lies anywhere
    // in the read, so a sample load is a 64-line memory instruction with a round trip of a few thousand cycles at
    // this kernel's occupancy (four wavefronts per CU) -- a step-by-step loop paid it per position (820 cycles per
    // position measured, -DTBA_PHASE_DEBUG=13).  The z-scores of a batch follow side by side; only the max-add
    // chain is serial.  No test of k < len inside a batch (a lone exec-mask branch per position costs more than the
    // position): what runs past the interval's end lands in the spare slots behind it and in registers after their
    // last use.
    auto fetch8 = [&](const double *x, int k0, double *xv) {
#pragma unroll
        for (int u = 0; u < 8; u++) xv[u] = x[k0 + u < len ? k0 + u : len - 1];
    };
    double xn[8];
    fetch8(sig, 0, xn);
    {
        const double mu = means[0], sd = sds[0], y = 1.0 / sd;
        double acc = 0;
        for (int k0 = 0; k0 < len; k0 += 8) { // first row: np.cumsum (resquiggle.py:352-361)
            double xv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) xv[u] = xn[u];
            if (k0 + 8 < len) fetch8(sig, k0 + 8, xn); else if (n > 1) fetch8(sig + 1, 0, xn);
#pragma unroll
            for (int u = 0; u < 8; u++) xv[u] = zscore(xv[u], mu, sd, y);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                acc = k0 + u == 0 ? xv[u] : acc + xv[u];
                rp[(k0 + u) * rs] = acc;
            }
        }
    }
    for (int i = 1; i < n; i++) {
        const double mu = means[i], sd = sds[i], y = 1.0 / sd;
        const double *__restrict__ x = sig + i;
        double stay = 0;
        u64 bits = 0;
        int w_done = -1;                              // last flag word of this base already written
        for (int k0 = 0; k0 < len; k0 += 8) {
            double xv[8], dg[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { xv[u] = xn[u]; dg[u] = rp[(k0 + u) * rs]; }
            if (k0 + 8 < len) fetch8(x, k0 + 8, xn); else if (i + 1 < n) fetch8(x + 1, 0, xn);
#pragma unroll
            for (int u = 0; u < 8; u++) xv[u] = zscore(xv[u], mu, sd, y);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k = k0 + u;
                const bool take = k == 0 || dg[u] > stay;
                bits |= (u64)(take && k > 0 && k < len) << ((k - 1) & 63);
                stay = xv[u] + (take ? dg[u] : stay);
                rp[k * rs] = stay;
                // (position k0 carries bit k0 - 1: at a multiple of 64 that is the last bit of the finished word)
                if (u == 0 && k0 > 0 && (k0 & 63) == 0) { w_done = (k0 >> 6) - 1; fp[(i * words + w_done) * fs] = bits; bits = 0; }
            }
        }
        if (len >= 2 && ((len - 2) >> 6) > w_done) fp[(i * words + ((len - 2) >> 6)) * fs] = bits;
    }
    // raw_traceback / c_base_traceback (pyx:165-182) in the closed form of raw_window_dp_wave
    i64 sig_start = (n - 1) + len - 1;
    for (int b = n - 1; b >= 1; b--) {
        const i64 cs = b, ne = (b - 1) + len;
        const i64 st = sig_start < ne ? sig_start : ne;
        i64 found;
        if (st < 0) return TBA_INTERNAL;
        if (st <= cs) found = st;
        else {
            found = cs;
            const i64 j = st - cs - 1;                // highest candidate flag (j <= len - 2)
            for (i64 w0 = j >> 6; w0 >= 0; w0--) {
                u64 v = fp[(b * words + w0) * fs];
                if (w0 == (j >> 6) && (j & 63) != 63) v &= (1ull << ((j & 63) + 1)) - 1ull;
                if (v) { found = cs + 1 + w0 * 64 + (63 - __clzll((long long)v)); break; }
            }
        }
        new_segs[b - 1] = found;
        sig_start = found - 1;
    }
    return TBA_OK;
}

// get_deletion_windows (resquiggle.py:462-498) + per-window scratch sizing; one thread per read.
// win[3*k..] = (start, end, scratch offset inside the read's slice); r.n_win, r.skip_off (need,
// turned into an arena offset by k_scan_skip).
// Windows of at least SKIP_WAVE_MIN positions (bases x admissible interval) whose interval is at
// most SKIP_WAVE_LEN samples are resolved by a whole wavefront out of LDS (k_skip_dp_wave): the
// kernel time of the lane-per-window form is the serial walk of its LARGEST window (RNA: mean 3.5 k
// positions per read, but 28 k in the largest window of a 10 k-read batch).  win[3k+2] = -1 marks
// them; the rest keep their offset inside the read's slice of the global scratch arena.
// Three LDS classes (interval length, flag words); windows of a class are queued in a global list
// (skipq: [c] = entries of the list of class c, [4 + c] = next entry to hand out) that a fixed
// grid of wavefronts drains, so no workgroup is launched per read.
// smallest window (positions) handed to the wave form.  Only used for raw_min_obs_per_base > 1
// (RNA): the lane form's DNA fast path (recurrence in a register, windows of a few hundred
// positions, ~55 per read) is faster than queueing (3.4 vs 3.9 ms per 10 k reads)
#define SKIP_WAVE_MIN 256
#define SKIP_LEN_S 320
#define SKIP_BITS_S 128
#define SKIP_LEN_M 640
#define SKIP_BITS_M 384
#define SKIP_LEN_B 1792
#define SKIP_BITS_B 1024 // u64 words of traceback flags (n * ceil(len / 64))

#define SKP_LDS_DELS 256   // (6 KB of LDS: the scan of the boundaries wants many wavefronts per CU)
__global__ __launch_bounds__(64) void k_skip_plan(ReadState *rs, i64 n_reads, const DevParams *dp,
    const i64 *dp_segs, i64 *segs_out, i64 *win_scratch, i64 *skipq, i32 *lists, i64 list_cap)
{
    (void)n_reads;
    ReadState &r = rs[blockIdx.x];
    const int lane = threadIdx.x;
    if (lane == 0) { r.n_win = 0; r.skip_off = 0; }
    if (r.status != TBA_OK) return;
    const i64 m = dp->p.raw_min_obs_per_base;
    const i64 n_segs = r.B + 1;
    const i64 *ds = dp_segs + r.seg_off;
    // the resolved boundaries start as a copy; the window kernels overwrite their interiors.  The same pass finds the
    // skipped bases (diff(segs) == 0), kept in order behind the window area.  Eight strides of loads in flight: one
    // wavefront per read, and a load -> use loop of 157 steps pays 157 memory round trips (half of this kernel).
    i64 *dels = win_scratch + 3 * r.seg_off + 2 * n_segs;
    i64 n_del = 0;
    for (i64 base = 0; base < n_segs; base += 64 * 8) {
        i64 a8[8], b8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const i64 d = base + 64 * u + lane;
            a8[u] = ds[d < n_segs ? d : n_segs - 1];
            b8[u] = ds[d + 1 < n_segs ? d + 1 : n_segs - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const i64 d = base + 64 * u + lane;
            if (d < n_segs) segs_out[r.seg_off + d] = a8[u];
            const bool flag = d + 1 < n_segs && b8[u] == a8[u];
            const u64 mask = __ballot(flag);
            if (flag) dels[n_del + __popcll(mask & ((1ull << lane) - 1ull))] = d;
            n_del += __popcll(mask);
        }
    }
    __syncthreads();
    // ---- the usual case (at most SKP_LDS_DELS skipped bases): windows in LDS, per-window work on all lanes.
    // Lane 0 alone over global memory (the form below, kept for reads with more deletions) paid a memory round
    // trip for every window it touched -- ~350 of them, 0.5 of the kernel's 0.84 ms.  Same windows, same statuses:
    // the first failing window in window order decides, as in the serial loops.
    if (n_del <= SKP_LDS_DELS) {
        __shared__ i64 s_aux[SKP_LDS_DELS];              // the deletions, later each window's scratch need / class
        __shared__ Win s_w[SKP_LDS_DELS];
        __shared__ i64 s_nw;
        __shared__ int s_st;
        const i64 dfw = dp->o.del_fix_window, mdfw = dp->o.max_del_fix_window;
        const double esf = dp->o.extra_sig_factor;
        for (i64 q = lane; q < n_del; q += 64) s_aux[q] = dels[q];
        if (lane == 0) s_st = TBA_OK;
        __syncthreads();
        if (lane == 0) {
            i64 nw = 0;
            for (i64 q = 0; q < n_del; q++) {
                const i64 d = s_aux[q];
                if (nw > 0 && d < s_w[nw - 1].e + dfw) s_w[nw - 1].e = d + dfw + 1;
                else { s_w[nw].s = d - dfw; s_w[nw].e = d + dfw + 1; nw++; }
            }
            if (nw > 0) { nw = merge_windows(s_w, nw); trim_windows(s_w, nw, n_segs); }
            s_nw = nw;
        }
        __syncthreads();
        if (s_nw == 0) return;
        // first failing window of a parallel check (code 0: fine), in window order
        auto first_code = [&](auto code_of) -> int {
            int res = 0;
            for (i64 i0 = 0; i0 < s_nw && res == 0; i0 += 64) {
                const i64 i = i0 + lane;
                const int c = i < s_nw ? code_of(i) : 0;
                const u64 mk = __ballot(c != 0);
                if (mk) res = __shfl(c, __ffsll((unsigned long long)mk) - 1, 64);
            }
            return res;
        };
        bool expanded = false;
        for (i64 it = 0; it < mdfw - dfw; it++) {
            bool any = false;
            const int err = first_code([&](i64 i) {
                const int ts = window_too_small(ds, n_segs, s_w[i], m, esf);
                if (ts > 0) any = true;
                return ts < 0 ? (int)TBA_INTERNAL : 0;
            });
            if (err) { if (lane == 0) r.status = err; return; }
            expanded = __ballot(any) != 0;
            if (!expanded) break;
            __syncthreads();
            for (i64 i = lane; i < s_nw; i += 64)       // (every lane re-derives its windows' verdicts: cheap, from cache)
                if (window_too_small(ds, n_segs, s_w[i], m, esf) > 0) { s_w[i].s -= 1; s_w[i].e += 1; }
            __syncthreads();
            if (lane == 0) { const i64 nw = merge_windows(s_w, s_nw); trim_windows(s_w, nw, n_segs); s_nw = nw; }
            __syncthreads();
        }
        if (expanded) {
            const int err = first_code([&](i64 i) {
                const int ts = window_too_small(ds, n_segs, s_w[i], m, esf);
                return ts < 0 ? (int)TBA_INTERNAL : (ts ? (int)TBA_NOT_ENOUGH_DEL_SIGNAL : 0);
            });
            if (err) { if (lane == 0) r.status = err; return; }
        }
        if (dp->o.max_raw_cpts >= 0) {
            const int err = first_code([&](i64 i) { return s_w[i].e - s_w[i].s > dp->o.max_raw_cpts ? (int)TBA_TOO_MANY_DELS : 0; });
            if (err) { if (lane == 0) r.status = err; return; }
        }
        // per window: class / scratch need (s_aux[i]: -1 wave form, -2 LDS column, -3 LDS flat, >= 0 arena doubles)
        i64 *w3 = win_scratch + 3 * r.seg_off;
        {
            const int err = first_code([&](i64 i) {
                const i64 s = s_w[i].s, e = s_w[i].e, n = e - s;
                if (s < 0 || e >= n_segs) return (int)TBA_INTERNAL;
                const i64 L = ds[e] - ds[s];
                const i64 len = L - (n - 1) * m;
                if (len <= 0 || n < 2) return (int)TBA_INTERNAL;
                w3[3 * i] = s; w3[3 * i + 1] = e;
                const i64 fw = n * ((len + 63) / 64);
                i64 code;
                if (m == 1 && len <= SKL_LEN && n <= SKL_N) code = -2;
                else if (m == 1 && len <= SKL_FLAT_LEN && fw <= SKL_FLAT_WORDS) code = -3;
                else {
                    code = raw_window_need(n, len);
                    if (m > 1 && n * len >= SKIP_WAVE_MIN) {
                        const bool fits = (n - 1) * m <= 512 && n <= 256;
                        int cls = -1;
                        if (fits && len <= SKIP_LEN_S && fw <= SKIP_BITS_S) cls = 0;
                        else if (fits && len <= SKIP_LEN_M && fw <= SKIP_BITS_M) cls = 1;
                        else if (fits && len <= SKIP_LEN_B && fw <= SKIP_BITS_B) cls = 2;
                        if (cls >= 0) { // queue for k_skip_dp_wave (a full list leaves the window to the lane form)
                            const i64 pos = (i64)atomicAdd((unsigned long long *)&skipq[cls], 1ull);
                            if (pos < list_cap) {
                                i32 *lst = lists + 2 * list_cap * cls;
                                lst[2 * pos] = (i32)blockIdx.x; lst[2 * pos + 1] = (i32)i;
                                code = -1;
                            }
                        }
                    }
                }
                s_aux[i] = code;
                return 0;
            });
            if (err) { if (lane == 0) r.status = err; return; }
        }
        __syncthreads();
        if (lane == 0) { // arena offsets in window order
            i64 acc = 0;
            for (i64 i = 0; i < s_nw; i++) {
                const i64 c = s_aux[i];
                if (c >= 0) { s_aux[i] = acc; acc += c; }
            }
            r.n_win = s_nw;
            r.skip_off = acc;
        }
        __syncthreads();
        for (i64 i = lane; i < s_nw; i += 64) w3[3 * i + 2] = s_aux[i];
        return;
    }
    if (lane != 0) return;
    // windows are built in place as (s, e) pairs, then widened to (s, e, off) triples back to
    // front; capacity 3 * (B + 1) entries per read
    Win *w = (Win *)(win_scratch + 3 * r.seg_off);
    // del_fix_window / max_del_fix_window / extra_sig_factor of resolve_skipped_bases_with_raw
    // (resquiggle.py:405-407; defaults _default_parameters.py:67,72,73)
    const i64 dfw = dp->o.del_fix_window, mdfw = dp->o.max_del_fix_window;
    const double esf = dp->o.extra_sig_factor;
    i64 nw = 0;
    for (i64 q = 0; q < n_del; q++) {
        const i64 d = dels[q];
        if (nw > 0 && d < w[nw - 1].e + dfw) w[nw - 1].e = d + dfw + 1;
        else { w[nw].s = d - dfw; w[nw].e = d + dfw + 1; nw++; }
    }
    if (nw == 0) return;
    bool expanded = false;
    nw = merge_windows(w, nw);
    trim_windows(w, nw, n_segs);
    for (i64 it = 0; it < mdfw - dfw; it++) {
        expanded = false;
        for (i64 i = 0; i < nw; i++) {
            int ts = window_too_small(ds, n_segs, w[i], m, esf);
            if (ts < 0) { r.status = TBA_INTERNAL; return; }
            if (ts) { expanded = true; w[i].s -= 1; w[i].e += 1; }
        }
        if (!expanded) break;
        nw = merge_windows(w, nw);
        trim_windows(w, nw, n_segs);
    }
    if (expanded) {
        for (i64 i = 0; i < nw; i++) {
            int ts = window_too_small(ds, n_segs, w[i], m, esf);
            if (ts < 0) { r.status = TBA_INTERNAL; return; }
            if (ts) { r.status = TBA_NOT_ENOUGH_DEL_SIGNAL; return; }
        }
    }
    if (dp->o.max_raw_cpts >= 0) {
        i64 mx = 0;
        for (i64 i = 0; i < nw; i++) mx = w[i].e - w[i].s > mx ? w[i].e - w[i].s : mx;
        if (mx > dp->o.max_raw_cpts) { r.status = TBA_TOO_MANY_DELS; return; }
    }
    i64 *w3 = win_scratch + 3 * r.seg_off;
    for (i64 i = nw - 1; i >= 0; i--) { // widen pairs to triples, back to front
        const i64 s = w[i].s, e = w[i].e;
        w3[3 * i] = s; w3[3 * i + 1] = e; w3[3 * i + 2] = 0;
    }
    i64 acc = 0;
    for (i64 i = 0; i < nw; i++) {
        const i64 s = w3[3 * i], e = w3[3 * i + 1], n = e - s;
        if (s < 0 || e >= n_segs) { r.status = TBA_INTERNAL; return; }
        const i64 L = ds[e] - ds[s];
        const i64 len = L - (n - 1) * m;
        if (len <= 0 || n < 2) { r.status = TBA_INTERNAL; return; }
        const i64 fw = n * ((len + 63) / 64);
        int cls = -1;
        if (m == 1 && len <= SKL_LEN && n <= SKL_N) { w3[3 * i + 2] = -2; continue; } // k_skip_dp out of LDS: column
        if (m == 1 && len <= SKL_FLAT_LEN && n * ((len + 63) / 64) <= SKL_FLAT_WORDS) { w3[3 * i + 2] = -3; continue; } // flat
        if (m > 1 && n * len >= SKIP_WAVE_MIN) {
            // (L = len + (n - 1) m must fit the staged signal, n the staged levels)
            const bool fits = (n - 1) * m <= 512 && n <= 256;
            if (fits && len <= SKIP_LEN_S && fw <= SKIP_BITS_S) cls = 0;
            else if (fits && len <= SKIP_LEN_M && fw <= SKIP_BITS_M) cls = 1;
            else if (fits && len <= SKIP_LEN_B && fw <= SKIP_BITS_B) cls = 2;
        }
        if (cls >= 0) { // queue for k_skip_dp_wave (a full list leaves the window to the lane form)
            const i64 pos = (i64)atomicAdd((unsigned long long *)&skipq[cls], 1ull);
            if (pos < list_cap) {
                i32 *lst = lists + 2 * list_cap * cls;
                lst[2 * pos] = (i32)blockIdx.x; lst[2 * pos + 1] = (i32)i;
                w3[3 * i + 2] = -1;
                continue;
            }
        }
        w3[3 * i + 2] = acc;
        acc += raw_window_need(n, len);
    }
    r.n_win = nw;
    r.skip_off = acc;
}

// raw-signal DP of one (large) window by one WAVEFRONT out of LDS.  Same arithmetic, in the same
// order, as raw_window_dp (c_reg_z_scores -> raw_forward_pass / c_base_forward_pass ->
// raw_traceback / c_base_traceback).  What is independent per signal position -- the z-scores of a
// base, the lag search and the diagonal source of every position (pyx:127-140), the traceback
// comparison of adjacent rows -- is spread over the 64 lanes; the two serial chains (np.cumsum,
// the stay recurrence) run on lane 0 out of registers, eight positions per LDS round trip.  Only
// two forward rows are live: the traceback's test "previous base's score > this base's score"
// (pyx:176-180) is evaluated for every position when a row is finished and kept as one bit.
// LDS is what limits the wavefronts in flight, and the wavefronts in flight are the kernel's speed (each is one lane's
// recurrence most of the time), so a position costs 36 bytes, not 72 (round 6): the diagonal sources are dropped
// into the row they are about to become (read a batch ahead of the write), a base's z-scores become their cumulative sums in
// place (the next base needs exactly those, and the array of the base before is free by then), the counters are 16-bit
// (<= LEN), and the largest class reads its signal from global memory (SIG_LDS = false: one coalesced pass per base).
// Workgroups per CU: 5 / 2 / 1 -> 6 / 4 / 2.
template <int LEN, int BITS, bool SIG_LDS>
struct SkipWaveSmem {
    double row[2][LEN];            // forward scores of the previous / current base
    double z[2][LEN];              // z-scores, then cumulative z-scores, of the current base; the previous base's
    short l[2][LEN];               // last-diagonal counters
    u64 bits[BITS];                // row b, word w: bits[b * words + w]
    double mu[256], sd[256];       // expected level / sd of the window's bases
    double sg[SIG_LDS ? LEN + 512 : 1]; // the window's signal (L = len + (n - 1) m)
};
#ifdef TBA_SKIP_STATS
#define SKP(i_) do { const i64 t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) skp[i_] += t_ - tl_; tl_ = __builtin_readcyclecounter(); } while (0)
#else
#define SKP(i_) do { } while (0)
#endif
template <int LEN, int BITS, bool SIG_LDS>
__device__ inline int raw_window_dp_wave(const double *sig, i64 L, const double *means,
    const double *sds, i64 n, i64 m, bool winsor, double mh, SkipWaveSmem<LEN, BITS, SIG_LDS> &S, i64 *new_segs
#ifdef TBA_SKIP_STATS
    , i64 *skp
#endif
    )
{
    const int lane = threadIdx.x;
#ifdef TBA_SKIP_STATS
    i64 tl_ = __builtin_readcyclecounter();
#endif
    if (n < 2) return TBA_INTERNAL;
    const i64 len = L - (n - 1) * m;
    if (len <= m || len > LEN) return TBA_INTERNAL; // (the planner's windows always have len > 2 m)
    const i64 words = (len + 63) / 64;
    if (n * words > BITS) return TBA_INTERNAL;
    double *cum = S.z[0], *zc = S.z[1]; // cum: np.cumsum of the previous base's scores; zc: this base's, in the making
    short *pl = S.l[0], *bl = S.l[1];
    if (L > LEN + 512 || n > 256) return TBA_INTERNAL;
    // the whole window's signal and levels come into LDS in one go (all loads in flight together:
    // one global round trip per window instead of one per base)
    if (SIG_LDS) for (i64 k = lane; k < L; k += 64) S.sg[k] = sig[k];
    for (i64 k = lane; k < n; k += 64) { S.mu[k] = means[k]; S.sd[k] = sds[k]; }
    for (i64 k = lane; k < len; k += 64) pl[k] = (short)m;
    __syncthreads();
    auto zrow = [&](i64 base, double *z) { // c_base_z_scores, pyx:17-32
        const double *x = (SIG_LDS ? S.sg : sig) + base * m;
        const double mu = S.mu[base], sd = S.sd[base];
        for (i64 k = lane; k < len; k += 64) {
            double v = (x[k] - mu) / sd;
            if (v > 0) v = -v;
            if (winsor && v < -mh) v = -mh;
            z[k] = v;
        }
    };
    zrow(0, cum);
    __syncthreads();
    if (lane == 0) { // first row: np.cumsum (resquiggle.py:352-361) = the cumulative z-scores too (in place)
        double acc = 0;
        for (i64 k0 = 0; k0 < len; k0 += 8) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = cum[k0 + u < len ? k0 + u : len - 1];
#pragma unroll
            for (int u = 0; u < 8; u++) { acc = k0 + u == 0 ? t[u] : acc + t[u]; t[u] = acc; }
#pragma unroll
            for (int u = 0; u < 8; u++) if (k0 + u < len) { S.row[0][k0 + u] = t[u]; cum[k0 + u] = t[u]; }
        }
    }
    int bad = 0;
    for (i64 i = 1; i < n; i++) { // c_base_forward_pass, pyx:99-163
        const double *pf = S.row[(i - 1) & 1];
        double *bf = S.row[i & 1];
        SKP(3);
        zrow(i, zc);
        __syncthreads();
        SKP(4);
        // cum = np.cumsum of the previous base's scores (left by the previous row's walk)
        const i64 k_last = len - m < len - 1 ? len - m : len - 1; // pos <= pe, k = pos - b_s < len
        for (i64 k = 1 + lane; k <= k_last; k += 64) { // diagonal sources
            i64 lag = 1;
            for (;;) {
                const i64 idx = k + m - lag;
                if (idx < 0 || idx >= len) { bad = 1; break; }
                if (pl[idx] + lag <= m) lag++;
                else break;
            }
            if (bad) break;
            const i64 di = k + m - lag;
            double diag = pf[di];
            if (lag > 1) diag += cum[k + m - 1] - cum[di];
            bf[k] = diag;                        // (the row two bases back is dead; lane 0 reads a batch, then writes it)
        }
        __syncthreads();
        SKP(5);
        if (lane == 0) { // the stay recurrence, and on the way this base's cumulative z-scores
            double stay_run = zc[0] + pf[m - 1];
            double csum = zc[0];
            i32 bl_run = 1;
            bf[0] = stay_run; bl[0] = 1;         // (zc[0] is its own cumulative sum)
            // A lone lane pays for every dependent instruction (~10 cycles each) and dearly for
            // every exec-mask branch, so a step is kept to: best = max(diag, stay) (the selected
            // value of "diag > stay ? diag : stay"; the strict compare only feeds the counter,
            // off the critical chain), stay = z + best.  Past the previous base's interval
            // (pyx:151-161) the diagonal is -inf: best = stay, i.e. z + stay -- the reference's
            // running sum with the operands swapped (addition commutes exactly).  Full batches of
            // eight positions first, the remainder one by one.
            i64 k0 = 1;
            for (; k0 + 8 <= len; k0 += 8) {
                double dv[8], zv[8], cv[8];
                i32 lv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const i64 k = k0 + u;
                    dv[u] = bf[k <= k_last ? k : k_last]; zv[u] = zc[k]; // (past k_last: whatever is there, not used)
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const double d = k0 + u <= k_last ? dv[u] : -INFINITY;
                    const bool take = d > stay_run;
                    stay_run = zv[u] + max_f64_raw(d, stay_run);
                    bl_run = take ? 1 : bl_run + 1;
                    csum = csum + zv[u];
                    dv[u] = stay_run; lv[u] = bl_run; cv[u] = csum;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) { bf[k0 + u] = dv[u]; bl[k0 + u] = (short)lv[u]; zc[k0 + u] = cv[u]; }
            }
            for (i64 k = k0; k < len; k++) {
                const double d = k <= k_last ? bf[k] : -INFINITY, zv = zc[k];
                const bool take = d > stay_run;
                stay_run = zv + max_f64_raw(d, stay_run);
                bl_run = take ? 1 : bl_run + 1;
                csum = csum + zv;
                bf[k] = stay_run; bl[k] = (short)bl_run; zc[k] = csum;
            }
        }
        if (__syncthreads_or(bad)) return TBA_INTERNAL;
        SKP(6);
        // traceback test of this base: position sp <-> j = sp - cs - 1 in this row, j + m in the
        // previous one; bit j = previous[j + m] > this[j]
        for (i64 w0 = 0; w0 < words; w0++) {
            const i64 j = w0 * 64 + lane;
            const bool f = j + m < len && pf[j + m < len ? j + m : 0] > bf[j < len ? j : 0];
            const u64 mk = __ballot(f);
            if (lane == 0) S.bits[i * words + w0] = mk;
        }
        __syncthreads();
        short *t = pl; pl = bl; bl = t;
        double *tz = cum; cum = zc; zc = tz;     // this base's cumulative scores are the next one's `cum`
        SKP(7);
    }
    int rc = TBA_OK;
    if (lane == 0) { // raw_traceback / c_base_traceback, pyx:165-182
        // The reference walks sp down from sig_start: the first m - 1 positions and those with
        // sp - 1 >= next_end are passed over, then the first position with sp <= curr_start or
        // with the flag set is returned.  Same result without the walk: start at
        // min(sig_start - (m - 1), next_end); at or below curr_start -> that position; else the
        // highest set flag at or below it (word-wise), else curr_start.
        i64 sig_start = (n - 1) * m + len - 1;
        for (i64 b = n - 1; b >= 1; b--) {
            const i64 cs = b * m, ne = (b - 1) * m + len;
            i64 st = sig_start - (m - 1);
            st = st < ne ? st : ne;
            i64 found;
            if (st < 0) { rc = TBA_INTERNAL; break; }
            if (st <= cs) found = st;
            else {
                found = cs;
                const u64 *bw = S.bits + b * words;
                i64 j = st - cs - 1; // highest candidate flag
                for (i64 w0 = j >> 6; w0 >= 0; w0--) {
                    u64 v = bw[w0];
                    if (w0 == (j >> 6) && (j & 63) != 63) v &= (1ull << ((j & 63) + 1)) - 1ull;
                    if (v) { found = cs + 1 + w0 * 64 + (63 - __clzll((long long)v)); break; }
                }
            }
            new_segs[b - 1] = found;
            sig_start = found - 1;
        }
    }
    return __syncthreads_or(rc != TBA_OK) ? TBA_INTERNAL : TBA_OK;
}

// drains one window queue: a fixed grid of one-wavefront workgroups, each fetching the next
// queued (read, window) until the list is empty
template <int LEN, int BITS, int CLS>
__global__ __launch_bounds__(64) void k_skip_dp_wave(ReadState *rs, const DevParams *dp,
    const double *norm, const double *ref_means, const double *ref_sds, const i64 *dp_segs,
    i64 *segs, const i64 *win_scratch, i64 *skipq, const i32 *list, i64 list_cap)
{
    constexpr bool SIG_LDS = CLS != 2;           // (the largest class: see SkipWaveSmem)
    __shared__ SkipWaveSmem<LEN, BITS, SIG_LDS> S;
    __shared__ i64 s_id;
    const tba_params &P = dp->p;
    const i64 m = P.raw_min_obs_per_base;
    i64 total = skipq[CLS];
    total = total < list_cap ? total : list_cap;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_id = (i64)atomicAdd((unsigned long long *)&skipq[4 + CLS], 1ull);
        __syncthreads();
        const i64 id = s_id;
        if (id >= total) break;
        ReadState &r = rs[list[2 * id]];
        const i64 i = list[2 * id + 1];
        const i64 *ds = dp_segs + r.seg_off;
        i64 *out = segs + r.seg_off;
        const double *sig = norm + r.raw_off + r.read_start;
        const i64 *w3 = win_scratch + 3 * r.seg_off;
        const i64 s = w3[3 * i], e = w3[3 * i + 1], n = e - s;
        const i64 sig_start = ds[s], sig_end = ds[e];
        int rc = TBA_OK;
#ifdef TBA_SKIP_CLASS_STATS
        const i64 tcs_ = __builtin_readcyclecounter();
#endif
#ifdef TBA_SKIP_STATS
        const i64 t0_ = __builtin_readcyclecounter();
#endif
        if (sig_start < 0 || sig_end > r.norm_len) rc = TBA_INTERNAL;
        else {
            rc = raw_window_dp_wave<LEN, BITS, SIG_LDS>(sig + sig_start, sig_end - sig_start,
                                               ref_means + r.ref_off + s, ref_sds + r.ref_off + s, n, m,
                                               P.do_winsorize_z != 0, P.max_half_z_score, S, out + s + 1
#ifdef TBA_SKIP_STATS
                                               , r.dbg
#endif
                                               );
            wave_mem_fence(); // (lane 0 wrote the boundaries, every lane reads them back)
            if (rc == TBA_OK)
                for (i64 k = threadIdx.x; k < n - 1; k += 64) out[s + 1 + k] += sig_start;
        }
        // (every failure of the window DP is the same "unexpected error" status)
        if (rc != TBA_OK && threadIdx.x == 0) r.status = rc;
#ifdef TBA_SKIP_CLASS_STATS   // per read: windows and cycles of every class (tools/stage_times.py, TBA_DBG_PHASES=1: the sums)
        if (threadIdx.x == 0) {
            atomicAdd((unsigned long long *)&r.dbg[CLS], 1ull);
            atomicAdd((unsigned long long *)&r.dbg[3 + CLS], (unsigned long long)(__builtin_readcyclecounter() - tcs_));
        }
#endif
#ifdef TBA_SKIP_STATS
        if (threadIdx.x == 0) {
            atomicAdd((unsigned long long *)&r.dbg[0], 1ull);
            atomicAdd((unsigned long long *)&r.dbg[1], (unsigned long long)(__builtin_readcyclecounter() - t0_));
            atomicAdd((unsigned long long *)&r.dbg[2], (unsigned long long)(n * (sig_end - sig_start - (n - 1) * m)));
        }
#endif
    }
}

// rq.resolve_skipped_bases_with_raw window loop + final checks (resquiggle.py:500-538).
// One wavefront per read, one window per lane at a time (windows own disjoint boundary ranges).
template <bool DNA_LDS> // raw_min_obs_per_base == 1: the windows k_skip_plan marked -2 run out of LDS
__global__ __launch_bounds__(64) void k_skip_dp(ReadState *rs, const DevParams *dp,
    const double *norm, const double *ref_means, const double *ref_sds, const i64 *dp_segs,
    i64 *segs, const i64 *win_scratch, double *arena)
{
    ReadState &r = rs[blockIdx.x];
    if (r.status != TBA_OK) return;
    const int lane = threadIdx.x;
    const tba_params &P = dp->p;
    const i64 m = P.raw_min_obs_per_base;
    const i64 n_segs = r.B + 1;
    const i64 *ds = dp_segs + r.seg_off;
    i64 *out = segs + r.seg_off;
    const double *sig = norm + r.raw_off + r.read_start; // norm_signal[read_start:...]
    const i64 n_norm = r.norm_len;
    const double *mu = ref_means + r.ref_off, *sd = ref_sds + r.ref_off;
    // (out already holds the copy of dp_segs made by k_skip_plan, with the interiors of the large
    // windows resolved by k_skip_dp_wave; what is left here are the other windows, one per lane
    // over global scratch, and the final checks)
    const i64 *w3 = win_scratch + 3 * r.seg_off;
    int rc = TBA_OK;
#if TBA_PHASE_DEBUG_OR0 == 13
    // -DTBA_PHASE_DEBUG=13: cycles of the wavefront (0 whole kernel, 1 its window loops, 2 the column phase, 3 the flat
    // phase, 6 windows)
    const i64 pt0 = (i64)__builtin_readcyclecounter();
    i64 pc_lds = 0, pc_arena = 0;
#endif
    if constexpr (DNA_LDS) {
        __shared__ SkipLaneSmem S_;
        const bool winsor = P.do_winsorize_z != 0;
        // phase 1: the small windows, every lane in its column
        for (i64 i = lane; i < r.n_win && rc == TBA_OK; i += 64) {
            if (w3[3 * i + 2] != -2) continue;
            const i64 s = w3[3 * i], e = w3[3 * i + 1], n = e - s;
            const i64 sig_start = ds[s], sig_end = ds[e];
            if (sig_start < 0 || sig_end > n_norm) { rc = TBA_INTERNAL; break; }
            const int rr = raw_window_dp_lane_lds(sig + sig_start, (int)(sig_end - sig_start - (n - 1)), mu + s, sd + s, (int)n,
                                                  winsor, P.max_half_z_score, &S_.row[0][lane], 64, &S_.flag[0][lane], 64, 1, out + s + 1);
            if (rr != TBA_OK) { rc = rr; break; }
            for (i64 k = 0; k < n - 1; k++) out[s + 1 + k] += sig_start;
        }
#if TBA_PHASE_DEBUG_OR0 == 13
        const i64 pt_a = (i64)__builtin_readcyclecounter();
#endif
        // phase 2: the larger ones, SKL_FLAT_N at a time, each in its quarter of the same memory
        for (i64 i0 = 0; i0 < r.n_win; i0 += 64) {
            const i64 i = i0 + lane;
            bool todo = i < r.n_win && w3[3 * i + 2] == -3 && rc == TBA_OK;
            for (;;) {
                const u64 mk = __ballot(todo);
                if (mk == 0) break;
                const int rank = __popcll(mk & ((1ull << lane) - 1ull));
                __syncthreads();                                  // (the previous round's rows are done with)
                if (todo && rank < SKL_FLAT_N) {
                    todo = false;
                    const i64 s = w3[3 * i], e = w3[3 * i + 1], n = e - s;
                    const i64 sig_start = ds[s], sig_end = ds[e];
                    int rr = TBA_INTERNAL;
                    if (sig_start >= 0 && sig_end <= n_norm) {
                        const int len = (int)(sig_end - sig_start - (n - 1));
                        rr = raw_window_dp_lane_lds(sig + sig_start, len, mu + s, sd + s, (int)n, winsor, P.max_half_z_score,
                                                    &S_.row[0][0] + rank * (SKL_FLAT_LEN + 8), 1,
                                                    &S_.flag[0][0] + rank * SKL_FLAT_WORDS, 1, (len + 63) >> 6, out + s + 1);
                    }
                    if (rr != TBA_OK) rc = rr;
                    else for (i64 k = 0; k < n - 1; k++) out[s + 1 + k] += sig_start;
                }
            }
        }
#if TBA_PHASE_DEBUG_OR0 == 13
        pc_lds = pt_a - pt0; pc_arena = (i64)__builtin_readcyclecounter() - pt_a;
#endif
    }
    // what is left: windows over global scratch (raw_min_obs_per_base > 1: all the small ones; DNA: none in practice)
    for (i64 i = lane; i < r.n_win && rc == TBA_OK; i += 64) {
        if (w3[3 * i + 2] < 0) continue;                      // k_skip_dp_wave's, or done above
        const i64 s = w3[3 * i], e = w3[3 * i + 1], n = e - s;
        const i64 sig_start = ds[s], sig_end = ds[e];
        if (sig_start < 0 || sig_end > n_norm) { rc = TBA_INTERNAL; break; }
        const int rr = raw_window_dp(sig + sig_start, sig_end - sig_start, mu + s, sd + s, n, m,
                                     P.do_winsorize_z != 0, P.max_half_z_score,
                                     arena + r.skip_off + w3[3 * i + 2], out + s + 1);
        if (rr != TBA_OK) { rc = rr; break; }
        for (i64 k = 0; k < n - 1; k++) out[s + 1 + k] += sig_start;
    }
#if TBA_PHASE_DEBUG_OR0 == 13
    {
        const i64 pt1 = (i64)__builtin_readcyclecounter();
        if (lane == 0) { r.dbg[1] = pt1 - pt0; r.dbg[2] = pc_lds; r.dbg[3] = pc_arena; r.dbg[6] = r.n_win; }
    }
#endif
    // first failing window in window order decides the status, as in the sequential loop
    // (all window failures here are non-Tombo errors, so any of them is "unexpected")
    if (__syncthreads_or(rc != TBA_OK)) {
        int code = rc != TBA_OK ? rc : 0x7fffffff;
        for (int o = 32; o >= 1; o >>= 1) { int t = __shfl_xor(code, o, 64); code = t < code ? t : code; }
        if (lane == 0) r.status = code;
        return;
    }
    wave_mem_fence(); // (the checks read boundaries other lanes wrote)
    int flag = 0; // 1: zero-length event
    // (eight strides of loads in flight: one after the other this loop was a third of the kernel -- 157 round trips)
    for (i64 i0 = lane; i0 + 1 < n_segs; i0 += 64 * 8) {
        i64 a8[8], b8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const i64 i = i0 + 64 * u < n_segs - 1 ? i0 + 64 * u : n_segs - 2;
            a8[u] = out[i]; b8[u] = out[i + 1];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) if (i0 + 64 * u + 1 < n_segs && b8[u] - a8[u] < 1) flag = 1;
    }
    if (__syncthreads_or(flag)) { if (lane == 0) r.status = TBA_ZERO_LEN; return; }
    if (lane == 0) {
        if (out[0] < 0) r.status = TBA_NEG_START;
        else if (out[n_segs - 1] > n_norm) r.status = TBA_PAST_END;
#if TBA_PHASE_DEBUG_OR0 == 13
        r.dbg[0] = (i64)__builtin_readcyclecounter() - pt0;
#endif
    }
}

// ts.calc_kmer_fitted_shift_scale(method='theil_sen') (tombo_stats.py:401-450) with
// c_compute_slopes (_c_helper.pyx:362-377): median of all pairwise slopes, then median
// intercept.  One workgroup per read; the (<= 1000) points sit in LDS, the n(n-1)/2 slopes are
// recomputed inside every radix-select pass instead of being stored (4 MB per read otherwise).
// Pair enumeration by circular distance: (i, (i+d) mod n); slope(i,j) == slope(j,i) bitwise.
#ifndef TSW_SAMPLE_DIST
#define TSW_SAMPLE_DIST 128 // distances in the window sample: at most ...
#endif
#ifndef TSW_SAMPLE_MIN
#define TSW_SAMPLE_MIN 48   // ... and at least (scratch permitting: see the sample size below)
#endif
#define TSW_MIN_POINTS 256  // below this the generic two-pass select is cheap anyway
#define TSW_REL 1e-5        // guard band of the approximate classification (see below)
#ifndef TSW_BM_U
#define TSW_BM_U 12        // samples of a base in flight per thread (base means; RNA's 40-sample bases: 8 -> 12 = -13 % of the phase, 16 spills)
#endif
#ifndef TSW_STEP
#define TSW_STEP 8          // partners per step of the pass over the pairs (their LDS loads go out together)
#endif
// The per-base means the fit needs (ts.compute_base_means = c_new_means, _c_helper.pyx:59-71,
// over the resolved boundaries) are computed here, for the <= 1000 sampled bases only: a thread
// sums its base's samples in order and divides once -- a read of 10 000 bases touches a tenth of
// its signal instead of all of it (the separate k_base_means pass of round 1: 2.7 ms, RNA 9 ms).
__global__ __launch_bounds__(SEL_NT, 6) void k_theil_sen(ReadState *rs, const DevParams *dp,
    const double *norm, const i64 *segs, const double *ref_means, i64 *samp_ind, double *scratch,
    double *scratch2)
{
    __shared__ alignas(16) BucketSmem sm; // (16: the sorted points of the pair pass are read as double2)
    __shared__ u32 s_ncand;
    __shared__ double s_win[2];
    __shared__ int s_win_ok;
    __shared__ double s_ev[MAX_TS_POINTS], s_md[MAX_TS_POINTS];
    ReadState &r = rs[blockIdx.x];
    if (r.status != TBA_OK) return;
    if (dp->o.skip_seq_scaling) return;
    const int tid = threadIdx.x;
    TBA_PHASE_T0(3);
    const double *mu = ref_means + r.ref_off;
    const double *x = norm + r.raw_off + r.read_start;
    const i64 *sg = segs + r.seg_off;
    auto base_mean = [&](i64 k) { // c_new_means: sequential sum, one divide
        const i64 a = sg[k], b = sg[k + 1];
        double acc = 0;
        // the adds are sequential (the reference's order), the loads are not: TSW_BM_U in flight (a
        // `load -> add` loop pays a memory round trip per sample of the base: 13 % of this kernel)
        for (i64 j = a; j < b; j += TSW_BM_U) {
            double t[TSW_BM_U];
#pragma unroll
            for (int u = 0; u < TSW_BM_U; u++) t[u] = x[j + u < b ? j + u : b - 1];
#pragma unroll
            for (int u = 0; u < TSW_BM_U; u++) if (j + u < b) acc += t[u];
        }
        return acc / (double)(b - a);
    };
    i64 n = r.B;
    if (n > MAX_TS_POINTS) {
        if (samp_ind == nullptr) { if (tid == 0) r.status = TBA_INTERNAL; return; }
        i64 *si = samp_ind + (i64)blockIdx.x * MAX_TS_POINTS;
        n = MAX_TS_POINTS;
        bool bad = false;
        // tba_opts.device_subsample: the indices are drawn here (k_prep_raw.h) and left in samp_ind
        const bool draw = dp->o.device_subsample != 0;
        const u64 key = subsample_key(dp->o.subsample_seed, dp->o.subsample_first_read + (i64)blockIdx.x);
        for (i64 i = tid; i < n; i += SEL_NT) {
            i64 k;
            if (draw) { k = keyed_perm(i, r.B, key); si[i] = k; }
            else k = si[i];
            if (k < 0 || k >= r.B) { bad = true; k = 0; }
            s_ev[i] = base_mean(k); s_md[i] = mu[k];
        }
        if (__syncthreads_or(bad)) { if (tid == 0) r.status = TBA_INTERNAL; return; }
    } else {
        for (i64 i = tid; i < n; i += SEL_NT) { s_ev[i] = base_mean(i); s_md[i] = mu[i]; }
        __syncthreads();
    }
    const i64 ns = n * (n - 1) / 2;
    if (ns <= 0) { if (tid == 0) r.status = TBA_INTERNAL; return; }
    TBA_PHASE(3, 0);
    // all n(n-1)/2 pairs by circular distance d: (i, (i+d) mod n) for d = 1..(n-1)/2, plus
    // (i, i + n/2) for i < n/2 when n is even; slope(i,j) == slope(j,i) bitwise.  The loops have
    // workgroup-uniform trip counts (the visitor ballots).
    auto slopes = [=](auto visit) {   // (closures by value: nothing here may pin a local to scratch)
        // a thread keeps point i and walks the distances four at a time (their LDS loads go out
        // together); dtop = dmax, plus the antipodal distance n/2 for even n (first half of the
        // points only)
        const int nn = (int)n, dmax = (nn - 1) / 2, dtop = dmax + ((nn & 1) ? 0 : 1);
        for (int i0 = 0; i0 < nn; i0 += SEL_NT) {
            const int i = i0 + tid;
            const bool okr = i < nn;
            const int ic = okr ? i : 0;
            const double ei = s_ev[ic], mi = s_md[ic];
            const int dlim = okr ? ((nn & 1) || i >= nn / 2 ? dmax : dtop) : 0; // my last distance
            for (int d0 = 1; d0 <= dtop; d0 += 4) {
                double ej[4], mj[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    int j = ic + d0 + u;
                    j = j >= nn ? j - nn : j;
                    j = j >= nn ? 0 : j; // d0 + u past dtop (never used)
                    ej[u] = s_ev[j]; mj[u] = s_md[j];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const double sl = (ei == ej[u]) ? 1000.0 : (mi - mj[u]) / (ei - ej[u]);
                    visit(sl, d0 + u <= dlim);
                }
            }
        }
    };
    // Median slope.  Generic form: two bucket-select passes over all n(n-1)/2 exact slopes
    // (slopes of a normalised read cluster around 1: [0.5, 1.5] is only the first bucket range,
    // any other distribution costs refinement passes, not correctness; k_select.h).
    //
    // Fast form (n >= TSW_MIN_POINTS), ONE pass over the pairs:
    //  1. a sample (TSW_SAMPLE_DIST evenly spread circular distances, approximate slopes) is
    //     histogrammed; the buckets at the sample quantiles 0.5 -/+ 4 sigma give a window
    //     [t1, t2) that holds the two middle ranks of ALL slopes with near certainty;
    //  2. every pair is classified against the window without a division (the points sorted by
    //     level, two compares per pair against the edges moved outwards by the relative guard
    //     band TSW_REL >> 2^-53: see the pass itself): safely below t1 -> counted; safely above
    //     t2 -> nothing; inside the window or within the guard band of an edge -> WHICH pair it
    //     is goes to a list in global scratch;
    //  3. the listed pairs (several thousand) are divided exactly, classified exactly, and the
    //     middle ranks are selected among the exact in-window slopes.
    // The count below t1 and the in-window multiset are exact, so the result is the reference's
    // np.median; whenever the window misses the middle ranks or the list overflows, the generic
    // form runs instead.
    double slope = 0;
    bool fast_done = false;
    // the listed pairs (8 bytes each: which pair, then its exact slope) in this read's slices of the csum and
    // score buffers (n_raw doubles each)
    const i64 cap1 = r.n_raw, cap2 = scratch2 != nullptr ? r.n_raw : 0;
    const i64 cap = cap1 + cap2 < 65536 ? cap1 + cap2 : 65536;
    if (n >= TSW_MIN_POINTS && scratch != nullptr && cap >= 4096) {
        double *cl1 = scratch + r.raw_off + blockIdx.x, *cl2 = scratch2 + r.raw_off;
        auto pair_at = [=](i64 k) { return k < cap1 ? cl1 + k : cl2 + (k - cap1); };
        const u32 cap_u = (u32)cap, cap1_u = (u32)(cap1 < 65536 ? cap1 : 65536);
        const int nn = (int)n, dmax = (nn - 1) / 2;
        // Sample size.  The window spans +-4 sigma of a sample quantile, so the list holds about
        // 4 ns / sqrt(sample) pairs.  Measured per workgroup: ~1.3 cycles per sample element,
        // ~15 per listed pair since round 6 (append, exact division, select; ~40 before) -- a flat
        // optimum around 48 k samples for 1000 points (profiles/r06_theil_sen_phases.txt); more
        // when the scratch is small (the list should stay under 60 % of it).
        int ds;
        {
            const double want = 4.0 * (double)ns / (0.6 * (double)cap);
            const double need = want * want / (double)nn;
            ds = need < (double)TSW_SAMPLE_MIN ? TSW_SAMPLE_MIN : (need > (double)TSW_SAMPLE_DIST ? TSW_SAMPLE_DIST : (int)need + 1);
            ds = ds < dmax ? ds : dmax;
        }
        const double glo = 0.5, gsc = (double)BS_NB / 1.0;
        for (int b = tid; b < BS_NB; b += SEL_NT) sm.hist[b] = 0;
        if (tid == 0) { s_ncand = 0; s_win_ok = 0; }
        __syncthreads();
        for (int t0 = 0; t0 < ds; t0++) { // sample: distance 1 + t0 * dmax / ds
            const int d = 1 + (int)(((i64)t0 * dmax) / ds);
            for (int i = tid; i < nn; i += SEL_NT) {
                int j = i + d; j = j >= nn ? j - nn : j;
                // the sample only steers the window: the approximate quotient is good enough
                // (all four LDS reads go out together: behind a test of ei == ej the levels were a
                // second, dependent round trip)
                const double ei = s_ev[i], ej = s_ev[j], mi = s_md[i], mj = s_md[j], b = ei - ej;
                double rr = __builtin_amdgcn_rcp(b);
                rr = __builtin_fma(__builtin_fma(-b, rr, 1.0), rr, rr);
                const double q = (mi - mj) * rr;
                const double sl = (b == 0.0) ? 1000.0 : q;
                atomicAdd(&sm.hist[bs_bucket(sl, glo, gsc)], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) { // wave 0: the buckets holding the sample ranks (0.5 -/+ dq) m
            const double m = (double)nn * ds;
            const double dq = 4.0 * sqrt(0.25 / m) + 0.0005;
            const i64 r1 = (i64)((0.5 - dq) * m), r2 = (i64)((0.5 + dq) * m);
            const int per = BS_NB / 64;
            i64 c = 0;
            for (int q = 0; q < per; q++) c += sm.hist[tid * per + q];
            i64 inc = c;
            for (int d = 1; d < 64; d <<= 1) {
                const i64 t = shfl_i64(inc, tid - d < 0 ? 0 : tid - d);
                if (tid >= d) inc += t;
            }
            // first bucket whose cumulative count exceeds the rank: exactly one lane owns it
            i64 acc = inc - c;
            int b1 = -1, b2 = -1;
            for (int q = 0; q < per; q++) {
                const i64 nx = acc + sm.hist[tid * per + q];
                if (acc <= r1 && r1 < nx) b1 = tid * per + q;
                if (acc <= r2 && r2 < nx) b2 = tid * per + q;
                acc = nx;
            }
            const u64 m1 = __ballot(b1 >= 0), m2 = __ballot(b2 >= 0);
            if (m1 && m2) {
                b1 = __shfl(b1, __ffsll((unsigned long long)m1) - 1, 64);
                b2 = __shfl(b2, __ffsll((unsigned long long)m2) - 1, 64);
                if (tid == 0 && b1 > 0 && b2 >= b1 && b2 < BS_NB - 1) {
                    s_win[0] = glo + (double)b1 / gsc;
                    s_win[1] = glo + (double)(b2 + 1) / gsc;
                    s_win_ok = 1;
                }
            }
        }
        __syncthreads();
        TBA_PHASE(3, 1);
        if (s_win_ok) {
            double A1, B2;
            {
                const double t1 = s_win[0], t2 = s_win[1]; // 0.5 < t1 < t2 < 1.5
                A1 = t1 - t1 * TSW_REL; B2 = t2 + t2 * TSW_REL;
            }
            i64 c_lo = 0;
            // Round 6: a pair costs two single-precision compares.  The points are sorted by (level, model
            // mean) into the histogram's LDS (free once the window is known; s_ev / s_md stay as they were),
            // and every point k gets u1[k] = m_k - A1 e_k and u2[k] = m_k - B2 e_k, rounded to float.  For
            // i < j in that order e_j >= e_i, so
            //    u1[j] < u1[i] - tau  =>  (m_j - m_i) - A1 (e_j - e_i) < 0  =>  e_j > e_i (equal levels come
            //                             in ascending m) and the pair's slope is below A1: counted;
            //    u2[j] > u2[i] + tau  =>  the slope is above B2, or e_j == e_i (the reference's max_slope =
            //                             1000): nothing;
            //    neither              =>  in the window or too close to tell: listed (a, b) from the double
            //                             points, divided exactly in step 3.
            // tau = 2^-20 (max |m| + 3 max |e|) is sixteen times the rounding of a float u (the double
            // arithmetic behind it is 2^-29 of that) -- against it twice plus the rounding of u[i] -/+ tau;
            // the computed slope is within 3 2^-53 of the real one and TSW_REL = 1e-5 covers that as before.
            // What tau adds to the list is the pairs within ~1e-5 / (e_j - e_i) of a window edge: a few.
            // A wavefront takes blocks of 64 rows i (a lane each) against chunks of 64 later points j: the
            // chunk is loaded once (a lane each), u[j] comes to the compares through v_readlane (an SGPR
            // operand: as 64-lane LDS broadcasts the loads alone took as long as the old pass), the lane
            // masks are combined and counted on the scalar unit.  Row blocks are handed out in mirrored
            // pairs (block b and the last-but-b share the chunks: n + 64 partners together, every
            // wavefront the same).  Before: two differences, a sign fold, two products, three double
            // compares per pair and 1.25 LDS loads of 16 bytes -- 52 % of this kernel
            // (profiles/r05_theil_sen_phases.txt).
            double2 *sp = (double2 *)sm.raw8;                       // sorted (e, m) ...
            float2 *sf = (float2 *)(sp + 1024);                     // ... and their (u1, u2)
            int np2 = 256;
            while (np2 < nn) np2 <<= 1;
            for (int k = tid; k < np2; k += SEL_NT) {
                double2 v;
                v.x = k < nn ? s_ev[k] : INFINITY; v.y = k < nn ? s_md[k] : INFINITY;
                sp[k] = v;
            }
            __syncthreads();
            for (int kk = 2; kk <= np2; kk <<= 1)
                for (int jj = kk >> 1; jj >= 1; jj >>= 1) {
                    for (int t = tid; t < np2 / 2; t += SEL_NT) {
                        const int i = 2 * t - (t & (jj - 1)), p = i + jj;
                        const double2 x = sp[i], y = sp[p];
                        const bool gt = x.x > y.x || (x.x == y.x && x.y > y.y);
                        if (gt == ((i & kk) == 0)) { sp[i] = y; sp[p] = x; }
                    }
                    // thread t's pair lies in the 128 points [128 (t / 64), +128) while jj <= 64: a wavefront
                    // works on its own points then, and its LDS operations complete in order -- the
                    // workgroup meets only around the steps that reach further (14 of the 55 for 1024 points)
                    const int jnext = jj > 1 ? jj >> 1 : (kk < np2 ? kk : 1 << 30);
                    if (jj > 64 || jnext > 64 || np2 / 2 > SEL_NT) __syncthreads();
                    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
                }
            double mx = 0;
            const int nblk = (nn + 63) / 64;
            for (int k = tid; k < 64 * nblk; k += SEL_NT) {
                float2 u;
                u.x = u.y = INFINITY;                                // past the end: never below, never listed
                if (k < nn) {
                    const double2 v = sp[k];
                    u.x = (float)(v.y - A1 * v.x); u.y = (float)(v.y - B2 * v.x);
                    mx = fmax(mx, fabs(v.y) + 3.0 * fabs(v.x));
                }
                sf[k] = u;
            }
            for (int m = 32; m >= 1; m >>= 1) mx = fmax(mx, shfl_f64(mx, (tid & 63) ^ m));
            if ((tid & 63) == 0) sm.redd[tid >> 6] = mx;
            __syncthreads();
            for (int w = 0; w < SEL_NT / 64; w++) mx = fmax(mx, sm.redd[w]);
            const float tau = (float)(0x1p-20 * mx);
            TBA_PHASE(3, 6);
            const int lane = tid & 63, wv = tid >> 6;
            i64 wave_lo = 0;
            for (int bk = wv; bk < (nblk + 1) / 2; bk += SEL_NT / 64) {
                const int bA = bk, bB = nblk - 1 - bk;              // bA <= bB; every row of bA exists
                const int rowA = 64 * bA + lane, rowB = 64 * bB + lane, rcB = rowB < nn ? rowB : nn - 1;
                const u64 okmB = __ballot(rowB < nn);
                const float2 fa = sf[rowA], fb = sf[rcB];
                const float cA1 = fa.x - tau, cA2 = fa.y + tau, cB1 = fb.x - tau, cB2 = fb.y + tau;
                // one chunk of 64 partners.  MODE 0: the rows of bA against their own block (j > i only);
                // 1: bA against a later chunk; 2: bA against bB's chunk, and bB against its own; 3: both
                auto chunk = [&](const int jc, auto mode_tag) __attribute__((always_inline)) {
                    constexpr int MODE = decltype(mode_tag)::value;
                    constexpr int NS = MODE >= 2 ? 2 : 1;
                    const float2 pv = sf[jc + lane];
                    // The listed pairs -- 1-2 % -- are not appended where they are found (a wave-uniform branch
                    // for one lane's store: ~100 issue cycles each, 3/4 of this pass when it was done so): a
                    // lane shifts the listed bit of every partner into a mask of its own (v_addc: one
                    // instruction), and after 32 partners all lanes append what they have together.
                    for (int h = 0; h < 64; h += 32) {
                        u32 lm[NS];
#pragma unroll
                        for (int q = 0; q < NS; q++) lm[q] = 0;
                        for (int g = h; g < h + 32; g += TSW_STEP) {
#pragma unroll
                            for (int u = 0; u < TSW_STEP; u++) {
                                const int jj = g + u;
                                const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pv.x), jj));
                                const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pv.y), jj));
                                const u64 tri = (1ull << jj) - 1ull;  // the rows of the chunk's own block before j
#pragma unroll
                                for (int q = 0; q < NS; q++) {
                                    const u64 lo = __ballot(s1 < (q ? cB1 : cA1)), hi = __ballot(s2 > (q ? cB2 : cA2));
                                    u64 ok = q ? okmB : ~0ull;
                                    if (MODE == (q ? 2 : 0)) ok &= tri;
                                    wave_lo += __popcll(lo & ok);      // safely below the window
                                    const u64 cm = ok & ~(lo | hi);    // inside, too close to an edge, NaN
                                    u64 carry_out;
                                    asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(lm[q]), "=s"(carry_out) : "s"(cm));
                                }
                            }
                        }
#pragma unroll
                        for (int q = 0; q < NS; q++) {
                            u32 m = lm[q];                             // bit 31 - k: partner jc + h + k
                            if (__ballot(m != 0) == 0) continue;
                            // where a lane's pairs go: the counts before it (bit plane by bit plane: a count is
                            // small), one counter bump per wavefront
                            const u32 cnt = (u32)__popc(m);
                            const u64 before = (1ull << lane) - 1ull;
                            u32 pos = 0, tot = 0;
                            for (int bp = 0; bp < 6; bp++) {
                                const u64 pl = __ballot((cnt >> bp) & 1u);
                                pos += (u32)__popcll(pl & before) << bp;
                                tot += (u32)__popcll(pl) << bp;
                                if (__ballot(cnt >> (bp + 1)) == 0) break;
                            }
                            u32 base = 0;
                            if (lane == 0) base = atomicAdd(&s_ncand, tot);
                            pos += (u32)__builtin_amdgcn_readfirstlane((int)base);
                            const i64 hi16 = (i64)((q ? rcB : rowA) << 16) | (jc + h);
                            while (m) {
                                const int k = __clz((int)m);
                                m &= ~(0x80000000u >> k);
                                // only WHICH pair (step 3 takes the points out of LDS again)
                                if (pos < cap_u) *(i64 *)(pos < cap1_u ? cl1 + pos : cl2 + (pos - cap1_u)) = hi16 + k;
                                pos++;
                            }
                        }
                    }
                };
                chunk(64 * bA, IntTag<0>{});
                for (int c = bA + 1; c < bB; c++) chunk(64 * c, IntTag<1>{});
                if (bB != bA) {
                    chunk(64 * bB, IntTag<2>{});
                    for (int c = bB + 1; c < nblk; c++) chunk(64 * c, IntTag<3>{});
                } else                                                // (the middle block of an odd number)
                    for (int c = bA + 1; c < nblk; c++) chunk(64 * c, IntTag<1>{});
            }
            c_lo = lane == 0 ? wave_lo : 0;
            c_lo = block_sum_i64(c_lo, &sm.rad);
            __threadfence_block();
            __syncthreads();
            TBA_PHASE(3, 2);
            const i64 n_c = s_ncand;
            if (n_c <= cap) {
                // exact slopes of the listed pairs; in-window ones stay (in place of their a),
                // the rest becomes +inf and the ones below t1 are counted
                // (the window edges come out of LDS again: kept in registers across the pass over the
                // pairs they were the four registers the kernel spilled)
                const double t1 = s_win[0], t2 = s_win[1];
                i64 lo_more = 0, inw = 0;
                for (i64 k = tid; k < n_c; k += SEL_NT) {
                    double *pr = pair_at(k);
                    const i64 ij = *(const i64 *)pr;
                    const double2 pi = sp[ij >> 16], pj = sp[ij & 0xffff];
                    const double a = pi.y - pj.y, b = pi.x - pj.x;
                    const double sl = b == 0.0 ? 1000.0 : a / b;               // (max_slope: _c_helper.pyx:371)
                    const bool below = sl < t1, in = sl >= t1 && sl < t2;
                    lo_more += below; inw += in;
                    pr[0] = in ? sl : INFINITY;
                }
                c_lo += block_sum_i64(lo_more, &sm.rad);
                inw = block_sum_i64(inw, &sm.rad);
                __threadfence_block();
                __syncthreads();
                // (workgroup-uniform: held in SGPRs across the selection -- as vector registers they were the kernel's one spill)
                auto uni = [](i64 v) {
                    return (i64)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)((u64)v >> 32)) << 32) |
                                 (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u64)v));
                };
                const i64 k_lo = uni((ns - 1) / 2 - c_lo), k_hi = uni(ns / 2 - c_lo);
                if (k_lo >= 0 && k_hi < inw) {
                    double a;                                              // (inlined by request, as the medians below)
                    [[clang::always_inline]] a = block_kth([=](i64 k) { return pair_at(k)[0]; }, n_c, k_lo, t1, t2, &sm);
                    const int found = sm.found;
                    const double nxt = sm.next;
                    __syncthreads();
                    double b = a;
                    if (k_hi != k_lo) {
                        if (found & 2) b = nxt; // the next order statistic came with the first one
                        else { [[clang::always_inline]] b = block_kth([=](i64 k) { return pair_at(k)[0]; }, n_c, k_hi, t1, t2, &sm); __syncthreads(); }
                    }
                    slope = (ns & 1) ? a : (a + b) / 2.0;
                    fast_done = true;
                }
            }
        }
    }
    TBA_PHASE(3, 3);
    if (!fast_done) { [[clang::always_inline]] slope = block_median_fe(slopes, ns, 0.5, 1.5, &sm); } // (inlined by request, as the intercepts below)
    TBA_PHASE(3, 4);
    // intercepts of a normalised read sit within a unit or so of 0 (first bucket range only)
    // (inlined by request: past the kernel's present size the compiler made it a call, with the closure in scratch)
    double inter;
    [[clang::always_inline]] inter = block_median_fast([=](i64 i) { return s_md[i] - (slope * s_ev[i]); }, n, -2.0,
                                                       2.0, &sm);
    TBA_PHASE(3, 5);
    TBA_PHASE_END(3);
    if (tid == 0) {
        if (slope == 0) { r.status = TBA_RESCALE_FAIL; return; }
        double scale_corr = 1 / slope;
        double shift_corr = -inter / slope;
        r.ts[0] = r.shift + (shift_corr * r.scale);
        r.ts[1] = r.scale * scale_corr;
        r.ts[2] = shift_corr;
        r.ts[3] = scale_corr;
        r.shift = r.ts[0];
        r.scale = r.ts[1];
        r.changed = (fabs(shift_corr) > SHIFT_CHANGE_THRESH ||
                     fabs(scale_corr - 1) > SCALE_CHANGE_THRESH) ? 1 : 0;
    }
}

// c_new_mean_stds (_c_helper.pyx:38-57) over the final signal and boundaries of every read of
// the batch: what write_new_fast5_group (tombo_helper.py:2341-2362) stores per base as
// norm_mean / norm_stdev.  Same wave-cooperative staging as k_rescale_absz; the variance loop runs
// over the staged samples again (population sd around the segment mean).  grid: (blocks, reads)
// the final signal of a read: the materialised norm_out, or (skip_norm_out batches) the signal of
// segment_signal rescaled on the fly with k_rescale_absz's expression (same operation, same bits)
struct FinalSignal {
    const double *p;
    bool rescale;
    double ca, cb;
    __device__ __forceinline__ double operator[](i64 i) const { return rescale ? (p[i] - ca) / cb : p[i]; }
};
__device__ __forceinline__ void sig_pair(FinalSignal x, i64 i, double &a, double &b)
{
    ld2(x.p + i, a, b);
    if (x.rescale) { a = (a - x.ca) / x.cb; b = (b - x.ca) / x.cb; }
}
__global__ __launch_bounds__(256) void k_base_stats(const ReadState *rs, const DevParams *dp,
    const double *norm_out, const double *norm, const i64 *segs, double *means, double *stds)
{
    const ReadState &r = rs[blockIdx.y];
    if (r.status != TBA_OK) return;
    const FinalSignal x = norm_out != nullptr
        ? FinalSignal{norm_out + r.raw_off, false, 0.0, 1.0}
        : FinalSignal{norm + r.raw_off + r.read_start, dp->o.skip_seq_scaling == 0, r.ts[2], r.ts[3]};
    const i64 *sg = segs + r.seg_off;
    constexpr int CAP = 768;
    __shared__ double s_seg[4 * CAP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *lds = s_seg + wave * CAP;
    const i64 n_segs = r.B;
    int gs = 64; // segments per wave step (fewer when they are long, see wave_segment_sums)
    if (n_segs > 0) {
        const double mean_len = (double)(sg[n_segs] - sg[0]) / (double)n_segs;
        while (gs > 4 && (double)gs * mean_len * 1.3 > (double)CAP) gs >>= 1;
    }
    for (i64 g = (i64)blockIdx.x * 4 + wave; g * gs < n_segs; g += (i64)gridDim.x * 4) {
        const i64 i = g * gs + lane;
        const bool ok = lane < gs && i < n_segs;
        const i64 i_end = g * gs + gs < n_segs ? g * gs + gs : n_segs;
        const i64 a = sg[ok ? i : i_end], b = sg[ok ? i + 1 : i_end];
        const i64 lo = sg[g * gs], hi = sg[i_end];
        const i64 span = hi - lo;
        const bool staged = span <= CAP;
        if (staged) {
            __builtin_amdgcn_wave_barrier();
            wave_stage<CAP / 128>(x, lo, span, lds, nullptr, [](double xv) { return xv; });
            __builtin_amdgcn_wave_barrier();
        }
        const double len = (double)(b - a);
        double s = 0, v = 0, m;
        if (staged) { // (an LDS and a global pointer must not share one variable: flat apertures)
            s = seq_sum_lds(lds, a - lo, b - lo);
            m = s / len;
            for (i64 j = a - lo; j < b - lo; j++) { const double d = lds[j] - m; v += d * d; }
        } else {
            for (i64 j = a; j < b; j++) s += x[j];
            m = s / len;
            for (i64 j = a; j < b; j++) { const double d = x[j] - m; v += d * d; }
        }
        if (ok) { means[r.ref_off + i] = m; stds[r.ref_off + i] = sqrt(v / len); }
    }
}

// norm_out[i] = (norm[read_start + i] - shift_corr) / scale_corr (resquiggle.py:1190; a plain
// trim copy when sequence rescaling is skipped) and, in the same pass, c_new_means over the
// final signal -> |z| per base (ts.get_read_seg_score, tombo_stats.py:2327-2338): a wavefront
// takes a group of consecutive bases, rescales the samples they span (coalesced read of norm, coalesced write of norm_out),
// keeps them in its LDS slice and sums every base from there.  The bases tile the trimmed
// signal exactly (segs[0] = 0, segs[B] = norm_len), so every sample is written once.
// grid: (blocks, reads)
#ifndef TBA_RSZ_WAVES
#define TBA_RSZ_WAVES 5
#endif
template <bool WRITE> // WRITE = false: skip_norm_out batch, the final signal is not materialised
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TBA_RSZ_WAVES))) void k_rescale_absz(const ReadState *rs, const DevParams *dp,
    const double *norm, double *norm_out, const i64 *segs, const double *ref_means,
    const double *ref_sds, double *absz)
{
    const ReadState &r = rs[blockIdx.y];
    if (r.status != TBA_OK) return;
    const double *x = norm + r.raw_off + r.read_start;
    double *y = WRITE ? norm_out + r.raw_off : nullptr;
    const i64 *sg = segs + r.seg_off;
    const bool skip = dp->o.skip_seq_scaling != 0;
    const double ca = r.ts[2], cb = r.ts[3];
    constexpr int CAP = 768;
    __shared__ double s_seg[4 * CAP];
    const int wave = threadIdx.x >> 6;
    const i64 n_segs = r.B;
    if (n_segs <= 0) return;
    const double *rm = ref_means + r.ref_off, *rsd = ref_sds + r.ref_off;
    double *az = absz + r.ref_off;
    struct Ref { double m, sd; };
    wave_segment_sums<CAP>(x, sg, n_segs, (i64)blockIdx.x * 4 + wave, (i64)gridDim.x * 4, s_seg + wave * CAP,
        WRITE ? y : nullptr,
        [&](double xv) { return skip ? xv : (xv - ca) / cb; },   // resquiggle.py:1190
        [&](i64 i) { return Ref{rm[i], rsd[i]}; },
        [&](i64 i, double s, i64 len, Ref ref) {
            const double m = s / (double)len;
            az[i] = fabs((m - ref.m) / ref.sd);
        },
        (double)(sg[n_segs] - sg[0]) / (double)n_segs);
}

// ts.get_read_seg_score (tombo_stats.py:2327-2338): np.mean in numpy's summation order
// (np.add.reduce: 8192-element chunks accumulated left to right, each summed pairwise: a size m
// splits at m/2 rounded down to a multiple of 8 until a piece has <= 128 elements, a leaf keeps 8
// strided partial sums -- np_pairwise_leaf).  One wavefront per read: lane 0 lists the leaves of
// a chunk (pre-order walk, left first: at most 127), the lanes sum a leaf each, lane 0 adds the
// leaf sums up in the recursion's order.  (One thread per read walked 10 000 values through a
// chain of dependent loads: 1.1 ms per 10 000 reads.)
__global__ __launch_bounds__(64) void k_final_score(ReadState *rs, i64 n_reads, const double *absz)
{
    __shared__ int l_start[128], l_len[128], s_nleaf;
    __shared__ double l_sum[128];
    __shared__ int st_a[16], st_n[16], st_st[16];
    __shared__ double st_v[16];
    const i64 ri = blockIdx.x;
    if (ri >= n_reads) return;
    ReadState &r = rs[ri];
    if (r.status != TBA_OK) return;
    const int lane = threadIdx.x;
    const double *a = absz + r.ref_off;
    const i64 n = r.B;
    double acc = 0.0;
    for (i64 c0 = 0; c0 < n; c0 += 8192) {
        const int m = (int)(n - c0 < 8192 ? n - c0 : 8192);
        if (lane == 0) {
            int sp = 0, nl = 0;
            st_a[0] = 0; st_n[0] = m;
            while (sp >= 0) {
                const int s0 = st_a[sp], len = st_n[sp];
                sp--;
                if (len <= 128) { l_start[nl] = s0; l_len[nl] = len; nl++; continue; }
                int n2 = len / 2;
                n2 -= n2 % 8;
                st_a[sp + 1] = s0 + n2; st_n[sp + 1] = len - n2; // right, visited after ...
                st_a[sp + 2] = s0; st_n[sp + 2] = n2;            // ... the left half
                sp += 2;
            }
            s_nleaf = nl;
        }
        __syncthreads();
        for (int k = lane; k < s_nleaf; k += 64) l_sum[k] = np_pairwise_leaf(a + c0 + l_start[k], l_len[k]);
        __syncthreads();
        if (lane == 0) { // the recursion again, leaf values from l_sum in visiting order
            int sp = 0, next = 0;
            double ret = 0;
            st_n[0] = m; st_st[0] = 0;
            while (sp >= 0) {
                const int len = st_n[sp];
                if (len <= 128) { ret = l_sum[next++]; sp--; continue; }
                int n2 = len / 2;
                n2 -= n2 % 8;
                if (st_st[sp] == 0) { st_st[sp] = 1; st_n[sp + 1] = n2; st_st[sp + 1] = 0; sp++; }
                else if (st_st[sp] == 1) { st_v[sp] = ret; st_st[sp] = 2; st_n[sp + 1] = len - n2; st_st[sp + 1] = 0; sp++; }
                else { ret = st_v[sp] + ret; sp--; }
            }
            acc += ret;
        }
        __syncthreads();
    }
    if (lane == 0) r.score = acc / (double)r.B;
}
