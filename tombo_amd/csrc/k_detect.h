// k_detect.h -- event detection without the score array: c_valid_cpts_w_cap (_c_helper.pyx:89-120)
// as two kernels that never write the change-point scores to memory.
//
// Round 1-3 form (k_segment.h): k_cumsum_scores writes one float64 score per sample (8 S bytes),
// k_peaks reads them ~2.3 times and a state byte per sample on top: 2.9 MB of the 10.9 MB a 10 kb
// read moved through HBM.  Here:
//   k_detect  the cumulative-sum pipeline of k_cumsum_scores (lane per read, 20 reads per
//             workgroup, 128-sample LDS tiles, left-to-right float64 adds) with two more wavefronts
//             that turn every scanned tile into scores IN REGISTERS and resolve the uncapped greedy
//             on them as the bit-sliced fixed point of k_peaks, streaming: a lane holds a 32-position
//             word of one read, five words per read and step (the last word of the previous step +
//             the 128 new positions); whatever is decided is final (decisions only ever use decided
//             neighbours), positions whose dependency chain reaches the not-yet-scanned future stay
//             open until the next step.  Only the TAKEN positions leave the CU: (score, position),
//             densely, in position order, ~S/4 of them (3 S bytes).
//   k_pick    one workgroup per read over that list: the score of the num_cpts-th best taken
//             position by exact selection, the picks at or above it in position order (ties at the
//             threshold fall to the higher index), the reference's early-stop
//             error.
// What a read cannot get here is left to the old pair of kernels, which run on flagged reads only
// (ReadState.ed_flag): a dependency chain longer than a word (32 positions of strictly rising
// priority), the early-stop error too close to call from the taken list, long reads (k_long.h),
// and every parameter set outside 2 * running_stat_width <= 32, min_obs_per_base = 3.
// Same arithmetic (the sums and |2 c[k+w] - c[k] - c[k+2w]| in the reference's order), same
// priority rule, exact selection: valid_cpts is bit-identical to the old path and to the oracle.
#pragma once
#include "k_segment.h"

#define DT_READS 20
#define DT_W2MAX 32                  // 2 * running_stat_width the fused form takes
#define DT_HSTRIDE (DT_W2MAX + 1)
#define DT_CSTRIDE 33
#define DT_MAX_ROUNDS 64

// bit i <- bit i + d of (next:cur);  bit i <- bit i - d of (cur:prev);  0 < d < 32
__device__ __forceinline__ u32 dt_down(u32 cur, u32 next, int d) { return __builtin_amdgcn_alignbit(next, cur, d); }
__device__ __forceinline__ u32 dt_up(u32 cur, u32 prev, int d) { return __builtin_amdgcn_alignbit(cur, prev, 32 - d); }
__device__ __forceinline__ u32 dt_lane_below(u32 x) // lane l <- lane l - 1 (lane 0: 0)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ u32 dt_lane_above(u32 x) // lane l <- lane l + 1 (lane 63: 0)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, false);
}

template <int R>
__global__ __launch_bounds__(256, 2) void k_detect(ReadState *rs, i64 n_reads, const DevParams *dp,
    const double *__restrict__ norm, double *__restrict__ dense, double *__restrict__ posbuf)
{
    static_assert(R >= 1 && R <= 8, "exclusion radius");
    __shared__ double tile[3][DT_READS * CS_STRIDE];
    __shared__ double halo[DT_READS * DT_HSTRIDE];   // row q: the 2w sums before the tile being scored
    __shared__ double carry[2][DT_READS * DT_CSTRIDE]; // scores of the last word of the previous step
    __shared__ i64 s_off[DT_READS], s_n[DT_READS];
    __shared__ u32 s_cnt[DT_READS];
    __shared__ int s_bad[DT_READS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const i64 r0 = (i64)blockIdx.x * DT_READS;
    const int w = (int)dp->p.running_stat_width, w2 = 2 * w;
    if (tid < DT_READS) {
        const i64 ri = r0 + tid;
        const bool ok = ri < n_reads && rs[ri].status == TBA_OK;
        const bool live = ok && !rs[ri].is_long;
        s_off[tid] = live ? rs[ri].raw_off : 0;
        s_n[tid] = live ? rs[ri].n_raw : 0;
        s_cnt[tid] = 0;
        s_bad[tid] = 0;
        if (ok && !live) { rs[ri].ed_flag = 1; rs[ri].n_taken = 0; } // long read: k_long.h + k_peaks
    }
    for (int k = tid; k < DT_READS * DT_HSTRIDE; k += 256) halo[k] = 0.0; // c[0] = 0, nothing before it
    __syncthreads();
    i64 n_max = 0;
    for (int q = 0; q < DT_READS; q++) n_max = s_n[q] > n_max ? s_n[q] : n_max;
    const i64 n_steps = (n_max + CS_CHUNK - 1) / CS_CHUNK;

    // ---- wave 1: loader, a row (read) per unit, two samples per lane and access
    double pa[DT_READS], pb[DT_READS];
    auto fetch = [&](i64 chunk) {
#pragma unroll
        for (int u = 0; u < DT_READS; u++) {
            const i64 k = chunk * CS_CHUNK + 2 * lane, n = s_n[u];
            const double *p = norm + s_off[u] + k;
            if (k + 1 < n) ld2(p, pa[u], pb[u]);
            else { pa[u] = k < n ? p[0] : 0.0; pb[u] = 0.0; }
        }
    };
    auto drop = [&](double *t) {
#pragma unroll
        for (int u = 0; u < DT_READS; u++) {
            t[u * CS_STRIDE + 2 * lane] = pa[u];
            t[u * CS_STRIDE + 2 * lane + 1] = pb[u];
        }
    };
    // (fetch and drop of a tile sit inside ONE step, in the loader's branch: registers carried across
    // the steps would be held through the greedy branch as well, and the kernel has none to spare)
    if (wave == 1) { fetch(0); drop(tile[0]); }
    __syncthreads();

    // ---- wave 0: the scan (lane = read)
    const i64 my_n = lane < DT_READS ? s_n[lane < DT_READS ? lane : 0] : 0;
    double acc = 0.0;
    // ---- waves 2, 3: the greedy (lane = one 32-slot word of a read; ten reads per wavefront)
    const int gq = lane / 5, h = lane - 5 * gq;      // read of this wave, word of the read
    const bool glane = wave >= 2 && lane < 50;
    const int q = (wave - 2) * 10 + gq;               // read of the workgroup (greedy waves)
    const i64 gn = glane ? s_n[glane ? q : 0] : 0;    // its length
    const i64 gri = r0 + (glane ? q : 0);
    double *dn = dense + (glane ? s_off[q] + gri : 0);
    i32 *pn = (i32 *)(posbuf + (glane ? s_off[q] : 0));
    u32 prevT = 0, prev_emitted = 0, prevX[R + 1];
#pragma unroll
    for (int d = 0; d <= R; d++) prevX[d] = 0;
    int cb = 0;

    for (i64 i = 0; i <= n_steps + 1; i++) {
        if (wave == 0) {
            if (i < n_steps && lane < DT_READS) {
                double *row = tile[i % 3] + lane * CS_STRIDE;
                const i64 left = my_n - i * CS_CHUNK;
                if (left >= CS_CHUNK) {
#pragma unroll 16
                    for (int k = 0; k < CS_CHUNK; k++) { acc = acc + row[k]; row[k] = acc; }
                } else {
                    for (int k = 0; k < CS_CHUNK; k++)
                        if (k < left) { acc = acc + row[k]; row[k] = acc; }
                }
            }
        } else if (wave == 1) {
            if (i + 1 < n_steps) { fetch(i + 1); drop(tile[(i + 1) % 3]); }
        } else if (i >= 1) {
            // tile j = i - 1: column t holds c[jC + 1 + t]; slot s = jC + t is the score whose window
            // ends there: position k = s + 1 - 2w, |2 c[k+w] - c[k] - c[k+2w]| (pyx:94-98) with
            // c[k+2w] = column t, c[k+w] = column t - w, c[k] = column t - 2w (negative: the halo)
            const i64 j = i - 1;
            const double *trow = tile[j % 3] + (glane ? q : 0) * CS_STRIDE;
            const double *hrow = halo + (glane ? q : 0) * DT_HSTRIDE;
            const double *crow = carry[cb] + (glane ? q : 0) * DT_CSTRIDE;
            double *cnext = carry[cb ^ 1] + (glane ? q : 0) * DT_CSTRIDE;
            const i64 S0 = j * CS_CHUNK;
            const i64 slot0 = S0 - 32 + 32 * h;       // first slot of my word
            const int col0 = 32 * (h - 1);            // its column in the tile (h >= 1)
            auto valid = [&](i64 s) { return glane && s >= w2 - 1 && s <= gn - 1; };
            auto c_at = [&](int col) { return col >= 0 ? trow[col] : hrow[w2 + col]; };
            auto score_at = [&](int t) { // slot of column t of the tile
                const double cc = c_at(t), cbv = c_at(t - w), ca = c_at(t - w2);
                return fabs(((2 * cbv) - ca) - cc);
            };
            u32 V = 0, G[R + 1];
#pragma unroll
            for (int d = 0; d <= R; d++) G[d] = 0;
            double hist[R];                           // the last R scores
            double first[R], last[R];
#pragma unroll
            for (int d = 0; d < R; d++) { hist[d] = 0.0; first[d] = 0.0; last[d] = 0.0; }
#pragma unroll 1
            for (int t8 = 0; t8 < 32; t8 += 8) {      // (eight positions' LDS reads in flight, not 32)
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int t = t8 + u;
                    const double s = h == 0 ? crow[t] : score_at(col0 + t);
                    const bool vt = valid(slot0 + t);
                    V |= (vt ? 1u : 0u) << t;
                    if (h == 4) cnext[t] = s;
#pragma unroll
                    for (int d = 1; d <= R; d++) {
                        // pair (t - d, t): "the neighbour at +d outranks me" for position t - d
                        // (ties fall to the higher index: >=)
                        const bool b = t - d >= 0 && vt && ((V >> ((t - d) & 31)) & 1u) && s >= hist[d - 1];
                        G[d] |= (b ? 1u : 0u) << ((t - d) & 31);
                    }
#pragma unroll
                    for (int d = R - 1; d >= 1; d--) hist[d] = hist[d - 1];
                    hist[0] = s;
#pragma unroll
                    for (int d = 0; d < R; d++) {
                        if (t == d) first[d] = s;
                        if (t == 32 - R + d) last[d] = s;
                    }
                }
            }
            // pairs into the next word: its first R scores come from the lane above; the future
            // (beyond the last word) counts as outranking wherever it exists
#pragma unroll
            for (int k = 0; k < R; k++) {
                const double bs = shfl_f64(first[k], (lane + 1) & 63);
                const bool bv = valid(slot0 + 32 + k);
#pragma unroll
                for (int d = k + 1; d <= R; d++) {
                    const int tp = 32 + k - d;        // my position of the pair, 32 - d .. 31
                    const bool b = bv && ((V >> tp) & 1u) && (h == 4 || bs >= last[tp - (32 - R)]);
                    G[d] |= (b ? 1u : 0u) << tp;
                }
            }
            // neighbour words: inside the read by DPP, at its ends the carried context / the future
            u32 vfut = 0;                             // validity of the 32 slots after the last word
            {
                const i64 f0 = S0 + 128;
                const i64 lo_s = w2 - 1 > f0 ? w2 - 1 - f0 : 0, hi_s = gn - 1 - f0; // valid: lo_s <= t <= hi_s
                if (glane && hi_s >= lo_s && lo_s < 32) {
                    const u32 up_to = hi_s >= 31 ? ~0u : ((2u << hi_s) - 1u);
                    vfut = up_to & (~0u << lo_s);
                }
            }
            const u32 va = dt_lane_above(V);          // (DPP outside the select: a lane masked off by a
            const u32 Vn = h == 4 ? vfut : va;        // divergent branch is no source for its neighbour)
            u32 X[R + 1], Hm[R + 1];
#pragma unroll
            for (int d = 1; d <= R; d++) {
                const u32 pv = V & dt_down(V, Vn, d);  // both ends of the pair (p, p + d) valid
                X[d] = pv & ~G[d];                     // p outranks p + d
                const u32 xb = dt_lane_below(X[d]);
                Hm[d] = V & dt_up(X[d], h == 0 ? prevX[d] : xb, d); // the neighbour at -d outranks me
            }
            u32 T = 0, S = 0, U = V;
            for (int round = 0; round < DT_MAX_ROUNDS; round++) {
                const u32 ta = dt_lane_above(T), tb = dt_lane_below(T);
                const u32 ua = dt_lane_above(U), ub = dt_lane_below(U);
                const u32 Tn = h == 4 ? 0u : ta, Tp = h == 0 ? prevT : tb;
                const u32 Un = h == 4 ? vfut : ua, Up = h == 0 ? 0u : ub;
                u32 at = 0, au = 0;
#pragma unroll
                for (int d = 1; d <= R; d++) {
                    at |= (G[d] & dt_down(T, Tn, d)) | (Hm[d] & dt_up(T, Tp, d));
                    au |= (G[d] & dt_down(U, Un, d)) | (Hm[d] & dt_up(U, Up, d));
                }
                const u32 nS = U & at, nT = U & ~at & ~au;
                T |= nT; S |= nS; U &= ~(nT | nS);
                if (__ballot((nS | nT) != 0) == 0) break;
            }
            // words 0..3 are final now; an open position there means a chain longer than a word
            if (glane && h < 4 && U != 0) s_bad[q] = 1;
            // emission, in position order: what word 0 decided late, words 1..3, and what word 4
            // has decided already (final as well; remembered, so that it is not emitted again)
            const u32 E = h == 0 ? T & ~prev_emitted : T;
            const int ce = glane ? __popc(E) : 0;
            int inc = ce;
#pragma unroll
            for (int dd = 1; dd < 5; dd <<= 1) {       // inclusive prefix over the 5 lanes of the read
                const int t = __shfl_up(inc, dd, 64);
                if (h >= dd) inc += t;
            }
            const int tot = __shfl(inc, lane - h + 4, 64); // (lanes >= 50: garbage, unused)
            u32 base = 0;
            if (glane) base = s_cnt[q];
            if (glane && h == 0) s_cnt[q] = base + (u32)tot;
            u32 o = base + (u32)(inc - ce);
            for (u32 m = glane ? E : 0u; m != 0; m &= m - 1u) {
                const int t = __ffs((int)m) - 1;
                const double s = h == 0 ? crow[t] : score_at(col0 + t);
                dn[o] = s;
                pn[o] = (i32)(slot0 + t - (w2 - 1));
                o++;
            }
            // context of the next step: its word 0 is this step's word 4, below it word 3
            prevT = (u32)__shfl((int)T, (lane + 3) & 63, 64);
#pragma unroll
            for (int d = 1; d <= R; d++) prevX[d] = (u32)__shfl((int)X[d], (lane + 3) & 63, 64);
            prev_emitted = (u32)__shfl((int)T, (lane + 4) & 63, 64);
            // the next tile's halo: the last 2w sums of this one
            if (glane && h == 4 && j < n_steps)
                for (int k = 0; k < w2; k++) halo[q * DT_HSTRIDE + k] = trow[CS_CHUNK - w2 + k];
            cb ^= 1;
        }
        __syncthreads();
    }
    if (wave >= 2 && glane && h == 0 && gn > 0) {
        ReadState &r = rs[gri];
        r.n_taken = (i64)s_cnt[q];
        r.ed_flag = s_bad[q];
    }
}

// The cap: keep the num_cpts best taken positions.  dense / posbuf: the taken (score, position)
// list k_detect left, ascending in position.  One workgroup per read.
__global__ __launch_bounds__(SEL_NT, 4) void k_pick(ReadState *rs, const DevParams *dp,
    const double *dense, const double *posbuf, i64 *valid_cpts)
{
    __shared__ BucketSmem sm;
    __shared__ i64 s_w[SEL_NT / 64];
    __shared__ i32 s_tie[2048];
    __shared__ u32 s_ntie;
    ReadState &r = rs[blockIdx.x];
    if (r.status != TBA_OK || r.ed_flag) return;
    const int tid = threadIdx.x;
    const i64 w = dp->p.running_stat_width;
    const i64 ns = r.n_raw + 1 - 2 * w, num_cands = ns - 2 * w, num_cpts = r.num_events;
    const double *dn = dense + r.raw_off + blockIdx.x;
    const i32 *pn = (const i32 *)(posbuf + r.raw_off);
    i64 *cpts = valid_cpts + r.ev_off;
    const i64 n_taken = r.n_taken;
    if (ns <= 0 || num_cpts <= 0) { if (tid == 0) r.status = TBA_INTERNAL; return; }
    if (n_taken < num_cpts) { if (tid == 0) r.status = TBA_FEWER_CPTS; return; }
    // range of the taken scores
    double mn = INFINITY, mx = -INFINITY;
    block_stream2<4>(n_taken, dn, [&](i64, double v) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; });
    for (int mm = 32; mm >= 1; mm >>= 1) {
        double a = shfl_xor_f64(mn, mm), b2 = shfl_xor_f64(mx, mm);
        mn = a < mn ? a : mn; mx = b2 > mx ? b2 : mx;
    }
    if ((tid & 63) == 0) { sm.redd[2 * (tid >> 6)] = mn; sm.redd[2 * (tid >> 6) + 1] = mx; }
    if (tid == 0) s_ntie = 0;
    __syncthreads();
    mn = sm.redd[0]; mx = sm.redd[1];
    for (int qq = 1; qq < SEL_NT / 64; qq++) {
        mn = sm.redd[2 * qq] < mn ? sm.redd[2 * qq] : mn;
        mx = sm.redd[2 * qq + 1] > mx ? sm.redd[2 * qq + 1] : mx;
    }
    __syncthreads();
    // score of the num_cpts-th best taken position (ascending rank n_taken - num_cpts)
    const double tval = block_kth([&](i64 i) { return dn[i]; }, n_taken, n_taken - num_cpts, mn, mx, &sm);
    __syncthreads();
    // taken above / at the threshold; the taken positions AT it, in position order
    i64 c_gt = 0, c_eq = 0;
    block_compact(
        n_taken, [&](i64 i) { return dn[i]; },
        [&](i64, double v) { c_gt += v > tval; c_eq += v == tval; return v == tval; },
        [&](i64 i, i64 o) { if (o < 2048) s_tie[o] = pn[i]; }, s_w);
    c_gt = block_sum_i64(c_gt, &sm.rad);
    c_eq = block_sum_i64(c_eq, &sm.rad);
    const i64 need_eq = num_cpts - c_gt;              // 1 <= need_eq <= c_eq
    // The reference raises when the rank of the last pick in the argsort order, + 1, reaches
    // num_cands (_c_helper.pyx:116-118).  That rank is below ns minus the positions that score
    // under the threshold, and every taken position under the threshold is one: no error while
    // those alone outnumber 2 * width.  Too close to call (or more ties than the list holds): the
    // kernels that keep the scores decide.
    const i64 c_lt = n_taken - c_gt - c_eq;
    if ((num_cpts > 1 && c_lt <= 2 * w) || c_eq > 2048 || need_eq < 1 || need_eq > c_eq) {
        if (tid == 0) r.ed_flag = 1;
        return;
    }
    // ties on the threshold score fall to the higher index: the need_eq last of them
    const i64 idx_thr = need_eq < c_eq ? (i64)s_tie[c_eq - need_eq] : -1;
    block_compact(
        n_taken, [&](i64 i) { return dn[i]; },
        [&](i64 i, double v) { return v > tval || (v == tval && (i64)pn[i] >= idx_thr); },
        [&](i64 i, i64 o) { if (o < num_cpts) cpts[o] = (i64)pn[i] + w; }, s_w);
    if (tid == 0) { r.n_cpts = num_cpts; r.n_ev = num_cpts - 1; }
    (void)num_cands;
}
