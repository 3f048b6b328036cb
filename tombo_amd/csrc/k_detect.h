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
// tile rows: column c of a 128-sample chunk sits at c + c / 32 (one pad per 32-slot word, + one
// per row): the four words of a read that the greedy lanes walk in step start 66 dwords apart instead
// of 64 -- on the same LDS bank every one of their reads was a 4-way conflict, and the LDS pipe,
// shared by the whole CU, is what bounds a step (measured: 9.8 k of its 17 k cycles) -- and the
// rows 266 dwords apart keep the scan's lane-per-read column walk conflict free
#define DT_STRIDE 133
#define DT_PC(c_) ((c_) + ((c_) >> 5))

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

// -DTBA_PHASE_DEBUG=7: cycles per role and part, summed over the steps, into dbg[] of the
// workgroup's first read: 0 scan, 1 loader, 2 greedy (wave 2) = 3 masks + 4 rounds + 5 emission + 6 tail
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 7
#define DT_T0() i64 dt_t_ = (i64)__builtin_readcyclecounter()
#define DT_T(i_) do { const i64 n_ = (i64)__builtin_readcyclecounter(); if (lane == 0) dt_acc[i_] += n_ - dt_t_; dt_t_ = n_; } while (0)
#else
#define DT_T0() do { } while (0)
#define DT_T(i_) do { } while (0)
#endif
// g <- 2 g + (|a| >= |b|): one compare into VCC, one add-with-carry (the C++ form is a compare, a
// select and a shift-or)
__device__ __forceinline__ u32 dt_shift_in_ge(u32 g, double a, double b)
{
    asm("v_cmp_ge_f64 vcc, |%1|, |%2|\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(g) : "v"(a), "v"(b) : "vcc");
    return g;
}
// The loader also IS the last pass of ts.normalize_raw_signal (tombo_stats.py:560-573, k_normalize
// launched with write_norm = 2 leaves it to this kernel): it reads the raw samples (RT: float64,
// float32 or the file's int16), normalises and clips them with the scale values k_normalize left in
// ReadState -- (x - shift) / scale as a multiplication by 1 / scale with two residual corrections,
// bit-identical to the division (div_by_recip) -- writes the normalised signal (the later stages
// read it) and drops it into the tile: one pass over the signal less, 8 S bytes read and 8 S
// written less per read than normalising first and reading the result back here.
template <int R, class RT>
__global__ __launch_bounds__(256, 2) void k_detect(ReadState *rs, i64 n_reads, const DevParams *dp,
    const RT *__restrict__ raw, double *__restrict__ norm, double *__restrict__ dense, double *__restrict__ posbuf,
    i64 dump_off)
{
    static_assert(R >= 1 && R <= 8, "exclusion radius");
    // one LDS array, so that every access of the greedy is smem[integer index]: a select between
    // two __shared__ objects is compiled as a branch around two loads
    constexpr int DT_TILE = DT_READS * DT_STRIDE;                 // doubles per tile buffer
    constexpr int DT_HALO = 3 * DT_TILE;                          // row q: the 2w sums before the tile being scored
    constexpr int DT_CARRY = DT_HALO + DT_READS * DT_HSTRIDE;     // [2][reads][33]: scores of the last word of the previous step
    constexpr int DT_ZROW = DT_CARRY + 2 * DT_READS * DT_CSTRIDE; // 32 zeros (see the greedy's addressing)
    __shared__ double smem[DT_ZROW + 32];
    double *const halo = smem + DT_HALO;
    double *const zrow = smem + DT_ZROW;
#define DT_TILEP(b_) (smem + (b_) * DT_TILE)
    __shared__ i64 s_off[DT_READS], s_n[DT_READS], s_st[DT_READS];
    __shared__ u32 s_cnt[DT_READS];
    __shared__ int s_bad[DT_READS];
    __shared__ double s_shift[DT_READS], s_scale[DT_READS], s_rcp[DT_READS], s_lo[DT_READS], s_hi[DT_READS];
    const int tid = threadIdx.x, lane = tid & 63;
    // roles: 0 scan, 1 loader, 2 / 3 greedy.  Two workgroups share a CU, one wavefront of each per
    // SIMD: with the roles rotated by two in every other workgroup a SIMD holds one of the two heavy
    // (greedy) wavefronts and one light one instead of two heavy ones (TBA_DT_NO_ROTATE: A/B switch)
#ifdef TBA_DT_NO_ROTATE
    const int wave = tid >> 6;
#else
    const int wave = ((tid >> 6) + 2 * (int)(blockIdx.x & 1)) & 3;
#endif
    const i64 r0 = (i64)blockIdx.x * DT_READS;
    const int w = (int)dp->p.running_stat_width, w2 = 2 * w;
    if (tid < DT_READS) {
        const i64 ri = r0 + tid;
        const bool ok = ri < n_reads && rs[ri].status == TBA_OK;
        const bool live = ok && !rs[ri].is_long && !rs[ri].ed_flag; // (flagged by k_normalize: scale out of the loader's range)
        s_off[tid] = live ? rs[ri].raw_off : 0;
        s_n[tid] = live ? rs[ri].n_raw : 0;
        s_st[tid] = live ? rs[ri].raw_off : dump_off;  // where the row's normalised samples go
        s_cnt[tid] = 0;
        s_bad[tid] = 0;
        const double sc = live ? rs[ri].scale : 1.0;
        s_shift[tid] = live ? rs[ri].shift : 0.0;
        s_scale[tid] = sc;
        s_rcp[tid] = 1.0 / sc;
        const bool lim = live && rs[ri].has_lims != 0;
        s_lo[tid] = lim ? rs[ri].lower : -INFINITY;
        s_hi[tid] = lim ? rs[ri].upper : INFINITY;
        if (ok && !live) { rs[ri].ed_flag = 1; rs[ri].n_taken = 0; } // long read: k_long.h + k_peaks
    }
    for (int k = tid; k < DT_READS * DT_HSTRIDE; k += 256) halo[k] = 0.0; // c[0] = 0, nothing before it
    if (tid < 32) zrow[tid] = 0.0;
    for (int k = tid; k < 2 * DT_READS * DT_CSTRIDE; k += 256) smem[DT_CARRY + k] = 0.0;
    __syncthreads();
    i64 n_max = 0;
    for (int q = 0; q < DT_READS; q++) n_max = s_n[q] > n_max ? s_n[q] : n_max;
    const i64 n_steps = (n_max + CS_CHUNK - 1) / CS_CHUNK;

    // ---- wave 1: loader, a row (read) per unit, two samples per lane and access
    double pa[DT_READS], pb[DT_READS];
    // (no divergent branches: a pair is always loaded from inside the read -- clamped -- and what lies
    // past the end is selected away on the way into the tile)
    auto norm_one = [&](double xv, int u) {
        // (reads without limits carry -inf / +inf: the clip is unconditional -- a test of the flag, a
        // value the compiler cannot prove uniform, became a branch around two instructions per value,
        // 80 of them per step with an LDS wait each)
        const double sh = s_shift[u], sc = s_scale[u], y = s_rcp[u], lo = s_lo[u], hi = s_hi[u];
        const double v = div_by_recip(xv - sh, sc, y);
        return v > hi ? hi : (v < lo ? lo : v);
    };
    // One loader step, four rows at a time: normalise the samples of chunk `chunk` (in registers since
    // the previous step), write them out, drop them into tile `t`, and ask for the same rows of chunk
    // `next` into the registers just freed -- every load then has a whole step to come back (issued
    // after the step's work they had a tenth of one, and the wait for them WAS the step: 18 k cycles).
    auto loader_step = [&](i64 chunk, double *t, i64 next) {
        const int pc = DT_PC(2 * lane);                // (2 lane and 2 lane + 1: same word)
#pragma unroll
        for (int u0 = 0; u0 < DT_READS; u0 += 4) {
            __builtin_amdgcn_sched_barrier(0);
            if (chunk >= 0) {
#pragma unroll
                for (int u = u0; u < u0 + 4; u++) {
                    // c_apply_outlier_thresh over (x - shift) / scale (_c_helper.pyx:73-87, tombo_stats.py:560-573)
                    // Every lane normalises the pair it loaded and stores it where it came from: past the
                    // end of the read that is the (clamped) last pair again -- the same values to the same
                    // place, also the last sample of a read of odd length -- and no branch splits the rows
                    // (a store under `if` sat in a block of its own behind an LDS read: 1 000 cycles a row).
                    // Dead rows store into the slack behind the signal buffer (s_st).
                    const i64 k = chunk * CS_CHUNK + 2 * lane, n = s_n[u];
                    const bool full = k + 1 < n, one = k < n;
                    i64 kc = k < n - 2 ? k : n - 2;
                    kc = kc < 0 ? 0 : kc;
                    const double va = norm_one(pa[u], u), vb = norm_one(pb[u], u);
                    st2(norm + s_st[u] + kc, va, vb);
                    t[u * DT_STRIDE + pc] = one ? (full ? va : vb) : 0.0;
                    t[u * DT_STRIDE + pc + 1] = full ? vb : 0.0;
                }
            }
            if (next >= 0) {
#pragma unroll
                for (int u = u0; u < u0 + 4; u++) {
                    const i64 k = next * CS_CHUNK + 2 * lane, n = s_n[u];
                    const RawSamples<RT> x{raw + s_off[u]};
                    i64 kc = k < n - 2 ? k : n - 2;    // (live reads have hundreds of samples; dead rows: n = 0)
                    kc = kc < 0 ? 0 : kc;
                    sig_pair(x, kc, pa[u], pb[u]);
                }
            }
        }
    };
    if (wave == 1) { loader_step(-1, nullptr, 0); loader_step(0, DT_TILEP(0), n_steps > 1 ? 1 : -1); }
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
    double smin = INFINITY, smax = -INFINITY;        // range of the scores this lane emitted (k_pick's select)
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 7
    i64 dt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    for (i64 i = 0; i <= n_steps + 1; i++) {
        DT_T0();
        if (wave == 0) {
            if (i < n_steps && lane < DT_READS) {
                double *row = DT_TILEP(i % 3) + lane * DT_STRIDE;
                const i64 left = my_n - i * CS_CHUNK;
                if (left >= CS_CHUNK) {
#pragma unroll 16
                    for (int k = 0; k < CS_CHUNK; k++) { acc = acc + row[DT_PC(k)]; row[DT_PC(k)] = acc; }
                } else {
                    for (int k = 0; k < CS_CHUNK; k++)
                        if (k < left) { acc = acc + row[DT_PC(k)]; row[DT_PC(k)] = acc; }
                }
            }
            DT_T(0);
        } else if (wave == 1) {
            if (i + 1 < n_steps) loader_step(i + 1, DT_TILEP((i + 1) % 3), i + 2 < n_steps ? i + 2 : -1);
            DT_T(1);
        } else if (i >= 1) {
            // tile j = i - 1: column t holds c[jC + 1 + t]; slot s = jC + t is the score whose window
            // ends there: position k = s + 1 - 2w, |2 c[k+w] - c[k] - c[k+2w]| (pyx:94-98) with
            // c[k+2w] = column t, c[k+w] = column t - w, c[k] = column t - 2w (negative: the halo)
            const i64 j = i - 1;
            const int qz = glane ? q : 0;
            const int it_row = (int)(j % 3) * DT_TILE + qz * DT_STRIDE;   // my read's row of the tile
            const int ih_row = DT_HALO + qz * DT_HSTRIDE;                 // ... of the halo
            const int ic_row = DT_CARRY + cb * DT_READS * DT_CSTRIDE + qz * DT_CSTRIDE;
            const int ic_next = DT_CARRY + (cb ^ 1) * DT_READS * DT_CSTRIDE + qz * DT_CSTRIDE;
            const double *trow = smem + it_row;
            const int S0 = (int)j * CS_CHUNK;         // (slots fit 32 bits: TBA_LONG_RAW reads are not here)
            const int slot0 = S0 - 32 + 32 * h;       // first slot of my word
            const int col0 = 32 * (h - 1);            // its column in the tile (h >= 1)
            const int gni = (int)gn;
            auto valid = [&](int s) { return glane && s >= w2 - 1 && s <= gni - 1; };
            // Addressing: the three sums of slot t sit at smem[ic + t], smem[ib + t'], smem[ia + t'']
            // with per-lane bases.  Word 0 has its scores in the carry: it reads them as "c[k+2w]"
            // against zeros for the other two sums, |(2*0 - 0) - s| = s.  Physical columns (DT_PC):
            // my word's own columns carry h - 1 pads; a sum w / 2w columns back sits in the word
            // below while t < w / 2w: one pad less (t' = t - 1), and for the first word of the tile
            // that is the halo.  All selects are integer selects on the index (no branches).
            const int pbase = col0 + (h - 1);
            const int ic = h == 0 ? ic_row : it_row + pbase;
            const int ib = h == 0 ? DT_ZROW : it_row + pbase - w;
            const int ia = h == 0 ? DT_ZROW : it_row + pbase - w2;
            const int ibl = h == 0 ? DT_ZROW : (h == 1 ? ih_row + w : ib - 1);  // base while t < w
            const int ial = h == 0 ? DT_ZROW : (h == 1 ? ih_row : ia - 1);      // base while t < 2w
            auto score_raw = [&](int t) {              // (2 c[k+w] - c[k]) - c[k+2w] of my slot t: the score up to its sign
                const double cc = smem[ic + t], cbv = smem[(t < w ? ibl : ib) + t], ca = smem[(t < w2 ? ial : ia) + t];
                return ((2 * cbv) - ca) - cc;
            };
            // validity of my 32 slots: w2 - 1 <= slot <= gn - 1
            u32 V = 0;
            {
                const int lo_t = w2 - 1 - slot0 > 0 ? w2 - 1 - slot0 : 0, hi_t = gni - 1 - slot0;
                if (glane && hi_t >= lo_t && lo_t < 32) {
                    const u32 up_to = hi_t >= 31 ? ~0u : ((2u << hi_t) - 1u);
                    V = up_to & (~0u << lo_t);
                }
            }
            // G[d] bit p: the neighbour at p + d outranks p (ties fall to the higher index: >=).
            // The compare of pair (t - d, t) is shifted in from the right as t walks up, so after 32
            // slots it sits at bit 31 - t: reversed and moved down by d it is bit t - d.  Pairs that
            // do not exist (t < d) fall off the top, invalid ends are masked.
            u32 G[R + 1], acc_g[R + 1];
#pragma unroll
            for (int d = 0; d <= R; d++) { G[d] = 0; acc_g[d] = 0; }
            double first[R], last[R], before[R];      // before: the R scores ahead of the current chunk
#pragma unroll
            for (int d = 0; d < R; d++) { first[d] = 0.0; last[d] = 0.0; before[d] = 0.0; }
#pragma unroll
            for (int t8 = 0; t8 < 32; t8 += 8) {      // eight slots' LDS reads in flight
                double sc[8 + R];                      // sc[R + u] = score of slot t8 + u
#pragma unroll
                for (int d = 0; d < R; d++) sc[d] = before[d];
                if (t8 >= w2) {                        // (no slot of the chunk reaches below its word: no selects)
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        sc[R + u] = ((2 * smem[ib + t8 + u]) - smem[ia + t8 + u]) - smem[ic + t8 + u];
                } else {
#pragma unroll
                    for (int u = 0; u < 8; u++) sc[R + u] = score_raw(t8 + u);
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
#pragma unroll
                    for (int d = 1; d <= R; d++) acc_g[d] = dt_shift_in_ge(acc_g[d], sc[R + u], sc[R + u - d]);
                }
                if (h == 4) {
#pragma unroll
                    for (int u = 0; u < 8; u++) smem[ic_next + t8 + u] = fabs(sc[R + u]);
                }
#pragma unroll
                for (int d = 0; d < R; d++) before[d] = sc[8 + d];
                if (t8 == 0) {
#pragma unroll
                    for (int d = 0; d < R; d++) first[d] = sc[R + d];
                }
            }
#pragma unroll
            for (int d = 0; d < R; d++) last[d] = before[d];
#pragma unroll
            for (int d = 1; d <= R; d++) G[d] = (__builtin_bitreverse32(acc_g[d]) >> d) & V & (V >> d);
            // pairs into the next word: its first R scores come from the lane above; the future
            // (beyond the last word) counts as outranking wherever it exists
#pragma unroll
            for (int k = 0; k < R; k++) {
                const double bs = shfl_f64(first[k], (lane + 1) & 63);
                const bool bv = valid(slot0 + 32 + k);
#pragma unroll
                for (int d = k + 1; d <= R; d++) {
                    const int tp = 32 + k - d;        // my position of the pair, 32 - d .. 31
                    const bool b = bv && ((V >> tp) & 1u) && (h == 4 || fabs(bs) >= fabs(last[tp - (32 - R)]));
                    G[d] |= (b ? 1u : 0u) << tp;
                }
            }
            // neighbour words: inside the read by DPP, at its ends the carried context / the future
            u32 vfut = 0;                             // validity of the 32 slots after the last word
            {
                const int f0 = S0 + 128;
                const int lo_s = w2 - 1 > f0 ? w2 - 1 - f0 : 0, hi_s = gni - 1 - f0; // valid: lo_s <= t <= hi_s
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
            DT_T(3);
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
            DT_T(4);
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
                const double s = fabs(score_raw(t));
                smin = s < smin ? s : smin; smax = s > smax ? s : smax;
                dn[o] = s;
                pn[o] = (i32)(slot0 + t - (w2 - 1));
                o++;
            }
            DT_T(5);
            // context of the next step: its word 0 is this step's word 4, below it word 3
            prevT = (u32)__shfl((int)T, (lane + 3) & 63, 64);
#pragma unroll
            for (int d = 1; d <= R; d++) prevX[d] = (u32)__shfl((int)X[d], (lane + 3) & 63, 64);
            prev_emitted = (u32)__shfl((int)T, (lane + 4) & 63, 64);
            // the next tile's halo: the last 2w sums of this one
            if (glane && j < n_steps)                  // (the read's five lanes share the copy)
                for (int k = h; k < w2; k += 5) smem[ih_row + k] = trow[CS_CHUNK - w2 + k + 3]; // (last word: 3 pads)
            cb ^= 1;
            DT_T(6);
        }
        __syncthreads();
    }
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 7
    if (lane == 0 && r0 < n_reads) {
        i64 *dbg = rs[r0].dbg;
        if (wave == 0) dbg[0] = dt_acc[0];
        if (wave == 1) dbg[1] = dt_acc[1];
        if (wave == 2) { dbg[2] = dt_acc[3] + dt_acc[4] + dt_acc[5] + dt_acc[6]; dbg[3] = dt_acc[3]; dbg[4] = dt_acc[4]; dbg[5] = dt_acc[5]; dbg[6] = dt_acc[6]; dbg[7] = n_steps; }
    }
#endif
    if (wave >= 2) {                                  // the read's range: over its five lanes
#pragma unroll
        for (int dd = 1; dd < 5; dd++) {
            const double a = shfl_f64(smin, (lane + dd) & 63), b = shfl_f64(smax, (lane + dd) & 63);
            if (h == 0) { smin = a < smin ? a : smin; smax = b > smax ? b : smax; }
        }
    }
    if (wave >= 2 && glane && h == 0 && gn > 0) {
        ReadState &r = rs[gri];
        r.n_taken = (i64)s_cnt[q];
        r.ed_flag = s_bad[q];
        r.ed_min = smin; r.ed_max = smax;
    }
}

// The cap: keep the num_cpts best taken positions.  dense / posbuf: the taken (score, position)
// list k_detect left, ascending in position.  One workgroup per read.
// ttest: the candidates of c_valid_cpts_w_cap_t_test (_c_helper.pyx:185-202: n - 2w scores, all of them
// candidates) instead of c_valid_cpts_w_cap's (n + 1 - 2w scores, the last 2w no candidates).
__global__ __launch_bounds__(SEL_NT, 4) void k_pick(ReadState *rs, const DevParams *dp,
    const double *dense, const double *posbuf, i64 *valid_cpts, int ttest)
{
    __shared__ BucketSmem sm;
    __shared__ i64 s_w[SEL_NT / 64];
    __shared__ i32 s_tie[2048];
    ReadState &r = rs[blockIdx.x];
    if (r.status != TBA_OK || r.ed_flag) return;
    const int tid = threadIdx.x;
    // (a read this kernel gives up below is flagged and k_peaks, which runs later, overwrites the form)
    if (tid == 0) r.ed_form = ttest ? TBA_ED_FORM_DETECT_TT_PICK : TBA_ED_FORM_DETECT_PICK;
    const i64 w = dp->p.running_stat_width;
    const i64 ns = ttest ? r.n_raw - 2 * w : r.n_raw + 1 - 2 * w, num_cands = ttest ? ns : ns - 2 * w;
    const i64 num_cpts = r.num_events;
    const double *dn = dense + r.raw_off + blockIdx.x;
    const i32 *pn = (const i32 *)(posbuf + r.raw_off);
    i64 *cpts = valid_cpts + r.ev_off;
    const i64 n_taken = r.n_taken;
    if (ns <= 0 || num_cpts <= 0) { if (tid == 0) r.status = TBA_INTERNAL; return; }
    if (n_taken < num_cpts) { if (tid == 0) r.status = TBA_FEWER_CPTS; return; }
    TBA_PHASE_T0(8);
    // range of the taken scores: k_detect / k_detect_tt kept it
    const double mn = r.ed_min, mx = r.ed_max;

    __syncthreads();
    // Score of the num_cpts-th best taken position (ascending rank n_taken - num_cpts) and, per
    // wavefront chunk of the list (compact_chunk), the positions above / at it: the compaction
    // starts from these counts.  The kernel is bound by its passes over the list (0.3 MB per 10 kb
    // read, five passes = 6.6 TB/s when select, counts and a two-pass compaction each made their
    // own), so the common case takes three: a histogram of the scores; one pass that counts the
    // scores in the buckets above the threshold's and collects the members of that bucket with
    // their wavefront (they decide the threshold and the rest of the counts); the emit.
    __shared__ i32 s_gt[SEL_NT / 64], s_eq[SEL_NT / 64];
    const int lane = tid & 63, wv = tid >> 6;
    i64 c0, c1;
    compact_chunk(n_taken, &c0, &c1);
    const double scale = (double)BS_NB / (mx - mn);
    bool fast = mx > mn && scale < 1e300 && n_taken > 192;
    double tval = 0.0;
    if (tid < SEL_NT / 64) { s_gt[tid] = 0; s_eq[tid] = 0; }
    if (fast) {
        if (tid == 0) { sm.nlev = 0; sm.k = n_taken - num_cpts; sm.cnt = n_taken; sm.n_cand = 0; }
        for (int b = tid; b < BS_NB; b += SEL_NT) sm.hist[b] = 0;
        __syncthreads();
        for (i64 i0 = c0; i0 < c1; i0 += 64 * COMPACT_RB) {
            double ld[COMPACT_RB];
#pragma unroll
            for (int k = 0; k < COMPACT_RB; k++) {
                const i64 i = i0 + 64 * k + lane;
                ld[k] = dn[i < c1 ? i : n_taken - 1];
            }
#pragma unroll
            for (int k = 0; k < COMPACT_RB; k++)
                if (i0 + 64 * k + lane < c1) atomicAdd(&sm.hist[bs_bucket(ld[k], mn, scale)], 1u);
        }
        __syncthreads();
        if (tid < 64) bs_locate(&sm, 0, mn, scale);
        __syncthreads();
        fast = sm.cnt <= BS_CAP;
    }
    if (fast) {
        // bs_bucket is monotone in the score: a higher bucket is a higher score
        const int bk = sm.bk[0];
        i32 above = 0;
        for (i64 i0 = c0; i0 < c1; i0 += 64 * COMPACT_RB) {
            double ld[COMPACT_RB];
#pragma unroll
            for (int k = 0; k < COMPACT_RB; k++) {
                const i64 i = i0 + 64 * k + lane;
                ld[k] = dn[i < c1 ? i : n_taken - 1];
            }
#pragma unroll
            for (int k = 0; k < COMPACT_RB; k++) {
                const bool in = i0 + 64 * k + lane < c1;
                const int b = bs_bucket(ld[k], mn, scale);
                above += __popcll(__ballot(in && b > bk));
                if (in && b == bk) { const u32 p = atomicAdd(&sm.n_cand, 1u); sm.cand[p] = ld[k]; s_tie[p] = wv; }
            }
        }
        __syncthreads();
        // rank the bucket's members (sm.cnt <= BS_CAP of them) by counting, as block_kth_fe does
        const int m = (int)sm.cnt;
        const i64 kk = sm.k;
        for (int a = tid; a < m; a += SEL_NT) {
            const double va = sm.cand[a];
            int less = 0, eq_before = 0;
            for (int b2 = 0; b2 < m; b2++) {
                const double vb = sm.cand[b2];
                less += vb < va;
                eq_before += (vb == va) && (b2 < a);
            }
            if (less + eq_before == kk) sm.result = va;    // exactly one member has this rank
        }
        __syncthreads();
        tval = sm.result;
        for (int a = tid; a < m; a += SEL_NT) {
            const double va = sm.cand[a];
            if (va > tval) atomicAdd(&s_gt[s_tie[a]], 1);
            if (va == tval) atomicAdd(&s_eq[s_tie[a]], 1);
        }
        if (lane == 0) atomicAdd(&s_gt[wv], above);
    } else {
        // (few or equal scores, or a crowded bucket: the general select, then the counts)
        auto elems = [&](auto visit) {
            constexpr int U = 8;
            for (i64 base = 0; base < n_taken; base += (i64)U * SEL_NT) {
                double vv[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const i64 i = base + (i64)u * SEL_NT + tid;
                    vv[u] = dn[i < n_taken ? i : n_taken - 1];
                }
#pragma unroll
                for (int u = 0; u < U; u++) visit(vv[u], base + (i64)u * SEL_NT + tid < n_taken);
            }
        };
        tval = block_kth_fe(elems, n_taken, n_taken - num_cpts, mn, mx, &sm);
        __syncthreads();
        i32 gt = 0, eq = 0;
        for (i64 i0 = c0; i0 < c1; i0 += 64 * COMPACT_RB) {
            double ld[COMPACT_RB];
#pragma unroll
            for (int k = 0; k < COMPACT_RB; k++) {
                const i64 i = i0 + 64 * k + lane;
                ld[k] = dn[i < c1 ? i : n_taken - 1];
            }
#pragma unroll
            for (int k = 0; k < COMPACT_RB; k++) {
                const bool in = i0 + 64 * k + lane < c1;
                gt += __popcll(__ballot(in && ld[k] > tval));
                eq += __popcll(__ballot(in && ld[k] == tval));
            }
        }
        if (lane == 0) { s_gt[wv] = gt; s_eq[wv] = eq; }
    }
    __syncthreads();
    TBA_PHASE(8, 0);
    i64 c_gt = 0, c_eq = 0, off = 0;
    for (int q = 0; q < SEL_NT / 64; q++) {
        if (q < wv) off += s_gt[q] + s_eq[q];
        c_gt += s_gt[q]; c_eq += s_eq[q];
    }
    TBA_PHASE(8, 1);
    const i64 need_eq = num_cpts - c_gt;              // 1 <= need_eq <= c_eq
    // The reference raises when the rank of the last pick in the argsort order, + 1, reaches
    // num_cands (_c_helper.pyx:116-118).  That rank is below ns minus the positions that score
    // under the threshold, and every taken position under the threshold is one: no error while
    // those alone outnumber ns - num_cands (2 * width; 0 for the t-test scores).  Too close to call
    // (or more ties than the list holds): the kernels that keep the scores decide.
    const i64 c_lt = n_taken - c_gt - c_eq;
    if ((num_cpts > 1 && c_lt <= ns - num_cands) || c_eq > 2048 || need_eq < 1 || need_eq > c_eq) {
        if (tid == 0) r.ed_flag = 1;
        return;
    }
    // (the positions are fetched with the scores: a load inside the emit would be one memory
    // round trip per row of 64 -- 51 in a row for a 10 kb read's wavefront)
    struct SP { double v; i32 p; };
    auto load = [&](i64 i) { return SP{dn[i], pn[i]}; };
    auto emit = [&](i64, i64 o, SP e) { if (o < num_cpts) cpts[o] = (i64)e.p + w; };
    if (need_eq == c_eq) {
        // every position at the threshold is kept: the wavefronts' offsets are known
        compact_chunk_emit(n_taken, off, load, [&](i64, SP e) { return e.v >= tval; }, emit);
    } else {
        // ties on the threshold score fall to the higher index: the need_eq last of them (rare:
        // the tied positions are listed, in position order, only then)
        block_compact(
            n_taken, [&](i64 i) { return dn[i]; }, [&](i64, double v) { return v == tval; },
            [&](i64 i, i64 o) { if (o < 2048) s_tie[o] = pn[i]; }, s_w);
        __syncthreads();
        const i64 idx_thr = (i64)s_tie[c_eq - need_eq];
        block_compact_chunks(
            n_taken, load, [&](i64, SP e) { return e.v > tval || (e.v == tval && (i64)e.p >= idx_thr); },
            emit, s_w);
    }
    TBA_PHASE(8, 2);
    TBA_PHASE_END(8);
    if (tid == 0) { r.n_cpts = num_cpts; r.n_ev = num_cpts - 1; }
}

// ---------------------------------------------------------------------------------------------
// The same for the t-test scores of RNA (c_valid_cpts_w_cap_t_test, _c_helper.pyx:144-202): a score
// needs the 2w samples around its position and nothing before them, so there is no scan to wait
// for -- one workgroup per read walks its signal in tiles of TT_NEW positions: all wavefronts
// compute the tile's scores into LDS (ttest_score, k_segment.h), then wavefront 0 resolves the
// uncapped greedy on them -- a lane per 32-position word, word 0 = the last word of the previous tile
// (carried: positions near a tile's end wait for the next one), early emission as in k_detect --
// and appends the taken (score, position) pairs to the read's list while the other wavefronts
// stage the next tile's samples.  k_pick caps the list.  Replaces k_scores_ttest (8 S bytes
// written) + k_peaks<5> (read ~2.3 times, + a state byte per sample: 20.7 of cfg4's 103 ms).
#define TT_WORDS 64
#define TT_NEW (32 * (TT_WORDS - 1))          // new positions per tile (word 0 is carried)
#define TT_SPAD(i_) ((i_) + ((i_) >> 5))      // score slot of tile position i: one pad per word
template <int R, int WS, class RT>
__global__ __launch_bounds__(SEL_NT, 4) void k_detect_tt(ReadState *rs, const DevParams *dp,
    const RT *__restrict__ raw, double *__restrict__ dense, double *__restrict__ posbuf)
{
    static_assert(R >= 1 && R <= 8, "exclusion radius");
    __shared__ double rawt[2][TT_NEW + 2 * TT_MAXW];     // the samples of a tile (+ 2w beyond); double buffered
    __shared__ double sbuf[2][TT_SPAD(32 * TT_WORDS) + 2]; // its scores, word h at 33 h; double buffered
    __shared__ u32 s_cnt;
    __shared__ int s_bad;
    ReadState &r = rs[blockIdx.x];
    if (r.status != TBA_OK) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = (int)dp->p.running_stat_width, w2 = 2 * w;
    const int n = (int)r.n_raw, ns = n - w2;             // positions 0 .. ns - 1
    if (w2 > 2 * TT_MAXW || r.n_raw > 0x7fffffffll / 2 || ns <= 0) {
        if (tid == 0) { r.ed_flag = 1; r.n_taken = 0; } // the kernels that keep the scores take it
        return;
    }
    const RawSamples<RT> x{raw + r.raw_off};
    double *dn = dense + r.raw_off + blockIdx.x;
    i32 *pn = (i32 *)(posbuf + r.raw_off);
    if (tid == 0) { s_cnt = 0; s_bad = 0; }
    const int n_tiles = (ns + TT_NEW - 1) / TT_NEW;
    // Wavefront 0 runs the greedy of tile i while wavefronts 1..7 (the "scorers": SC_NT threads)
    // compute the scores of tile i + 1 and fetch the samples of tile i + 2; one barrier per tile.
    constexpr int SC_NT = SEL_NT - 64;
    constexpr int PER = (TT_NEW + 2 * TT_MAXW + SC_NT - 1) / SC_NT; // samples per scorer thread and tile
    const int st = tid - 64;                             // scorer thread index (wave >= 1)
    double pre[PER];
    auto fetch = [&](int tile) {
        const int p0 = tile * TT_NEW, span = TT_NEW + w2;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int c = st + u * SC_NT;
            pre[u] = (c < span && p0 + c < n) ? x[p0 + c] : 0.0;
        }
    };
    auto drop = [&](double *rt) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int c = st + u * SC_NT;
            if (c < TT_NEW + 2 * TT_MAXW) rt[c] = pre[u];
        }
    };
    auto scores = [&](int tile) {                        // tile's scores -> sbuf[tile & 1], word 0 = carried word
        double *sb = sbuf[tile & 1];
        const double *sp = sbuf[(tile & 1) ^ 1], *rt = rawt[tile & 1];
        const int P0 = tile * TT_NEW;
        if (tile < n_tiles) {
            if constexpr (WS > 0 && TT_NEW % WS == 0) {
                // A scorer thread walks a CHAIN of positions WS apart: the right window of one is the
                // left window of the next, so k scores cost k + 1 windows instead of 2 k (the scorers
                // were the longer side of a step: 14 k of its 16 k cycles, -DTBA_PHASE_DEBUG=9).
                // Residue r = thread % WS, segment = thread / WS; the TT_NEW / WS chain steps of a residue
                // are cut into SEGS segments of CH_LO or CH_LO + 1 steps.
                constexpr int STEPS = TT_NEW / WS, SEGS = SC_NT / WS, CH_LO = STEPS / SEGS, N_HI = STEPS - CH_LO * SEGS;
                const int rr = st % WS, seg = st / WS;
                if (seg < SEGS) {
                    const int len = seg < N_HI ? CH_LO + 1 : CH_LO;
                    const int j0 = seg < N_HI ? (CH_LO + 1) * seg : (CH_LO + 1) * N_HI + CH_LO * (seg - N_HI);
                    double m1, v1;
                    tt_window<WS>(rt + (WS * j0 + rr), m1, v1);
#pragma unroll
                    for (int k = 0; k < CH_LO + 1; k++) {
                        if (k < len) {
                            const int c = WS * (j0 + k) + rr;
                            double m2, v2;
                            tt_window<WS>(rt + (c + WS), m2, v2);
                            sb[TT_SPAD(32 + c)] = P0 + c < ns ? tt_combine(m1, m2, v1, v2) : 0.0;
                            m1 = m2; v1 = v2;
                        }
                    }
                }
            } else {
                for (int c = st; c < TT_NEW; c += SC_NT) {
                    const double v = P0 + c < ns ? (WS > 0 ? ttest_score<WS>(rt + c, WS) : ttest_score<0>(rt + c, w)) : 0.0;
                    sb[TT_SPAD(32 + c)] = v;
                }
            }
        }
        if (st < 32) sb[st] = tile > 0 ? sp[TT_SPAD(32 * (TT_WORDS - 1) + st)] : 0.0; // word 0 <- last word
    };
    if (wave >= 1) {
        fetch(0);
        drop(rawt[0]);
    }
    __syncthreads();
    if (wave >= 1) {
        if (n_tiles > 1) fetch(1);
        scores(0);
        if (n_tiles > 1) drop(rawt[1]);
    }
    __syncthreads();
    u32 prevT = 0, prev_emitted = 0, prevX[R + 1];
#pragma unroll
    for (int d = 0; d <= R; d++) prevX[d] = 0;
    double smin = INFINITY, smax = -INFINITY;            // range of the emitted scores (k_pick's select)
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 9
    // cycles of the greedy wavefront / of one scorer wavefront inside their sections, and of the whole loop
    i64 tt_acc = 0;
    const i64 tt_t0 = (i64)__builtin_readcyclecounter();
#define TT_T0() const i64 tt_a_ = (i64)__builtin_readcyclecounter()
#define TT_T1() tt_acc += (i64)__builtin_readcyclecounter() - tt_a_
#else
#define TT_T0() do { } while (0)
#define TT_T1() do { } while (0)
#endif
    for (int i = 0; i <= n_tiles; i++) {                 // (one more step finishes the carried word)
        const double *sb = sbuf[i & 1];
        const int P0 = i * TT_NEW;                       // first new position of the tile
        TT_T0();
        if (wave != 0) {
            // tile i + 1: its samples sit in rawt[(i + 1) & 1]; tile i + 2's are fetched meanwhile and
            // dropped into rawt[i & 1], which nobody reads any more (tile i was scored a step ago)
            if (i + 1 <= n_tiles) {
                if (i + 2 < n_tiles) fetch(i + 2);
                scores(i + 1);
                if (i + 2 < n_tiles) drop(rawt[i & 1]);
            }
        } else {
            // ---- greedy: lane h = positions P0 - 32 + 32 h .. + 31, scores at sb[33 h ..]
            // One wavefront against seven scorers, and the step ends when BOTH are done: at equal
            // priority the arbiter stretched this wavefront's ~3 k cycles of issue over 13 k (of a 16 k
            // step; -DTBA_PHASE_DEBUG=9).  It goes first; the scorers fill what it leaves.
#ifndef TBA_TT_NO_PRIO
            __builtin_amdgcn_s_setprio(3);
#endif
            const int h = lane;
            const int pos0 = P0 - 32 + 32 * h;
            const double *row = sb + 33 * h;
            u32 V = 0;
            {
                const int lo_t = pos0 < 0 ? -pos0 : 0, hi_t = ns - 1 - pos0;
                if (hi_t >= lo_t && lo_t < 32) {
                    const u32 up_to = hi_t >= 31 ? ~0u : ((2u << hi_t) - 1u);
                    V = up_to & (~0u << lo_t);
                }
            }
            u32 G[R + 1], acc_g[R + 1];
#pragma unroll
            for (int d = 0; d <= R; d++) { G[d] = 0; acc_g[d] = 0; }
            double first[R], last[R], before[R];
#pragma unroll
            for (int d = 0; d < R; d++) { first[d] = 0.0; last[d] = 0.0; before[d] = 0.0; }
#pragma unroll
            for (int t8 = 0; t8 < 32; t8 += 8) {
                double sc[8 + R];
#pragma unroll
                for (int d = 0; d < R; d++) sc[d] = before[d];
#pragma unroll
                for (int u = 0; u < 8; u++) sc[R + u] = row[t8 + u];
#pragma unroll
                for (int u = 0; u < 8; u++) {
#pragma unroll
                    for (int d = 1; d <= R; d++) acc_g[d] = dt_shift_in_ge(acc_g[d], sc[R + u], sc[R + u - d]);
                }
#pragma unroll
                for (int d = 0; d < R; d++) before[d] = sc[8 + d];
                if (t8 == 0) {
#pragma unroll
                    for (int d = 0; d < R; d++) first[d] = sc[R + d];
                }
            }
#pragma unroll
            for (int d = 0; d < R; d++) last[d] = before[d];
#pragma unroll
            for (int d = 1; d <= R; d++) G[d] = (__builtin_bitreverse32(acc_g[d]) >> d) & V & (V >> d);
            // pairs into the next word (the lane above); the future counts as outranking
            const bool top = h == TT_WORDS - 1;
#pragma unroll
            for (int k = 0; k < R; k++) {
                const double bs = shfl_f64(first[k], (lane + 1) & 63);
                const int pb = pos0 + 32 + k;
                const bool bv = pb >= 0 && pb < ns;
#pragma unroll
                for (int d = k + 1; d <= R; d++) {
                    const int tp = 32 + k - d;
                    const bool b = bv && ((V >> tp) & 1u) && (top || bs >= last[tp - (32 - R)]);
                    G[d] |= (b ? 1u : 0u) << tp;
                }
            }
            u32 vfut = 0;                                 // validity of the 32 positions after the tile
            {
                const int hi_s = ns - 1 - (P0 + TT_NEW);
                if (hi_s >= 0) vfut = hi_s >= 31 ? ~0u : ((2u << hi_s) - 1u);
            }
            const u32 va = dt_lane_above(V);
            const u32 Vn = top ? vfut : va;
            u32 X[R + 1], Hm[R + 1];
#pragma unroll
            for (int d = 1; d <= R; d++) {
                const u32 pv = V & dt_down(V, Vn, d);
                X[d] = pv & ~G[d];
                const u32 xb = dt_lane_below(X[d]);
                Hm[d] = V & dt_up(X[d], h == 0 ? prevX[d] : xb, d);
            }
            u32 T = 0, S = 0, U = V;
            for (int round = 0; round < DT_MAX_ROUNDS; round++) {
                const u32 ta = dt_lane_above(T), tb = dt_lane_below(T);
                const u32 ua = dt_lane_above(U), ub = dt_lane_below(U);
                const u32 Tn = top ? 0u : ta, Tp = h == 0 ? prevT : tb;
                const u32 Un = top ? vfut : ua, Up = h == 0 ? 0u : ub;
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
            if (!top && U != 0) s_bad = 1;                // a chain longer than a word: not for this kernel
            // emission in position order; the last word's decisions are final too and emitted now
            const u32 E = h == 0 ? T & ~prev_emitted : T;
            const int ce = __popc(E);
            int inc = ce;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const int t = __shfl_up(inc, dd, 64);
                if (lane >= dd) inc += t;
            }
            const int tot = __shfl(inc, 63, 64);
            const u32 base = s_cnt;
            if (lane == 0) s_cnt = base + (u32)tot;
            u32 o = base + (u32)(inc - ce);
            for (u32 m = E; m != 0; m &= m - 1u) {
                const int t = __ffs((int)m) - 1;
                const double sv = row[t];
                smin = sv < smin ? sv : smin; smax = sv > smax ? sv : smax;
                dn[o] = sv;
                pn[o] = pos0 + t;
                o++;
            }
            // context of the next tile: its word 0 is this tile's last word, below it word 62
            prevT = (u32)__shfl((int)T, TT_WORDS - 2, 64);
#pragma unroll
            for (int d = 1; d <= R; d++) prevX[d] = (u32)__shfl((int)X[d], TT_WORDS - 2, 64);
            prev_emitted = (u32)__shfl((int)T, TT_WORDS - 1, 64);
#ifndef TBA_TT_NO_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        }
        TT_T1();
        __syncthreads();
    }
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 9
    if (lane == 0 && wave == 0) { r.dbg[0] = tt_acc; r.dbg[2] = (i64)__builtin_readcyclecounter() - tt_t0; r.dbg[3] = n_tiles; }
    if (lane == 0 && wave == 1) r.dbg[1] = tt_acc;
    if (lane == 0 && wave == 7) r.dbg[4] = tt_acc;
#endif
    if (wave == 0) {
        for (int mm = 32; mm >= 1; mm >>= 1) {
            const double a = shfl_xor_f64(smin, mm), b = shfl_xor_f64(smax, mm);
            smin = a < smin ? a : smin; smax = b > smax ? b : smax;
        }
    }
    if (tid == 0) { r.n_taken = (i64)s_cnt; r.ed_flag = s_bad; r.ed_min = smin; r.ed_max = smax; }
}
