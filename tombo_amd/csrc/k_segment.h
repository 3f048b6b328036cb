// k_segment.h -- per-read normalisation and event detection (segment_signal,
// resquiggle.py:1057-1120) as batch kernels.
#pragma once
#include "k_select.h"

// ---------------------------------------------------------------------------------------------
// ts.normalize_raw_signal (tombo_stats.py:482-573) for 'median', 'median_const_scale' and given
// scale values; one workgroup per read.  mode 0: DNA order (normalise first); mode 1: RNA,
// called after event detection with scale values already in ReadState.
// RT: the raw sample type at the boundary (double, float, or the int16 DAC values of the FAST5
// file, resquiggle.py:1397); widening to float64 is exact, so every pass sees the values the
// reference sees after its own int16 -> float64 promotion.
// The read's `norm` slice serves as scratch (window lists) until the final pass writes the signal;
// write_norm == 0: only the scale values are produced; 2: see below.
// Medians: int16 input -> one counting pass (block_int_medians); float input -> one pass per
// median through a sampled window (block_median_window); short reads and every failure of the
// fast forms -> the generic bucket select (two passes per median).
template <class RT> struct raw_is_int { static constexpr bool value = false; };
template <> struct raw_is_int<int16_t> { static constexpr bool value = true; };
#define NORM_WINDOW_MIN 16384 // below this the generic select is cheap anyway

template <class RT>
// (second launch bound = wavefronts per SIMD the register allocation must leave room for: these
// workgroup-per-read kernels are latency bound, their time follows the resident workgroups per
// CU -- see tools/occupancy.hip -- and without the bound the allocator drifts across an
// occupancy step with every edit)
__global__ __launch_bounds__(SEL_NT, 4) void k_normalize(ReadState *rs, const DevParams *dp,
    const RT *raw, double *norm, const double *sv_in, int mode, int write_norm)
{
    __shared__ BucketSmem sm;
    ReadState &r = rs[blockIdx.x];
    if (r.status != TBA_OK) return;
    const int tid = threadIdx.x;
    TBA_PHASE_T0(2);
    const i64 n = r.n_raw;
    const RawSamples<RT> x{raw + r.raw_off};
    double *y = norm + r.raw_off;
    const tba_opts &o = dp->o;
    double shift, scale, lo = 0, hi = 0;
    double xlo = 0, xhi = 0, mn = 0, mx = 0; // middle order statistics / range of the raw signal
    bool have_lims = false, have_dev = false;
    double dlo = 0, dhi = 0; // middle order statistics of |x - shift|
    if (r.sv_flags & 1) {
        shift = sv_in[4 * blockIdx.x + 0];
        scale = sv_in[4 * blockIdx.x + 1];
        if (r.sv_flags & 2) { have_lims = true; lo = sv_in[4 * blockIdx.x + 2]; hi = sv_in[4 * blockIdx.x + 3]; }
    } else if (mode == 1 && !o.has_const_scale && o.use_rna_event_scale) {
        // get_scale_values_from_events result, tombo_stats.py:217-233
        shift = r.shift; scale = r.scale; have_lims = true; lo = r.lower; hi = r.upper;
    } else {
        // strided sample, kept in registers: range guess for the selects and window steering
        double samp[WS_PER];
        double smn = INFINITY, smx = -INFINITY;
#pragma unroll
        for (int q = 0; q < WS_PER; q++) {
            const i64 si = (n * (i64)(q * SEL_NT + tid)) / (SEL_NT * WS_PER);
            samp[q] = x[si];
            smn = samp[q] < smn ? samp[q] : smn; smx = samp[q] > smx ? samp[q] : smx;
        }
        for (int mm = 32; mm >= 1; mm >>= 1) {
            double a = shfl_xor_f64(smn, mm), b2 = shfl_xor_f64(smx, mm);
            smn = a < smn ? a : smn; smx = b2 > smx ? b2 : smx;
        }
        if ((tid & 63) == 0) { sm.redd[2 * (tid >> 6)] = smn; sm.redd[2 * (tid >> 6) + 1] = smx; }
        __syncthreads();
        smn = sm.redd[0]; smx = sm.redd[1];
        for (int w = 1; w < SEL_NT / 64; w++) {
            smn = sm.redd[2 * w] < smn ? sm.redd[2 * w] : smn;
            smx = sm.redd[2 * w + 1] > smx ? sm.redd[2 * w + 1] : smx;
        }
        __syncthreads();
        double span = smx - smn;
        span = span > 0 ? span : 1.0;
        mn = smn - span; mx = smx + span;
        TBA_PHASE(2, 0);
        bool have_med = false;
        if constexpr (raw_is_int<RT>::value) {
            // both medians from one counting pass (the deviations of the scale are ranked from the
            // same histogram once the median is known)
            if (n >= 4096 && block_int_medians(x, n, (int)smn, (int)smx, &sm, &xlo, &xhi, &dlo, &dhi)) {
                shift = (n & 1) ? xlo : (xlo + xhi) / 2.0;
                have_med = true;
                if (!o.has_const_scale) { scale = (n & 1) ? dlo : (dlo + dhi) / 2.0; have_dev = true; }
            }
            __syncthreads();
        }
        // Up to three medians over the signal, one call site (a loop the compiler must not unroll:
        // three inlined copies of the selects cost 40 VGPRs and a workgroup per CU):
        //   what 0: np.median(x) -> shift;   what 1: np.median(|x - shift|) -> scale (mad);
        //   what 2: np.median(|norm - np.median(norm)|) -> the outlier limits of the normalised
        //           signal (c_apply_outlier_thresh's caller, tombo_stats.py:560-570).
        // Each first tries the one-pass sampled window (the window list lives in the read's norm
        // slice, which is only written by the final pass), then the generic two-pass select.
        // RNA with scale_values=None and no event scaling normalises without an outlier threshold.
        const bool thresh = o.has_outlier_thresh && !(mode == 1 && !o.has_const_scale);
        double med = 0, mad = 0;
#pragma nounroll
        for (int what = 0; what < 3; what++) {
            if (what == 0 && have_med) continue;
            if (what == 1) {
                if (o.has_const_scale) { scale = o.const_scale; continue; }
                if (have_dev) continue;
            }
            if (what == 2) {
                if (!thresh) break;
                // np.median(norm): x -> (x - shift) / scale is monotone, so the middle order
                // statistics of the normalised signal are the images of the raw ones found above
                // (same two values the reference averages; for a negative const scale they swap
                // places, the sum does not care)
                const double ylo = (xlo - shift) / scale, yhi = (xhi - shift) / scale;
                med = (n & 1) ? ylo : (ylo + yhi) / 2.0;
                if (have_dev && med == 0.0 && scale > 0) {
                    // |norm - 0| = RN(|x - shift| / scale) is a monotone image of the deviations
                    // the scale was just selected from, so its middle order statistics are the
                    // images of theirs: no pass over the signal (med is exactly 0 for every odd
                    // length and for the even ones whose two middle samples sit symmetrically
                    // around the shift; float input of even length rarely does: med is a few
                    // 1e-16 there and the deviations have to be ranked again)
                    const double m_lo = dlo / scale, m_hi = dhi / scale;
                    mad = (n & 1) ? m_lo : (m_lo + m_hi) / 2.0;
                    break;
                }
            }
            auto of = [&](double xv) {
                return what == 0 ? xv : (what == 1 ? fabs(xv - shift) : fabs((xv - shift) / scale - med));
            };
            auto val = [&](i64 i) { return of(x[i]); };
            double a_lo = 0, a_hi = 0, res = 0;
            bool done = false;
            if (!raw_is_int<RT>::value && n >= NORM_WINDOW_MIN) {
                double ds[WS_PER];
#pragma unroll
                for (int q = 0; q < WS_PER; q++) ds[q] = of(samp[q]);
                done = block_median_window<2>(x, of, n, ds, y, n, &sm, &a_lo, &a_hi);
                if (done) res = (n & 1) ? a_lo : (a_lo + a_hi) / 2.0;
                __syncthreads();
            }
            if (!done) { // bucket range: anything works, a good guess saves refinement levels
                double rhi = mx;
                if (what == 1) { const double a = mx - shift, b2 = shift - mn; rhi = a > b2 ? a : b2; }
                if (what == 2) { const double e0 = of(mn), e1 = of(mx); rhi = e0 > e1 ? e0 : e1; }
                res = block_median_fast(val, n, what == 0 ? mn : 0.0, rhi, &sm, &a_lo, &a_hi);
            }
            if (what == 0) { shift = res; xlo = a_lo; xhi = a_hi; have_med = true; }
            else if (what == 1) { scale = res; dlo = a_lo; dhi = a_hi; have_dev = true; }
            else mad = res;
            TBA_PHASE(2, 1 + what);
        }
        if (thresh) {
            lo = med - (mad * o.outlier_thresh);
            hi = med + (mad * o.outlier_thresh);
            have_lims = true;
        }
    }
    // A scale of exactly 0 (the MAD of a flat signal, e.g. a saturated int16 read): the reference
    // divides by it under np.seterr(all='raise') (resquiggle.py:29, tombo_stats.py:19,553) and dies
    // with a FloatingPointError -- an unexpected error, whatever the batch size or the form.
    if (scale == 0.0) {
        if (tid == 0) r.status = TBA_INTERNAL;
        return;
    }
    // The normalised signal is written once, at the end: the passes in between recompute
    // (x - shift) / scale on the fly (same operation, same bits).
    TBA_PHASE(2, 3);
    // write_norm == 2: only the reads whose normalised signal k_detect's loader (k_detect.h) will not
    // write on its way: the long ones (k_long.h takes their scan), and the reads whose scale is
    // outside the range in which the loader's reciprocal form of the division is the division bit
    // for bit (no overflow or underflow of quotient and residuals for any sample: 2^-500 .. 2^500,
    // NaN fails the test too) -- those are flagged for the kernels that keep the scores here
    const bool recip_ok = fabs(scale) >= 0x1p-500 && fabs(scale) <= 0x1p500 && fabs(shift) <= 0x1p500;
    if (write_norm == 2 && !recip_ok && tid == 0) r.ed_flag = 1;
    if (write_norm == 1 || (write_norm == 2 && (r.is_long || !recip_ok))) {
        if (have_lims) {
            // c_apply_outlier_thresh, _c_helper.pyx:73-87
            block_map2<2>(n, x, y, [&](double xv) {
                const double v = (xv - shift) / scale;
                return v > hi ? hi : (v < lo ? lo : v);
            });
        } else {
            block_map2<2>(n, x, y, [&](double xv) { return (xv - shift) / scale; });
        }
    }
    TBA_PHASE(2, 4);
    TBA_PHASE_END(2);
    if (tid == 0) {
        r.shift = shift; r.scale = scale; r.lower = lo; r.upper = hi;
        r.has_lims = have_lims ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// np.cumsum([0, x]) with its left-to-right float64 accumulation order (the order feeds the
// change-point ranking, so a parallel scan is not an option): one lane per read.
// csum has n_raw + 1 entries per read at raw_off + read_index.
__global__ __launch_bounds__(64) void k_cumsum(const ReadState *rs, i64 n_reads,
    const double *__restrict__ norm, double *__restrict__ csum)
{
    i64 ri = (i64)blockIdx.x * 64 + threadIdx.x;
    if (ri >= n_reads) return;
    const ReadState &r = rs[ri];
    if (r.status != TBA_OK) return;
    const double *__restrict__ x = norm + r.raw_off;
    double *__restrict__ c = csum + r.raw_off + ri;
    double acc = 0.0;
    c[0] = acc;
    const i64 n = r.n_raw;
    i64 i = 0;
    for (; i + 16 <= n; i += 16) { // loads first, then the dependent adds, then the stores
        double t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = x[i + k];
#pragma unroll
        for (int k = 0; k < 16; k++) { acc = acc + t[k]; t[k] = acc; }
#pragma unroll
        for (int k = 0; k < 16; k++) c[i + 1 + k] = t[k];
    }
    for (; i < n; i++) {
        acc = acc + x[i];
        c[i + 1] = acc;
    }
}

// c_valid_cpts_w_cap scores, _c_helper.pyx:94-98: |(2*c[k+w]) - c[k] - c[k+2w]|
__global__ __launch_bounds__(256) void k_scores_dna(const ReadState *rs, const DevParams *dp,
    const double *csum, double *score)
{
    const ReadState &r = rs[blockIdx.y];
    if (r.status != TBA_OK) return;
    const i64 w = dp->p.running_stat_width;
    const i64 ns = r.n_raw + 1 - 2 * w;
    const double *c = csum + r.raw_off + blockIdx.y;
    double *s = score + r.raw_off;
    for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < ns; k += (i64)gridDim.x * 256)
        s[k] = fabs(((2 * c[k + w]) - c[k]) - c[k + 2 * w]);
}

// k_cumsum + k_scores_dna in one pass (2 * running_stat_width <= 64): the cumulative sum never
// goes to memory.  Its left-to-right accumulation needs one LANE per read, and a lane per read
// would touch 64 different cache lines per global access, so the signal is transposed through
// LDS and the parts of a step are pipelined: a workgroup (4 waves) owns CS_READS reads and three
// CS_READS x CS_CHUNK tiles (rows padded by one to spread the banks); in step i waves 1..3 (a)
// drop the samples they fetched during step i-1 into tile (i+1) % 3, (b) fetch chunk i+2 into
// registers, (c) turn tile (i-1) % 3 -- scanned during step i-1 -- into scores (the 2w sums
// before the tile come from a per-row halo) and write them out, all as coalesced half rows,
// while wave 0 walks tile i % 3 column-wise (lane = read).  One barrier per step; a step costs
// about one memory round trip, 32 reads x 128 samples per workgroup keep enough of them in
// flight (measured against 64 x 64 tiles, an LDS-only barrier and separate load / store waves:
// all slower).
// Reads per workgroup: a template parameter, chosen per launch (cs_reads_for).  A workgroup's LDS
// is CS_READS x 3.6 KB: 32 reads = 115 KB = one workgroup per CU (8 192 reads per round of
// workgroups), 20 reads = 72 KB = two per CU (10 240 per round).  The kernel is latency bound -- a
// round costs n_steps memory round trips whatever its width -- so what counts is the number of
// rounds: the 10 000-read batch is 2 rounds at 32 (the second 22 % full: 7.1 ms) and 1 at 20
// (3.3 ms); a 4 096-read batch is one round either way and the wider workgroup, alone on its
// CU, is faster (2.4 vs 3.7 ms).  (Normalising the raw samples inside the loaders, so that
// k_normalize needs no final pass, was measured twice: 7 -> 19 ms at 32 reads (spills), 3.3 ->
// 8.2 ms at 20 (float64 input; 15 ms for int16) against 2.8 ms saved in k_normalize.)
#define CS_CHUNK 128
#define CS_STRIDE (CS_CHUNK + 1)
__host__ inline int cs_reads_for(i64 n_reads)
{
    const i64 r32 = (n_reads + 8191) / 8192, r20 = (n_reads + 10239) / 10240;
    return r20 < r32 ? 20 : 32;
}
// MODE 1 (identify_stalls, tombo_stats.py:277: np.cumsum(all_raw_signal)): the same pipeline over
// the RAW samples (any boundary type, widened exactly), and what the store step writes is the
// cumulative sum itself: score[raw_off + read + k] = sum of the first k samples, k = 0..n_raw.
// only_flagged: the reads k_detect / k_pick (k_detect.h) left to this kernel (ReadState.ed_flag).
template <int CS_READS, class RT = double, int MODE = 0>
__global__ __launch_bounds__(256) void k_cumsum_scores(const ReadState *rs, i64 n_reads,
    const DevParams *dp, const RT *__restrict__ norm, double *__restrict__ score, int only_flagged = 0)
{
    // half rows per loader wave: 3 x CS_UNITS >= 2 x CS_READS, even so halves pair up
    constexpr int CS_UNITS = (((2 * CS_READS + 2) / 3) + 1) & ~1;
    __shared__ double tile[3][CS_READS * CS_STRIDE];
    __shared__ double halo[MODE == 0 ? CS_READS * 64 : 1]; // row q: the 2w sums before the tile being stored
    __shared__ i64 s_off[CS_READS], s_n[CS_READS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const i64 r0 = (i64)blockIdx.x * CS_READS;
    const int w = (int)dp->p.running_stat_width, w2 = 2 * w;
    if (tid < CS_READS) {
        const i64 ri = r0 + tid;
        // (long reads have a workgroup of their own: k_cumsum_scores_long, k_long.h)
        const bool live = ri < n_reads && rs[ri].status == TBA_OK && !rs[ri].is_long &&
                          (!only_flagged || rs[ri].ed_flag);
        s_off[tid] = live ? rs[ri].raw_off : 0;
        s_n[tid] = live ? rs[ri].n_raw : 0;
        if (MODE == 1 && live) score[rs[ri].raw_off + ri] = 0.0; // c[0]
    }
    if (MODE == 0)
        for (int k = tid; k < CS_READS * 64; k += 256) halo[k] = 0.0; // c[0] = 0, nothing before it
    __syncthreads();
    i64 n_max = 0;
    for (int q = 0; q < CS_READS; q++) n_max = s_n[q] > n_max ? s_n[q] : n_max;
    const i64 n_steps = (n_max + CS_CHUNK - 1) / CS_CHUNK;
    // loader waves: unit u is half `x & 1` of row `x >> 1`, x = (wave - 1) * CS_UNITS + u
    i64 uoff[CS_UNITS]; // raw_off of the unit's read
    i64 un[CS_UNITS];   // its length (0: no such unit)
    int ucol[CS_UNITS];
    double pre[CS_UNITS]; // the chunk fetched during the previous step
    if (wave > 0) {
#pragma unroll
        for (int u = 0; u < CS_UNITS; u++) {
            const int x = (wave - 1) * CS_UNITS + u, q = x >> 1;
            const bool have = x < 2 * CS_READS;
            ucol[u] = (x & 1) * 64 + lane;
            un[u] = have ? s_n[have ? q : 0] : 0;
            uoff[u] = s_off[have ? q : 0];
        }
    }
    auto fetch = [&](i64 chunk) {
#pragma unroll
        for (int u = 0; u < CS_UNITS; u++) {
            const i64 k = chunk * CS_CHUNK + ucol[u];
            pre[u] = k < un[u] ? (double)norm[uoff[u] + k] : 0.0;
        }
    };
    auto drop = [&](double *t) {
#pragma unroll
        for (int u = 0; u < CS_UNITS; u++) {
            const int x = (wave - 1) * CS_UNITS + u;
            if (x < 2 * CS_READS) t[(x >> 1) * CS_STRIDE + ucol[u]] = pre[u];
        }
    };
    if (wave > 0) { fetch(0); drop(tile[0]); if (n_steps > 1) fetch(1); }
    __syncthreads();
    const i64 my_n = lane < CS_READS ? s_n[lane < CS_READS ? lane : 0] : 0;
    double acc = 0.0; // wave 0: running sum of read r0 + lane
    for (i64 i = 0; i <= n_steps; i++) { // one extra step turns the last tile into scores
        if (wave == 0) {
            if (i < n_steps && lane < CS_READS) {
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
        } else {
            if (i + 1 < n_steps) drop(tile[(i + 1) % 3]);
            if (i + 2 < n_steps) fetch(i + 2);
            if (MODE == 1 && i >= 1) { // the sums of tile i - 1 go out as they are
                const double *t = tile[(i - 1) % 3];
                double cc[CS_UNITS];
#pragma unroll
                for (int u = 0; u < CS_UNITS; u++) {
                    const int x = (wave - 1) * CS_UNITS + u, q = x < 2 * CS_READS ? x >> 1 : 0;
                    cc[u] = t[q * CS_STRIDE + ucol[u]];
                }
#pragma unroll
                for (int u = 0; u < CS_UNITS; u++) {
                    const int x = (wave - 1) * CS_UNITS + u, q = x < 2 * CS_READS ? x >> 1 : 0;
                    const i64 k = (i - 1) * CS_CHUNK + ucol[u]; // sample index; c index k + 1
                    if (k < un[u]) score[uoff[u] + (r0 + q) + 1 + k] = cc[u];
                }
            }
            if (MODE == 0 && i >= 1) {
                // tile j = i - 1 holds c[jC + 1 .. jC + C] (column t <-> c index jC + 1 + t);
                // halo row q holds c[jC + 1 - 2w .. jC].  Column t closes the window of position
                // k = jC + 1 + t - 2w: score[k] = |2 c[k + w] - c[k] - c[k + 2w]| (pyx:94-98)
                const double *t = tile[(i - 1) % 3];
                const i64 m0 = (i - 1) * CS_CHUNK + 1; // c index of column 0
                double ca[CS_UNITS], cb[CS_UNITS], cc[CS_UNITS];
#pragma unroll
                for (int u = 0; u < CS_UNITS; u++) { // all LDS reads first
                    const int x = (wave - 1) * CS_UNITS + u, q = x < 2 * CS_READS ? x >> 1 : 0;
                    const double *row = t + q * CS_STRIDE, *hr = halo + q * 64;
                    const int c = ucol[u];
                    cc[u] = row[c];
                    // c[k] sits 2w columns to the left, c[k + w] w columns: tile or halo
                    ca[u] = c >= w2 ? row[c - w2] : hr[c];
                    cb[u] = c >= w ? row[c - w] : hr[c + w];
                }
#pragma unroll
                for (int u = 0; u < CS_UNITS; u++) {
                    const i64 k = m0 + ucol[u] - w2;
                    if (k >= 0 && k < un[u] + 1 - w2) score[uoff[u] + k] = fabs(((2 * cb[u]) - ca[u]) - cc[u]);
                }
                __builtin_amdgcn_wave_barrier();
                // next halo: the last 2w columns of this tile (second half of each row)
#pragma unroll
                for (int u = 0; u < CS_UNITS; u++) {
                    const int x = (wave - 1) * CS_UNITS + u;
                    if (x < 2 * CS_READS && (x & 1) && lane >= 64 - w2)
                        halo[(x >> 1) * 64 + lane - (64 - w2)] = cc[u];
                }
            }
        }
        __syncthreads();
    }
}

// c_valid_cpts_w_cap_t_test scores, _c_helper.pyx:152-183 (sequential sums inside each window)
// Each workgroup step stages 256 + 2w consecutive samples in LDS (one coalesced load) and every
// thread reads its two windows from there; WS > 0: the window width as a compile-time constant
// (w = 12 is the RNA default), loops fully unrolled, each sample read once.
template <int WS, class Acc>
__device__ __forceinline__ double ttest_score(Acc t, int w)
{
    const int W = WS > 0 ? WS : w;
    double m1 = 0, m2 = 0, var1 = 0, var2 = 0, d;
    if constexpr (WS > 0) {
        double a[WS], b[WS];
#pragma unroll
        for (int j = 0; j < WS; j++) { a[j] = t[j]; b[j] = t[WS + j]; }
#pragma unroll
        for (int j = 0; j < WS; j++) m1 += a[j];
        m1 /= (double)WS;
#pragma unroll
        for (int j = 0; j < WS; j++) m2 += b[j];
        m2 /= (double)WS;
#pragma unroll
        for (int j = 0; j < WS; j++) { d = a[j] - m1; var1 += d * d; }
#pragma unroll
        for (int j = 0; j < WS; j++) { d = b[j] - m2; var2 += d * d; }
    } else {
        for (int j = 0; j < W; j++) m1 += t[j];
        m1 /= (double)W;
        for (int j = 0; j < W; j++) m2 += t[W + j];
        m2 /= (double)W;
        for (int j = 0; j < W; j++) { d = t[j] - m1; var1 += d * d; }
        for (int j = 0; j < W; j++) { d = t[W + j] - m2; var2 += d * d; }
    }
    if (var1 + var2 == 0) return 0.0;
    return m1 > m2 ? (m1 - m2) / sqrt(var1 + var2) : (m2 - m1) / sqrt(var1 + var2);
}
// The two windows of position pos are the windows STARTING at pos and at pos + w: each is the same
// expression of its own w samples (sum from the left, one division, squared deviations summed from
// the left), so a window computed once serves as the right window of pos - w and the left one of pos.
template <int WS, class Acc>
__device__ __forceinline__ void tt_window(Acc t, double &m, double &var)
{
    double a[WS];
#pragma unroll
    for (int j = 0; j < WS; j++) a[j] = t[j];
    double sm = 0, sv = 0, d;
#pragma unroll
    for (int j = 0; j < WS; j++) sm += a[j];
    sm /= (double)WS;
#pragma unroll
    for (int j = 0; j < WS; j++) { d = a[j] - sm; sv += d * d; }
    m = sm; var = sv;
}
__device__ __forceinline__ double tt_combine(double m1, double m2, double var1, double var2)
{
    if (var1 + var2 == 0) return 0.0;
    return m1 > m2 ? (m1 - m2) / sqrt(var1 + var2) : (m2 - m1) / sqrt(var1 + var2);
}
#define TT_MAXW 64
template <class RT>
__global__ __launch_bounds__(256) void k_scores_ttest(const ReadState *rs, const DevParams *dp,
    const RT *raw, double *score, int only_flagged = 0)
{
    __shared__ double tile[256 + 2 * TT_MAXW];
    const ReadState &r = rs[blockIdx.y];
    if (r.status != TBA_OK) return;
    if (only_flagged && !r.ed_flag) return; // k_detect_tt / k_pick (k_detect.h) finished this read
    const i64 w = dp->p.running_stat_width;
    const i64 ns = r.n_raw - 2 * w;
    const RawSamples<RT> x{raw + r.raw_off};
    double *s = score + r.raw_off;
    if (w > TT_MAXW) { // no tile: straight from memory
        for (i64 pos = (i64)blockIdx.x * 256 + threadIdx.x; pos < ns; pos += (i64)gridDim.x * 256)
            s[pos] = ttest_score<0>(x + pos, (int)w);
        return;
    }
    // the samples of the NEXT step are fetched into registers before this step's windows are
    // evaluated (the load latency hides behind the ~150 float64 operations of a score)
    const int span = 256 + 2 * (int)w, tid = threadIdx.x;
    const bool second = tid + 256 < span; // the 2w samples past the 256: first threads, 2nd load
    const i64 step = (i64)gridDim.x * 256;
    double v0 = 0.0, v1 = 0.0;
    auto fetch = [&](i64 p0) {
        const i64 q0 = p0 + tid, q1 = p0 + tid + 256;
        v0 = q0 < r.n_raw ? x[q0] : 0.0;
        v1 = second && q1 < r.n_raw ? x[q1] : 0.0;
    };
    i64 p0 = (i64)blockIdx.x * 256;
    if (p0 < ns) fetch(p0);
    for (; p0 < ns; p0 += step) {
        __syncthreads();
        tile[tid] = v0;
        if (second) tile[tid + 256] = v1;
        __syncthreads();
        if (p0 + step < ns) fetch(p0 + step);
        const i64 pos = p0 + tid;
        if (pos < ns) s[pos] = w == 12 ? ttest_score<12>(tile + tid, 12) : ttest_score<0>(tile + tid, (int)w);
    }
}

// ---------------------------------------------------------------------------------------------
// The capped greedy of c_valid_cpts_w_cap (_c_helper.pyx:100-120): candidates in descending
// score order, a candidate is taken unless within +-(min_base_obs-1) of a taken one, stop at
// num_cpts.  Parallel form: resolve the uncapped greedy as a fixed point
// over the priority DAG (a position is taken iff every higher-priority neighbour is suppressed,
// suppressed iff some higher-priority neighbour is taken), then keep the num_cpts best taken
// positions by exact selection.  Priority = (score, index) descending -- identical to
// np.argsort(score)[::-1] for tie-free scores; ties fall to the higher index.
// One workgroup per read.  state: 0 undecided, 1 taken, 2 suppressed.
__device__ __forceinline__ bool prio_before(double sp, i64 p, double sq, i64 q)
{
    return sp > sq || (sp == sq && p > q);
}

// The greedy, bit-sliced: one tile per WAVEFRONT at a time (no workgroup barriers), 64 lanes x
// 64 consecutive positions; lane l holds, as 64-bit words (bit i <-> position base + 64 l + i):
//   V        position inside the signal
//   G[d]     the neighbour at +d (d = 1..R) outranks this position (ties fall to the higher
//            index, so that is score[p + d] >= score[p]); "this position outranks the one at
//            -d" is the complement over valid pairs, shifted up by d
//   T, S     taken / suppressed so far
// The masks come from one compare (= ballot: the compare writes a lane mask) per 64 positions and
// offset, selected into their lane.  A round is then ~20 bit operations per offset for 4096 positions:
// taken-by-a-higher-neighbour suppresses, no-undecided-higher-neighbour takes; the neighbour
// words cross lanes through DPP.  Decisions are only ever taken from decided neighbours, so
// whatever is decided here is final; the first and last lane of a tile are halo (their outside
// neighbours count as undecided), positions whose dependency chain leaves the tile or outlasts
// the round bound stay 0 and are finished by the global rounds of k_peaks.
// Returns this lane's count of core positions left undecided.
#define PKB_SPAN 4096
#define PKB_CORE (PKB_SPAN - 128)
#define PKB_MAX_ROUNDS 96
struct W64 { u32 lo, hi; };
__device__ __forceinline__ W64 w_and(W64 a, W64 b) { return {a.lo & b.lo, a.hi & b.hi}; }
__device__ __forceinline__ W64 w_or(W64 a, W64 b) { return {a.lo | b.lo, a.hi | b.hi}; }
__device__ __forceinline__ W64 w_andn(W64 a, W64 b) { return {a.lo & ~b.lo, a.hi & ~b.hi}; }
// bit i <- bit i + d of the 128-bit (next:cur);  bit i <- bit i - d of (cur:prev);  0 < d < 32
__device__ __forceinline__ W64 w_down(W64 cur, W64 next, int d)
{
    return {__builtin_amdgcn_alignbit(cur.hi, cur.lo, d), __builtin_amdgcn_alignbit(next.lo, cur.hi, d)};
}
__device__ __forceinline__ W64 w_up(W64 cur, W64 prev, int d)
{
    return {__builtin_amdgcn_alignbit(cur.lo, prev.hi, 32 - d), __builtin_amdgcn_alignbit(cur.hi, cur.lo, 32 - d)};
}
__device__ __forceinline__ W64 w_from_lane_below(W64 x, W64 lane0) // lane l <- lane l-1
{
    return {(u32)__builtin_amdgcn_update_dpp((int)lane0.lo, (int)x.lo, 0x138, 0xf, 0xf, false),
            (u32)__builtin_amdgcn_update_dpp((int)lane0.hi, (int)x.hi, 0x138, 0xf, 0xf, false)};
}
__device__ __forceinline__ W64 w_from_lane_above(W64 x, W64 lane63) // lane l <- lane l+1
{
    return {(u32)__builtin_amdgcn_update_dpp((int)lane63.lo, (int)x.lo, 0x130, 0xf, 0xf, false),
            (u32)__builtin_amdgcn_update_dpp((int)lane63.hi, (int)x.hi, 0x130, 0xf, 0xf, false)};
}
__device__ __forceinline__ W64 w_ballot_to_lane(W64 old, u64 m, int g) // word of lane g <- m
{
    // (v_writelane_b32 would do this in two instructions, but on gfx9 its lane select has to
    // go through m0 next to the SGPR value -- not worth an inline-asm clobber of m0)
    const bool mine = (int)(threadIdx.x & 63) == g;
    return {mine ? (u32)m : old.lo, mine ? (u32)(m >> 32) : old.hi};
}
// Taken scores of the core positions are appended to `dense` on the way (order is irrelevant to
// the selection that follows): one LDS counter bump per 64 positions.
template <int R>
__device__ i64 peaks_bits(const double *s, unsigned char *st, i64 ns, double *dense, u32 *n_dense,
                          double &mn, double &mx, i64 *tdbg = nullptr)
{
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 4
    // wave 0's cycles in the three parts of a tile, rounds and tiles (sums over its tiles)
#define TILE_T(i_) do { if (tdbg && threadIdx.x == 0) { const i64 t_ = (i64)__builtin_readcyclecounter(); tdbg[i_] += t_ - tile_t_; tile_t_ = t_; } } while (0)
    i64 tile_t_ = (i64)__builtin_readcyclecounter();
#else
#define TILE_T(i_) do { } while (0)
#endif
    static_assert(R >= 1 && R < 32, "exclusion radius");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i64 left = 0;
    const i64 n_tiles = (ns + PKB_CORE - 1) / PKB_CORE;
    const W64 zero = {0u, 0u}, ones = {~0u, ~0u};
    for (i64 tile = wave; tile < n_tiles; tile += SEL_NT / 64) {
        const i64 g0 = tile * PKB_CORE - 64; // position of bit 0 of lane 0
        W64 V = zero, G[R + 1];
#pragma unroll
        for (int d = 1; d <= R; d++) G[d] = zero;
        // masks: group g = positions g0 + 64 g + lane.  Every score is loaded once; the
        // neighbour at +d is the value d lanes up: one wave_shl:1 DPP step per offset, lane 63
        // taking lane d-1 of the next group.  Groups come in chunks of GC whose loads are in
        // flight together, the next chunk's issued before this one's masks are built.
#ifndef PKB_GC
#define PKB_GC 4 // (4 / 4: no register spills at 128 VGPRs; radius 2 fits 80 = three workgroups per CU)
#endif
        constexpr int GC = PKB_GC;
        auto load_group = [&](int g) {
            const i64 p = g0 + 64 * (i64)g + lane;
            return s[p < 0 ? 0 : (p >= ns ? ns - 1 : p)];
        };
        double cu[GC], nx[GC];
#pragma unroll
        for (int u = 0; u < GC; u++) cu[u] = load_group(u);
        for (int c0 = 0; c0 < 64; c0 += GC) {
#pragma unroll
            for (int u = 0; u < GC; u++) // (group 64: the first R positions past the tile)
                nx[u] = (u == 0 || c0 + GC < 64) ? load_group(c0 + GC + u) : 0.0;
#pragma unroll
            for (int u = 0; u < GC; u++) {
                const int g = c0 + u;
                const i64 p = g0 + 64 * (i64)g + lane;
                const bool vp = p >= 0 && p < ns;
                const double sp = cu[u], up = u + 1 < GC ? cu[u + 1 < GC ? u + 1 : 0] : nx[0];
                V = w_ballot_to_lane(V, __ballot(vp), g);
                double sq = sp;
#pragma unroll
                for (int d = 1; d <= R; d++) {
                    const double edge = __hiloint2double(
                        __builtin_amdgcn_readlane(__double2hiint(up), d - 1),
                        __builtin_amdgcn_readlane(__double2loint(up), d - 1));
                    sq = wave_shl1_f64(sq, edge); // score at p + d
                    G[d] = w_ballot_to_lane(G[d], __ballot(vp && p + d < ns && sq >= sp), g);
                }
            }
#pragma unroll
            for (int u = 0; u < GC; u++) cu[u] = nx[u];
        }
        TILE_T(0);
        // validity of the words just outside the tile
        const i64 after = ns - (g0 + PKB_SPAN); // positions of the signal past the tile
        const W64 v_after = after >= 64 ? ones : (after <= 0 ? zero :
            (after >= 32 ? W64{~0u, (1u << (after - 32)) - 1u} : W64{(1u << after) - 1u, 0u}));
        const W64 v_before = g0 > 0 ? ones : zero;
        const W64 Vn = w_from_lane_above(V, v_after), Vp = w_from_lane_below(V, v_before);
        (void)Vp;
        // Hp[d]: neighbour +d outranks me (= G[d]);  Hm[d]: neighbour -d outranks me
        W64 Hm[R + 1];
#pragma unroll
        for (int d = 1; d <= R; d++) {
            const W64 pv = w_and(V, w_down(V, Vn, d));            // both ends of the pair valid
            const W64 x = w_andn(pv, G[d]);                        // pair (p, p+d): p outranks p+d
            // below the tile nothing is known: the missing neighbours count as outranking
            const W64 x_prev = w_from_lane_below(x, v_before);
            Hm[d] = w_and(V, w_up(x, x_prev, d));
        }
        W64 T = zero, S = zero, U = V;
        const W64 u_before = v_before, u_after = v_after;        // outside: undecided where valid
        for (int round = 0; round < PKB_MAX_ROUNDS; round++) {
            const W64 Tn = w_from_lane_above(T, zero), Tp = w_from_lane_below(T, zero);
            const W64 Un = w_from_lane_above(U, u_after), Up = w_from_lane_below(U, u_before);
            W64 at = zero, au = zero;
#pragma unroll
            for (int d = 1; d <= R; d++) {
                at = w_or(at, w_or(w_and(G[d], w_down(T, Tn, d)), w_and(Hm[d], w_up(T, Tp, d))));
                au = w_or(au, w_or(w_and(G[d], w_down(U, Un, d)), w_and(Hm[d], w_up(U, Up, d))));
            }
            const W64 nS = w_and(U, at);
            const W64 nT = w_andn(w_andn(U, at), au);
            T = w_or(T, nT); S = w_or(S, nS);
            U = w_andn(w_andn(U, nT), nS);
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 4
            if (tdbg && threadIdx.x == 0) tdbg[3]++;
#endif
            if (__ballot((nS.lo | nS.hi | nT.lo | nT.hi) != 0) == 0) break;
        }
        TILE_T(1);
        // core lanes 1..62 -> one state byte per position, taken scores -> dense list.  The
        // slot of every taken position is known up front (prefix of the per-word counts: one
        // counter bump per tile), and the scores of four groups are fetched before they are used.
        const int cl = (lane >= 1 && lane < 63) ? __popc(T.lo) + __popc(T.hi) : 0;
        int inc = cl;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        const int excl = inc - cl;
        const int total = __builtin_amdgcn_readlane(inc, 63);
        u32 dbase = 0;
        if (total) { // (taken bits only ever sit on valid positions)
            if (lane == 0) dbase = atomicAdd(n_dense, (u32)total);
            dbase = (u32)__builtin_amdgcn_readfirstlane((int)dbase);
        }
        auto out_group = [&](int g, double v) {
            // the group's taken / suppressed words as lane predicates (inverse ballot: the word
            // becomes the exec mask) and the rank of a taken lane inside its word (v_mbcnt)
            const u32 tl = __builtin_amdgcn_readlane((int)T.lo, g), th = __builtin_amdgcn_readlane((int)T.hi, g);
            const u32 sl = __builtin_amdgcn_readlane((int)S.lo, g), sh = __builtin_amdgcn_readlane((int)S.hi, g);
            const bool tb = __builtin_amdgcn_inverse_ballot_w64(((u64)th << 32) | tl);
            const bool sb = __builtin_amdgcn_inverse_ballot_w64(((u64)sh << 32) | sl);
            const i64 p = g0 + 64 * g + lane;
            if (p < ns) st[p] = (unsigned char)(tb ? 1 : (sb ? 2 : 0)); // (never both)
            if (tb) {
                const u32 below = __builtin_amdgcn_mbcnt_hi(th, __builtin_amdgcn_mbcnt_lo(tl, 0u));
                dense[dbase + (u32)__builtin_amdgcn_readlane(excl, g) + below] = v;
                mn = v < mn ? v : mn; mx = v > mx ? v : mx;
            }
        };
        auto score_at = [&](int g) { const i64 p = g0 + 64 * g + lane; return s[p < ns ? p : ns - 1]; };
#ifndef PKB_OU
#define PKB_OU 4
#endif
        constexpr int OU = PKB_OU; // groups whose scores are in flight together
        int g = 1;
        for (; g + OU - 1 < 63; g += OU) {
            double v[OU];
#pragma unroll
            for (int u = 0; u < OU; u++) v[u] = score_at(g + u);
#pragma unroll
            for (int u = 0; u < OU; u++) out_group(g + u, v[u]);
        }
        {
            double v[OU];
#pragma unroll
            for (int u = 0; u < OU; u++) v[u] = g + u < 63 ? score_at(g + u) : 0.0;
#pragma unroll
            for (int u = 0; u < OU; u++) if (g + u < 63) out_group(g + u, v[u]);
        }
        if (lane >= 1 && lane < 63) left += __popc(U.lo) + __popc(U.hi);
        TILE_T(2);
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 4
        if (tdbg && threadIdx.x == 0) tdbg[4]++;
#endif
    }
#undef TILE_T
    return left;
}

// RT: the exclusion radius min_obs_per_base - 1 the tile code is compiled for (2: DNA default,
// 5: RNA default -- one kernel each, so that the narrow one does not carry the wide one's
// registers); 0: any radius, everything through the global rounds.
template <int RT>
__global__ __launch_bounds__(SEL_NT, 4) void k_peaks(ReadState *rs, const DevParams *dp,
    const double *score, unsigned char *state, double *dense, i64 *valid_cpts, int ttest,
    int only_flagged = 0, int form = TBA_ED_FORM_SCORES_PEAKS)
{
    __shared__ BucketSmem sm;
    __shared__ i64 s_w[SEL_NT / 64];
    __shared__ i64 s_idx_thr;
    __shared__ u32 s_ndense;
    ReadState &r = rs[blockIdx.x];
    if (r.status != TBA_OK) return;
    if (only_flagged && !r.ed_flag) return; // k_detect / k_pick (k_detect.h) finished this read
    const int tid = threadIdx.x;
    // (the long reads of any batch are scanned a workgroup each, k_long.h: the latency form)
    if (tid == 0) r.ed_form = form == TBA_ED_FORM_SCORES_PEAKS && r.is_long && 2 * dp->p.running_stat_width <= 64 ? TBA_ED_FORM_WG_SCAN_PEAKS : form;
    const i64 w = dp->p.running_stat_width, m = dp->p.min_obs_per_base;
    const i64 ns = ttest ? r.n_raw - 2 * w : r.n_raw + 1 - 2 * w;
    const i64 num_cands = ttest ? ns : ns - 2 * w;
    const i64 num_cpts = r.num_events;
    const double *s = score + r.raw_off;
    unsigned char *st = state + r.raw_off;
    double *dn = dense + r.raw_off + blockIdx.x; // scratch: taken scores, densely packed
    i64 *cpts = valid_cpts + r.ev_off;
    if (ns <= 0 || num_cpts <= 0) { if (tid == 0) r.status = TBA_INTERNAL; return; }
    double mn = INFINITY, mx = -INFINITY; // range of the taken scores
    bool fused = false, had_leftovers = false;

    // Phase 1: resolve the greedy inside LDS tiles (core PK_CORE positions + PK_HALO each side).
    // A position is only decided from neighbours that are themselves decided, so every decision
    // made here is final; positions whose dependency chain leaves the tile stay undecided (0)
    // and are finished by the global rounds below (rare: chains are a few positions long).
    TBA_PHASE_T0(1);
    {
        i64 left_undecided = 0;
        if (tid == 0) s_ndense = 0;
        __syncthreads();
        if constexpr (RT > 0) {
            if (m - 1 != RT) { if (tid == 0) r.status = TBA_INTERNAL; return; } // (host dispatch)
            left_undecided = peaks_bits<RT>(s, st, ns, dn, &s_ndense, mn, mx, r.dbg);
            fused = true;
            __syncthreads();
        } else { // unusual min_obs_per_base: everything goes through the global rounds
            for (i64 p = tid; p < ns; p += SEL_NT) st[p] = 0;
            left_undecided = 1;
            __syncthreads();
        }
        left_undecided = block_sum_i64(left_undecided, &sm.rad);
        had_leftovers = left_undecided > 0;
        TBA_PHASE(1, 0);
        // Phase 2: global rounds for whatever the tiles could not settle
        for (i64 round = 0; left_undecided > 0 && round <= ns; round++) {
            i64 undecided = 0;
            for (i64 p = tid; p < ns; p += SEL_NT) {
                if (st[p] != 0) continue;
                const double sp_ = s[p];
                bool any_taken = false, any_undecided = false;
                i64 q0 = p - m + 1 < 0 ? 0 : p - m + 1, q1 = p + m - 1 >= ns ? ns - 1 : p + m - 1;
                for (i64 q = q0; q <= q1; q++) {
                    if (q == p) continue;
                    if (!prio_before(s[q], q, sp_, p)) continue;
                    unsigned char sq = st[q];
                    if (sq == 1) any_taken = true;
                    else if (sq == 0) any_undecided = true;
                }
                if (any_taken) st[p] = 2;
                else if (!any_undecided) st[p] = 1;
                else undecided++;
            }
            __threadfence_block();
            left_undecided = block_sum_i64(undecided, &sm.rad);
        }
    }
    TBA_PHASE(1, 1);
    // taken scores -> dense array (+ their range): done by the tiles, unless some positions had
    // to be settled by the global rounds (then one ordered compaction pass redoes it)
    i64 n_taken;
    if (fused && !had_leftovers) {
        __threadfence_block();
        __syncthreads();
        n_taken = s_ndense;
    } else {
        mn = INFINITY; mx = -INFINITY;
        n_taken = block_compact(
            ns, [&](i64 p) { return st[p]; }, [&](i64, unsigned char t) { return t == 1; },
            [&](i64 p, i64 o) { double v = s[p]; dn[o] = v; mn = v < mn ? v : mn; mx = v > mx ? v : mx; },
            s_w);
    }
    if (n_taken < num_cpts) { if (tid == 0) r.status = TBA_FEWER_CPTS; return; }
    for (int mm = 32; mm >= 1; mm >>= 1) {
        double a = shfl_xor_f64(mn, mm), b2 = shfl_xor_f64(mx, mm);
        mn = a < mn ? a : mn; mx = b2 > mx ? b2 : mx;
    }
    if ((tid & 63) == 0) { sm.redd[2 * (tid >> 6)] = mn; sm.redd[2 * (tid >> 6) + 1] = mx; }
    __syncthreads();
    mn = sm.redd[0]; mx = sm.redd[1];
    for (int q = 1; q < SEL_NT / 64; q++) {
        mn = sm.redd[2 * q] < mn ? sm.redd[2 * q] : mn;
        mx = sm.redd[2 * q + 1] > mx ? sm.redd[2 * q + 1] : mx;
    }
    __syncthreads();
    TBA_PHASE(1, 2);
    // score of the num_cpts-th best taken position (ascending rank n_taken - num_cpts)
    const double tval = block_kth([&](i64 i) { return dn[i]; }, n_taken, n_taken - num_cpts, mn,
                                  mx, &sm);
    __syncthreads();
    TBA_PHASE(1, 3);
    // one pass: ordered compaction of the picks (the .sort() of tombo_helper.py:76-82), taking
    // every taken position at or above the threshold score, and on the way the counts that tell
    // whether that was right: taken above / at the threshold, all positions above / at it.
    // Positions AT the threshold score are also histogrammed by position bucket (taken ones and
    // all of them): on quantised DAC input exact ties are the rule and the tie rule (priority to
    // the higher index) is resolved from these histograms without another pass.
    int tshift = 0;
    while ((ns >> tshift) >= 2048) tshift++;
    u32 *h_tk = sm.hist, *h_all = sm.hist + 2048;
    for (int b = tid; b < 4096; b += SEL_NT) sm.hist[b] = 0;
    __syncthreads();
    i64 c_gt = 0, c_eq = 0, a_gt = 0, a_eq = 0;
    struct ScoreState { double v; unsigned char t; };
    block_compact(
        ns, [&](i64 p) { return ScoreState{s[p], st[p]}; },
        [&](i64 p, ScoreState e) {
            const double v = e.v;
            const bool tk = e.t == 1;
            a_gt += v > tval; a_eq += v == tval;
            c_gt += tk && v > tval; c_eq += tk && v == tval;
            if (v == tval) {
                atomicAdd(&h_all[p >> tshift], 1u);
                if (tk) atomicAdd(&h_tk[p >> tshift], 1u);
            }
            return tk && v >= tval;
        },
        [&](i64 p, i64 o) { if (o < num_cpts) cpts[o] = p + w; }, s_w);
    c_gt = block_sum_i64(c_gt, &sm.rad);
    c_eq = block_sum_i64(c_eq, &sm.rad);
    a_gt = block_sum_i64(a_gt, &sm.rad);
    a_eq = block_sum_i64(a_eq, &sm.rad);
    const i64 need_eq = num_cpts - c_gt;
    i64 before = a_gt;    // rank of the last pick in the argsort order
    if (need_eq < c_eq || a_eq > 1) {
        // exact ties on the threshold score: the picks at that score are the need_eq
        // highest-index taken positions.  Wave 0 walks the bucket histogram from the top to the
        // bucket holding the need_eq-th of them, then that bucket's positions 64 at a time
        // (lane 0 = highest index), counting on the way how many positions of that score -- taken
        // or not -- outrank the last pick.
        __shared__ i64 s_extra;
        if (tid < 64) {
            const int lane = tid;
            i64 left = need_eq, extra = 0;
            int bq = 0;
            for (int b = (int)((ns - 1) >> tshift); b >= 0; b--) {
                const u32 c = h_tk[b];
                if ((i64)c >= left) { bq = b; break; }
                left -= c; extra += h_all[b];
            }
            const i64 lo_p = (i64)bq << tshift;
            i64 hi_p = ((i64)bq + 1) << tshift;
            hi_p = hi_p < ns ? hi_p : ns;
            i64 thr = lo_p;
            for (i64 top = hi_p; top > lo_p; top -= 64) {
                const i64 p = top - 1 - lane;
                const bool ok = p >= lo_p;
                const bool eq = ok && s[ok ? p : lo_p] == tval;
                const bool tk = eq && st[ok ? p : lo_p] == 1;
                const u64 mt = __ballot(tk), ma = __ballot(eq);
                const int c = __popcll(mt);
                if ((i64)c >= left) {
                    u64 m = mt;
                    for (i64 q = 1; q < left; q++) m &= m - 1;          // drop the first left-1 picks
                    const int pos = __ffsll((unsigned long long)m) - 1;  // lane of the last pick
                    thr = top - 1 - pos;
                    extra += __popcll(ma & ((1ull << pos) - 1ull));
                    break;
                }
                left -= c; extra += __popcll(ma);
            }
            if (lane == 0) { s_idx_thr = thr; s_extra = extra; }
        }
        __syncthreads();
        const i64 idx_thr = s_idx_thr; // lowest-index pick at the threshold score
        before = a_gt + s_extra;
        if (need_eq < c_eq)
            block_compact(
                ns, [&](i64 p) { return ScoreState{s[p], st[p]}; },
                [&](i64 p, ScoreState e) { return e.t == 1 && (e.v > tval || (e.v == tval && p >= idx_thr)); },
                [&](i64 p, i64 o) { if (o < num_cpts) cpts[o] = p + w; }, s_w);
    }
    // the reference raises when rank + 1 >= num_cands (cand_idx is advanced past the pick before
    // the bound check, _c_helper.pyx:116-118)
    if (num_cpts > 1 && before + 1 >= num_cands) { if (tid == 0) r.status = TBA_FEWER_CPTS; return; }
    TBA_PHASE(1, 4);
    TBA_PHASE(1, 5);
    TBA_PHASE_END(1);
    if (tid == 0) { r.n_cpts = num_cpts; r.n_ev = num_cpts - 1; }
}

// ts.remove_stall_cpts (tombo_stats.py:1576-1597): a change point is dropped when it lies
// strictly inside the first stall interval whose end is >= the change point (the reference's
// forward walk; interval ends ascend).  One thread per read (RNA only, few intervals).
__global__ __launch_bounds__(SEL_NT) void k_remove_stalls(ReadState *rs, i64 n_reads,
    const i64 *stall_ints, i64 *valid_cpts, double *scratch)
{
    // workgroup per read: the interval the reference's forward walk would be looking at when it
    // reaches a change point is the first one whose end is >= the point (the last interval once
    // the point is past every end) -- a binary search; survivors are compacted in order through
    // the read's scratch slice
    __shared__ i64 s_w[SEL_NT / 64];
    (void)n_reads;
    ReadState &r = rs[blockIdx.x];
    if (r.status != TBA_OK || r.n_stall == 0) return;
    const i64 *st = stall_ints + 2 * r.stall_off;
    i64 *c = valid_cpts + r.ev_off;
    i64 *tmp = (i64 *)(scratch + r.raw_off + blockIdx.x);
    const i64 ns = r.n_stall, n = r.n_cpts;
    const i64 out = block_compact(
        n, [&](i64 i) { return c[i]; },
        [&](i64, i64 v) {
            i64 lo = 0, hi = ns - 1; // first interval with end >= v, else the last one
            while (lo < hi) { const i64 mid = (lo + hi) >> 1; if (st[2 * mid + 1] >= v) hi = mid; else lo = mid + 1; }
            return !(st[2 * lo] < v && v < st[2 * lo + 1]);
        },
        [&](i64 i, i64 o) { tmp[o] = c[i]; }, s_w);
    __threadfence_block();
    __syncthreads();
    for (i64 i = threadIdx.x; i < out; i += SEL_NT) c[i] = tmp[i];
    if (threadIdx.x == 0) {
        r.n_cpts = out;
        r.n_ev = out - 1;
        if (out < 2) r.status = TBA_INTERNAL;
    }
}

// c_new_means (_c_helper.pyx:59-71) over the event boundaries: sequential sum, one divide.
// grid: (blocks, reads)
// CAP: samples a wavefront stages per step (wave_segment_sums): 448 = 64 events of ~5 samples (DNA);
// RNA events are ~15 samples, at 448 a step would be 16 events on a quarter of the lanes: 1 280.
template <class RT, int CAP = 448>
// scale_events != 0: only the events ts.get_scale_values_from_events looks at are needed (the
// first min(rna_scale_num_events, int(frac * n_cpts)) - 1, tombo_stats.py:220-224).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, CAP > 448 ? 4 : 8)))
void k_event_means(const ReadState *rs, const DevParams *dp,
    const RT *sig, const i64 *valid_cpts, double *event_means, int scale_events)
{
    const ReadState &r = rs[blockIdx.y];
    if (r.status != TBA_OK) return;
    const RawSamples<RT> x{sig + r.raw_off};
    const i64 *c = valid_cpts + r.ev_off;
    double *em = event_means + r.ev_off;
    i64 n = r.n_cpts - 1;
    if (scale_events) {
        const tba_opts &o = dp->o;
        i64 ne = o.rna_scale_num_events;
        if ((double)r.n_cpts * o.rna_scale_max_frac_events < (double)ne)
            ne = (i64)((double)r.n_cpts * o.rna_scale_max_frac_events);
        if (ne > r.n_cpts) ne = r.n_cpts;
        n = ne - 1 < n ? (ne - 1 > 0 ? ne - 1 : 0) : n;
    }
    __shared__ double s_seg[4 * CAP];
    const int wave = threadIdx.x >> 6;
    struct None {};
    wave_segment_sums<CAP>(x, c, n, (i64)blockIdx.x * 4 + wave, (i64)gridDim.x * 4, s_seg + wave * CAP,
                      nullptr, [](double v) { return v; }, [](i64) { return None{}; },
                      [&](i64 e, double s, i64 len, None) { em[e] = s / (double)len; },
                      (double)r.n_raw / (double)(r.n_cpts > 1 ? r.n_cpts - 1 : 1));
}

// ts.get_scale_values_from_events (tombo_stats.py:217-233): median / MAD of the first
// min(10000, int(0.75 * n_cpts)) - 1 raw event means; one workgroup per read (RNA).
// event_means must hold c_new_means(raw, valid_cpts) on entry.
__global__ __launch_bounds__(SEL_NT) void k_rna_event_scale(ReadState *rs, const DevParams *dp,
    const double *event_means)
{
    __shared__ SelectSmem sm;
    ReadState &r = rs[blockIdx.x];
    if (r.status != TBA_OK) return;
    if (r.sv_flags & 1) return; // scale values given: not used
    const tba_opts &o = dp->o;
    if (o.has_const_scale || !o.use_rna_event_scale) return;
    i64 ne = o.rna_scale_num_events;
    if ((double)r.n_cpts * o.rna_scale_max_frac_events < (double)ne)
        ne = (i64)((double)r.n_cpts * o.rna_scale_max_frac_events);
    if (ne > r.n_cpts) ne = r.n_cpts;
    if (ne < 2 || !o.has_outlier_thresh) { if (threadIdx.x == 0) r.status = TBA_INTERNAL; return; }
    const double *em = event_means + r.ev_off;
    double med = block_median([&](i64 i) { return f64_key(em[i]); }, ne - 1, &sm);
    double mad = block_median([&](i64 i) { return f64_key(fabs(em[i] - med)); }, ne - 1, &sm);
    if (threadIdx.x == 0) {
        r.shift = med; r.scale = mad; r.lower = -o.outlier_thresh; r.upper = o.outlier_thresh;
        r.has_lims = 1;
    }
}

// TomboModel.get_exp_levels_from_seq (tombo_stats.py:834-862): k-mer code -> level mean / sd.
// grid: (blocks, reads)
// defer != 0 (the side stream of a full run: this kernel runs beside the segmentation stage): an
// invalid base is only noted (ReadState.bad_seq) and becomes the read's status where the stage stands in
// the reference's order (k_seq_status) -- a read that ALSO fails in segmentation fails with that error.
__global__ __launch_bounds__(256) void k_ref_levels(ReadState *rs, const DevParams *dp,
    const uint8_t *seq, const double *kmer_means, const double *kmer_sds, double *ref_means,
    double *ref_sds, int defer = 0)
{
    ReadState &r = rs[blockIdx.y];
    if (r.status != TBA_OK) return;
    const i64 K = dp->kmer_width;
    const uint8_t *s = seq + r.seq_off;
    bool bad = false;
    // Two bases per thread: their K + 1 <= 8 codes in one (unaligned) 8-byte load instead of 2 K byte loads, the two
    // levels and the two sds as 16-byte stores (st2: 8-byte alignment is enough).  The generic form below takes what
    // the pairs leave: every base when K > 7, and the last bases of the read (an 8-byte load there would run past
    // the read's codes).
    const i64 n_codes = r.B + K - 1;
    i64 n_paired = 0;                                    // bases [0, n_paired) are done in pairs
    if (K <= 7) { n_paired = n_codes - 8 + 2; n_paired = n_paired < 0 ? 0 : (n_paired > r.B ? r.B : n_paired); n_paired &= ~(i64)1; }
    for (i64 i = 2 * ((i64)blockIdx.x * 256 + threadIdx.x); i < n_paired; i += 2 * (i64)gridDim.x * 256) {
        u64 w;                                           // (i + 8 <= n_codes for every i < n_paired)
        __builtin_memcpy(&w, s + i, 8);
        if (w & 0xfcfcfcfcfcfcfcfcull) {                 // a code > 3 among the eight bytes: only the K + 1 used ones count
            for (i64 j = 0; j <= K; j++) if (((w >> (8 * j)) & 0xff) > 3) bad = true;
        }
        i64 c0 = 0, c1 = 0;
        for (i64 j = 0; j < K; j++) {
            c0 = c0 * 4 + ((w >> (8 * j)) & 3);
            c1 = c1 * 4 + ((w >> (8 * (j + 1))) & 3);
        }
        st2(ref_means + r.ref_off + i, kmer_means[c0], kmer_means[c1]);
        st2(ref_sds + r.ref_off + i, kmer_sds[c0], kmer_sds[c1]);
    }
    for (i64 i = n_paired + (i64)blockIdx.x * 256 + threadIdx.x; i < r.B; i += (i64)gridDim.x * 256) {
        i64 code = 0;
        for (i64 j = 0; j < K; j++) {
            uint8_t b = s[i + j];
            if (b > 3) bad = true;
            code = code * 4 + (b & 3);
        }
        ref_means[r.ref_off + i] = kmer_means[code];
        ref_sds[r.ref_off + i] = kmer_sds[code];
    }
    if (bad) { if (defer) r.bad_seq = 1; else r.status = TBA_INVALID_SEQ; }
}
__global__ __launch_bounds__(64) void k_seq_status(ReadState *rs, i64 n_reads)
{
    const i64 ri = (i64)blockIdx.x * 64 + threadIdx.x;
    if (ri < n_reads && rs[ri].status == TBA_OK && rs[ri].bad_seq) rs[ri].status = TBA_INVALID_SEQ;
}
