// k_prep_raw.h -- the worker's per-read preparation on the device:
//   _resquiggle_worker.adjust_map_res   tombo/resquiggle.py:1506-1530  (RNA flip, stall detection)
//   ts.identify_stalls, mean-window     tombo/tombo_stats.py:269-368
// The cumulative sum under the stall metric is k_cumsum_scores<.., RT, 1> (k_segment.h): its
// left-to-right float64 order is part of the metric's bits for float input; int16 DAC input --
// exact integer sums in any order -- has a path of its own without a cumulative sum in memory
// (k_stall_metric_i16, below).
#pragma once
#include "tba_common.h"

// raw_signal[::-1] in place (resquiggle.py:1516).  grid: (blocks, reads)
template <class RT>
__global__ __launch_bounds__(256) void k_reverse_raw(const ReadState *rs, RT *raw)
{
    const ReadState &r = rs[blockIdx.y];
    RT *x = raw + r.raw_off;
    const i64 n = r.n_raw, half = n / 2;
    for (i64 i0 = (i64)blockIdx.x * 1024 + threadIdx.x; i0 < half; i0 += (i64)gridDim.x * 1024) {
        RT a[4], b[4]; // all eight loads first
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const i64 i = i0 + u * 256;
            if (i < half) { a[u] = x[i]; b[u] = x[n - 1 - i]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const i64 i = i0 + u * 256;
            if (i < half) { x[i] = b[u]; x[n - 1 - i] = a[u]; }
        }
    }
}

// bit array of one read inside a shared word buffer: read i owns [raw_off / 64 + i, ... + n_raw / 64]
__device__ __forceinline__ i64 stall_word_base(const ReadState &r, i64 read_index) { return (r.raw_off >> 6) + read_index; }

// compute_running_mean_diffs (tombo_stats.py:273-301) + the threshold test (:332-334), one lane
// per position of the metric array, 64 positions = one ballot word of "metric <= threshold".
//   c[k] = sum of the first k samples (csum + raw_off + read index, n_raw + 1 entries)
//   moving_average[j] = (c[j + mw] - c[j]) / mw                      (:277-283; j = 0: c[mw] - 0)
//   offsets[k][p] = moving_average[mw k + p], p < n - ws + 1          (:286-292)
//   diff_sums = diffs[0].copy(); for d in diffs: diff_sums += d       (:298-300: diffs[0] twice)
//   metric[ws // 2 ... + p] = diff_sums / len(diffs), NaN elsewhere   (:312-327)
// The two divisions have loop-invariant divisors: div_by_recip (tba_common.h) is IEEE division.
// NW > 0: n_windows as a compile-time constant (7 = MEAN_STALL_PARAMS), else <= 16 at run time.
#define SM_T 2048   // metric positions per workgroup step
#define SM_MAXW 1024 // widest window staged in LDS (wider: straight from memory)
#ifndef TBA_STALL_WAVES
#define TBA_STALL_WAVES 4   // wavefronts per SIMD the register allocation leaves room for (104 VGPRs; 5 would need <= 96)
#endif
template <int NW>
__global__ __launch_bounds__(256, TBA_STALL_WAVES) void k_stall_metric(const ReadState *rs, const DevParams *dp,
    const double *csum, u64 *bits)
{
    // the SM_T + window_size sums under a chunk of positions are staged in LDS once (a position
    // reads 8 of them 50 apart: from L2 that was 64 bytes per position and 8.6 ms per 10 k RNA reads)
    __shared__ double C[SM_T + SM_MAXW + 1];
    const ReadState &r = rs[blockIdx.y];
    if (r.status != TBA_OK) return;
    const tba_opts &o = dp->o;
    const i64 n = r.n_raw, ws = o.stall_window_size, mw = o.stall_mini_window_size;
    const int nw = NW > 0 ? NW : (int)o.stall_n_windows;
    const i64 n_words = (n + 63) >> 6;
    u64 *bw = bits + stall_word_base(r, blockIdx.y);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const i64 n_pos = n - ws + 1;              // <= 0: read shorter than the window, no metric
    const i64 start_offset = (i64)((double)ws * 0.5);
    const double *c = csum + r.raw_off + blockIdx.y;
    const double dmw = (double)mw, rmw = 1.0 / dmw;
    const double dnd = (double)(nw * (nw - 1) / 2), rnd = 1.0 / dnd;
    const double thr = o.stall_threshold;
    const bool staged = ws <= SM_MAXW;
    const int N = SM_T + (int)(staged ? ws : 0) + 1;
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 11
    // cycles of a workgroup's thread 0 by part, summed over its chunks into the read's dbg[]:
    // 0 the chunk's sums into LDS (memory wait), 1 moving averages, 2 the metric; 3 chunks
    i64 sm_acc[3] = {0, 0, 0}, sm_n = 0, sm_t = (i64)__builtin_readcyclecounter();
#define SM_PH(i_) do { const i64 t_ = (i64)__builtin_readcyclecounter(); sm_acc[i_] += t_ - sm_t; sm_t = t_; } while (0)
#else
#define SM_PH(i_) do { } while (0)
#endif
    for (i64 q0 = (i64)blockIdx.x * SM_T; q0 < (n_words << 6); q0 += (i64)gridDim.x * SM_T) {
        const i64 p0 = q0 - start_offset;
        if (staged) {
            __syncthreads();
            SM_PH(2);
            for (int i = tid; i < N; i += 256) {
                const i64 k = p0 + i;
                C[i] = (k >= 0 && k <= n) ? c[k] : 0.0;
            }
            __syncthreads();
            SM_PH(0);
            // A window mean is shared by the (up to n_windows) positions whose k-th window it is:
            // moving_average[j] is computed ONCE per j of the chunk, in place of c[j] (through
            // registers: the difference reaches mini_window_size slots ahead), and a position reads
            // its means 50 apart.  Same subtraction, same division, a seventh of them (the kernel was
            // bound by its 7 divisions per position: 35 of ~90 float64 instructions).
            constexpr int PER = (SM_T + SM_MAXW + 1 + 255) / 256;
            double mq[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int i = tid + u * 256;
                if (i + (int)mw < N) mq[u] = div_by_recip(C[i + (int)mw] - C[i], dmw, rmw);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int i = tid + u * 256;
                if (i + (int)mw < N) C[i] = mq[u];
            }
            __syncthreads();
            SM_PH(1);
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 11
            sm_n++;
#endif
        }
        for (int wq = wave; wq < SM_T / 64; wq += 4) {
            const i64 w = (q0 >> 6) + wq;
            if (w >= n_words) break;
            const int i = wq * 64 + lane;
            const i64 p = p0 + i;
            const bool valid = p >= 0 && p < n_pos;
            bool below = false;
            if (valid) {
                double m[NW > 0 ? NW : 16];
                double prev = staged ? 0.0 : c[p];
#pragma unroll
                for (int k = 0; k < (NW > 0 ? NW : 16); k++) {
                    if (NW > 0 || k < nw) {
                        if (staged) m[k] = C[i + (int)mw * k];
                        else {
                            const double nxt = c[p + mw * (k + 1)];
                            m[k] = div_by_recip(nxt - prev, dmw, rmw);
                            prev = nxt;
                        }
                    }
                }
                double acc = fabs(m[0] - m[1]); // diffs[0].copy()
#pragma unroll
                for (int ii = 0; ii < (NW > 0 ? NW : 16); ii++)
#pragma unroll
                    for (int jj = ii + 1; jj < (NW > 0 ? NW : 16); jj++)
                        if (NW > 0 || jj < nw) acc = acc + fabs(m[ii] - m[jj]);
                below = div_by_recip(acc, dnd, rnd) <= thr;
            }
            const u64 word = __ballot(below);
            if (lane == 0) bw[w] = word;
        }
    }
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 11
    SM_PH(2);
    if (tid == 0) {
        i64 *dbg = const_cast<ReadState &>(r).dbg;
        for (int k = 0; k < 3; k++) atomicAdd((unsigned long long *)&dbg[k], (unsigned long long)sm_acc[k]);
        atomicAdd((unsigned long long *)&dbg[3], (unsigned long long)sm_n);
        atomicAdd((unsigned long long *)&dbg[4], 1ull);
    }
#endif
#undef SM_PH
}

// Runs of "below" longer than min_consecutive_obs (tombo_stats.py:332-340): one thread per word
// finds the runs that START in it and follows each to its end (almost all runs die within a few
// words; a real stall is followed to its end by one thread).  Qualifying runs are appended to the
// read's slice of `ints` in arrival order (k_stall_merge sorts them).
__global__ __launch_bounds__(256) void k_stall_runs(ReadState *rs, const DevParams *dp, const u64 *bits,
    i64 *ints)
{
    ReadState &r = rs[blockIdx.y];
    if (r.status != TBA_OK) return;
    const i64 n = r.n_raw, n_words = (n + 63) >> 6, min_obs = dp->o.stall_min_consecutive_obs;
    const i64 cap = n / (min_obs + 1) + 2;
    const u64 *bw = bits + stall_word_base(r, blockIdx.y);
    i64 *out = ints + 2 * r.stall_off;
    for (i64 t = (i64)blockIdx.x * 256 + threadIdx.x; t < n_words; t += (i64)gridDim.x * 256) {
        const u64 w = bw[t];
        if (w == 0) continue;
        const u64 prevbit = t > 0 ? bw[t - 1] >> 63 : 0;
        u64 starts = w & ~((w << 1) | prevbit);
        while (starts) {
            const int b = __builtin_ctzll(starts);
            starts &= starts - 1;
            const i64 s = (t << 6) + b;
            const u64 z = b == 63 ? 0 : (~w & (~0ull << (b + 1)));
            i64 e;
            if (z) e = (t << 6) + __builtin_ctzll(z);
            else {
                i64 tw = t + 1;
                while (tw < n_words && bw[tw] == ~0ull) tw++;
                e = tw < n_words ? (tw << 6) + __builtin_ctzll(~bw[tw]) : n;
            }
            if (e - s > min_obs) {
                const i64 slot = (i64)atomicAdd((unsigned long long *)&r.n_stall, 1ull);
                if (slot < cap) { out[2 * slot] = s; out[2 * slot + 1] = e; }
            }
        }
    }
}

// The tail of identify_stalls (tombo_stats.py:346-364): intervals in position order, widened by
// window_size // 2 - edge_buffer on both sides and merged where they then overlap.  One thread per
// read (a read has a handful of stalls).
__global__ __launch_bounds__(64) void k_stall_merge(ReadState *rs, i64 n_reads, const DevParams *dp,
    i64 *ints)
{
    const i64 ri = (i64)blockIdx.x * 64 + threadIdx.x;
    if (ri >= n_reads) return;
    ReadState &r = rs[ri];
    if (r.status != TBA_OK || r.n_stall == 0) return;
    const tba_opts &o = dp->o;
    const i64 cap = r.n_raw / (o.stall_min_consecutive_obs + 1) + 2;
    i64 n = r.n_stall;
    if (n > cap) { r.status = TBA_INTERNAL; r.n_stall = 0; return; } // (cannot happen: see cap)
    i64 *v = ints + 2 * r.stall_off;
    for (i64 i = 1; i < n; i++) { // insertion sort by start
        const i64 s = v[2 * i], e = v[2 * i + 1];
        i64 j = i;
        while (j > 0 && v[2 * (j - 1)] > s) { v[2 * j] = v[2 * (j - 1)]; v[2 * j + 1] = v[2 * (j - 1) + 1]; j--; }
        v[2 * j] = s; v[2 * j + 1] = e;
    }
    const i64 ex = o.stall_window_size / 2 - o.stall_edge_buffer;
    if (ex > 0) {
        i64 m = 0; // merged intervals so far - 1
        v[0] -= ex; v[1] += ex;
        for (i64 i = 1; i < n; i++) {
            const i64 s = v[2 * i] - ex, e = v[2 * i + 1] + ex;
            if (s > v[2 * m + 1]) { m++; v[2 * m] = s; v[2 * m + 1] = e; }
            else v[2 * m + 1] = e;
        }
        n = m + 1;
    }
    r.n_stall = n;
}

// ---------------------------------------------------------------------------------------------
// Device-side Theil-Sen subsample (tba_opts.device_subsample): index t of the subsample is the
// image of t under a keyed pseudo-random permutation of [0, n) -- a 6-round Feistel network over
// the 2 * hb >= log2(n) bits, cycle-walked into the range.  The first 1000 images of a random
// permutation are a uniform draw without replacement, which is what np.random.choice(n, 1000,
// replace=False) (tombo_stats.py:411-416) is; every index is computed independently (counter
// based: no state, any thread order).
__device__ __forceinline__ u32 mix32(u32 h)
{
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ u64 splitmix64(u64 x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ u64 subsample_key(u64 seed, i64 read_index)
{
    return splitmix64(seed ^ splitmix64((u64)read_index + 1));
}
__device__ inline i64 keyed_perm(i64 t, i64 n, u64 key)
{
    if (n <= 1) return 0;
    const int bits = 64 - __builtin_clzll((u64)(n - 1));
    const int hb = (bits + 1) >> 1;
    const u32 mask = hb >= 32 ? 0xffffffffu : ((1u << hb) - 1u);
    const u32 k0 = (u32)key, k1 = (u32)(key >> 32);
    u64 x = (u64)t;
    do {
        u32 L = (u32)(x >> hb) & mask, R = (u32)x & mask;
#pragma unroll
        for (int rd = 0; rd < 6; rd++) {
            const u32 f = mix32((R ^ ((rd & 1) ? k1 : k0)) + (u32)rd * 0x9e3779b9u);
            const u32 nr = L ^ (f & mask);
            L = R; R = nr;
        }
        x = ((u64)L << hb) | R;
    } while (x >= (u64)n);
    return (i64)x;
}

// The same metric for int16 DAC input -- what a FAST5 file holds -- without the cumulative sum in
// memory: integer window sums are exact in any order, so a workgroup takes a chunk of SI_T metric
// positions, loads the SI_T + window_size samples under it into LDS (2 bytes per sample from HBM,
// once), turns them into an exclusive int32 prefix sum there (block scan) and reads every 50-sample
// window sum as a difference of two prefix values.  np.cumsum on an int16 array is int64
// (tombo_stats.py:277), the difference of two of its entries the exact window sum, `/
// mini_window_size` its float64 quotient: the doubles that enter the mean differences are the same
// bits as on the float path.  grid: (blocks, reads); window_size <= SI_MAXW (else the float path).
#define SI_T 2048
#define SI_MAXW 1024
template <int NW>
__global__ __launch_bounds__(256) void k_stall_metric_i16(const ReadState *rs, const DevParams *dp,
    const int16_t *raw, u64 *bits)
{
    __shared__ i32 P[SI_T + SI_MAXW + 8];
    __shared__ double M[SI_T + SI_MAXW + 8]; // moving_average of the chunk, once per window start (see k_stall_metric)
    __shared__ i32 s_wave[4];
    const ReadState &r = rs[blockIdx.y];
    if (r.status != TBA_OK) return;
    const tba_opts &o = dp->o;
    const i64 n = r.n_raw;
    const int ws = (int)o.stall_window_size, mw = (int)o.stall_mini_window_size;
    const int nw = NW > 0 ? NW : (int)o.stall_n_windows;
    const i64 n_words = (n + 63) >> 6;
    u64 *bw = bits + stall_word_base(r, blockIdx.y);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const i64 n_pos = n - ws + 1;
    const i64 start_offset = (i64)((double)ws * 0.5);
    const int16_t *x = raw + r.raw_off;
    const double dmw = (double)mw, rmw = 1.0 / dmw;
    const double dnd = (double)(nw * (nw - 1) / 2), rnd = 1.0 / dnd;
    const double thr = o.stall_threshold;
    const int N = SI_T + ws;                 // samples under a chunk
    const int per = (N + 255) / 256;         // per thread, contiguous
    for (i64 q0 = (i64)blockIdx.x * SI_T; q0 < (n_words << 6); q0 += (i64)gridDim.x * SI_T) {
        const i64 p0 = q0 - start_offset;    // sample index of the chunk's first window start
        __syncthreads();
        for (int i = tid; i < N; i += 256) {
            const i64 k = p0 + i;
            P[i] = (k >= 0 && k < n) ? (i32)x[k] : 0;
        }
        __syncthreads();
        // exclusive prefix sum of P[0..N) in place, P[N] = total
        const int a = tid * per, b = a + per < N ? a + per : N;
        i32 mine = 0;
        for (int i = a; i < b; i++) mine += P[i];
        i32 inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const i32 t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        i32 off = inc - mine;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        __syncthreads(); // (everyone has read its own elements' sum; now they are overwritten)
        for (int i = a; i < b; i++) { const i32 t = P[i]; P[i] = off; off += t; }
        if (tid == 0) P[N] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3]; // the total
        __syncthreads();
        for (int i = tid; i + mw <= N; i += 256) M[i] = div_by_recip((double)(P[i + mw] - P[i]), dmw, rmw);
        __syncthreads();
        // metric of the chunk's positions, 64 per ballot word
        for (int wq = wave; wq < SI_T / 64; wq += 4) {
            const i64 w = (q0 >> 6) + wq;
            if (w >= n_words) break;
            const int i = wq * 64 + lane;   // offset of my window start inside the chunk
            const i64 p = p0 + i;
            const bool valid = p >= 0 && p < n_pos;
            bool below = false;
            if (valid) {
                double m[NW > 0 ? NW : 16];
#pragma unroll
                for (int k = 0; k < (NW > 0 ? NW : 16); k++)
                    if (NW > 0 || k < nw) m[k] = M[i + mw * k];
                double acc = fabs(m[0] - m[1]); // diffs[0].copy()
#pragma unroll
                for (int ii = 0; ii < (NW > 0 ? NW : 16); ii++)
#pragma unroll
                    for (int jj = ii + 1; jj < (NW > 0 ? NW : 16); jj++)
                        if (NW > 0 || jj < nw) acc = acc + fabs(m[ii] - m[jj]);
                below = div_by_recip(acc, dnd, rnd) <= thr;
            }
            const u64 word = __ballot(below);
            if (lane == 0) bw[w] = word;
        }
    }
}
