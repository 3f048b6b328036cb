// k_select.h -- exact order statistics inside one workgroup (radix select on order-preserving
// 64-bit keys).  Used for np.median (normalisation, Theil-Sen) and for the event-detection cap.
// Pure comparisons / counting: bit-exact by construction.
#pragma once
#include "tba_common.h"

#define SEL_NT 256 // threads per workgroup for every kernel that uses these helpers

struct SelectSmem {
    u32 hist[256];
    u64 prefix;
    i64 k;
    i64 n_less;
    i64 n_eq;
    u64 red[SEL_NT / 64];
    double bcast;
};

// one histogram pass contribution with leader aggregation (skewed digits are the common case:
// sign/exponent bytes of a nanopore signal are nearly constant)
__device__ __forceinline__ void hist_add(u32 *hist, bool part, u32 bin)
{
    const int lane = threadIdx.x & 63;
    u64 act = __ballot(part);
    if (act == 0) return;
    int leader = __ffsll((unsigned long long)act) - 1;
    u32 lb = (u32)__shfl((int)bin, leader, 64);
    u64 same = __ballot(part && bin == lb);
    if (part) {
        if (bin == lb) {
            if (lane == leader) atomicAdd(&hist[lb], (u32)__popcll(same));
        } else {
            atomicAdd(&hist[bin], 1u);
        }
    }
}

// k-th smallest key (0-based) of { f(i) : 0 <= i < n }.  Every thread of the workgroup must
// call; results in sm->prefix (key), sm->n_less (#keys < key), sm->n_eq (#keys == key).
template <class F>
__device__ void block_select(F f, i64 n, i64 k, SelectSmem *sm)
{
    const int tid = threadIdx.x;
    if (tid == 0) { sm->prefix = 0; sm->k = k; sm->n_less = 0; sm->n_eq = 0; }
    __syncthreads();
    for (int pass = 7; pass >= 0; pass--) {
        sm->hist[tid & 255] = 0; // SEL_NT == 256
        __syncthreads();
        const u64 prefix = sm->prefix;
        const int sh = 8 * pass;
        const u64 himask = pass == 7 ? 0ull : (~0ull << (sh + 8));
        for (i64 base = 0; base < n; base += SEL_NT) {
            i64 i = base + tid;
            bool part = false;
            u32 bin = 0;
            if (i < n) {
                u64 key = f(i);
                part = (key & himask) == prefix;
                bin = (u32)((key >> sh) & 255);
            }
            hist_add(sm->hist, part, bin);
        }
        __syncthreads();
        if (tid < 64) {
            // wave 0 locates the bin holding rank k: 4 bins per lane + shuffle scan
            u32 h0 = sm->hist[4 * tid], h1 = sm->hist[4 * tid + 1], h2 = sm->hist[4 * tid + 2],
                h3 = sm->hist[4 * tid + 3];
            i64 c = (i64)h0 + h1 + h2 + h3;
            i64 inc = c;
            for (int d = 1; d < 64; d <<= 1) {
                i64 t = shfl_i64(inc, tid - d < 0 ? 0 : tid - d);
                if (tid >= d) inc += t;
            }
            i64 exc = inc - c;
            i64 kk = sm->k;
            if (exc <= kk && kk < inc) {
                i64 r = kk - exc;
                u32 b;
                i64 less = exc;
                u32 cnt;
                if (r < h0) { b = 0; cnt = h0; }
                else if (r < (i64)h0 + h1) { b = 1; less += h0; cnt = h1; }
                else if (r < (i64)h0 + h1 + h2) { b = 2; less += (i64)h0 + h1; cnt = h2; }
                else { b = 3; less += (i64)h0 + h1 + h2; cnt = h3; }
                sm->prefix = prefix | ((u64)(4 * tid + b) << sh);
                sm->k = kk - less;
                sm->n_less += less;
                sm->n_eq = cnt;
            }
        }
        __syncthreads();
    }
}

// smallest key strictly greater than `key` (all threads call; result in sm->prefix;
// ~0 if none)
template <class F>
__device__ void block_min_greater(F f, i64 n, u64 key, SelectSmem *sm)
{
    const int tid = threadIdx.x;
    u64 best = ~0ull;
    for (i64 i = tid; i < n; i += SEL_NT) {
        u64 kx = f(i);
        if (kx > key && kx < best) best = kx;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        u64 o = (u64)shfl_i64((i64)best, (tid & 63) ^ m);
        if (o < best) best = o;
    }
    if ((tid & 63) == 0) sm->red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        u64 b = sm->red[0];
        for (int w = 1; w < SEL_NT / 64; w++) if (sm->red[w] < b) b = sm->red[w];
        sm->prefix = b;
    }
    __syncthreads();
}

// np.median of { key_f64(f(i)) }: middle order statistic, or (lo + hi) / 2 for even n.
// All threads call and all get the value.
template <class F>
__device__ double block_median(F f, i64 n, SelectSmem *sm)
{
    i64 k_lo = (n - 1) / 2;
    block_select(f, n, k_lo, sm);
    u64 key_lo = sm->prefix;
    i64 n_le = sm->n_less + sm->n_eq;
    __syncthreads();
    double lo = key_f64(key_lo);
    if (n & 1) return lo;
    double hi;
    if (n_le > k_lo + 1) {
        hi = lo;
    } else {
        block_min_greater(f, n, key_lo, sm);
        hi = key_f64(sm->prefix);
        __syncthreads();
    }
    return (lo + hi) / 2.0;
}

// workgroup sum of an i64 (all threads call; all get the total)
__device__ inline i64 block_sum_i64(i64 v, SelectSmem *sm)
{
    const int tid = threadIdx.x;
    for (int m = 32; m >= 1; m >>= 1) v += shfl_i64(v, (tid & 63) ^ m);
    __syncthreads();
    if ((tid & 63) == 0) sm->red[tid >> 6] = (u64)v;
    __syncthreads();
    i64 t = 0;
    for (int w = 0; w < SEL_NT / 64; w++) t += (i64)sm->red[w];
    __syncthreads();
    return t;
}
