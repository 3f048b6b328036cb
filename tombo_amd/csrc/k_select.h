// k_select.h -- exact order statistics inside one workgroup (radix select on order-preserving
// 64-bit keys).  Used for np.median (normalisation, Theil-Sen) and for the event-detection cap.
// Pure comparisons / counting: bit-exact by construction.
#pragma once
#include "tba_common.h"

// Memory-level parallelism.  A `load -> use -> store` loop has ONE request per thread in flight:
// at 32 waves per CU that is 16 KB per CU, i.e. 2.5-3 TB/s at the latency of a busy memory
// system however simple the loop body (what bounded k_normalize, k_rescale_absz and
// k_event_means in round 1).  The streaming loops below therefore issue U independent loads
// first and only then consume them.
//
// block_stream: workgroup-strided pass over [0, n); body(i, load(i)) for every i (no collectives
// inside body: the tail runs with part of the threads).
template <int U, class Load, class Body>
__device__ __forceinline__ void block_stream(i64 n, Load load, Body body)
{
    const i64 tid = threadIdx.x, nt = blockDim.x;
    i64 base = 0;
    for (; base + (i64)U * nt <= n; base += (i64)U * nt) {
        decltype(load((i64)0)) v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = load(base + (i64)u * nt + tid);
#pragma unroll
        for (int u = 0; u < U; u++) body(base + (i64)u * nt + tid, v[u]);
    }
    for (i64 i = base + tid; i < n; i += nt) body(i, load(i));
}

// block_stream2: the same pass with two consecutive elements per access (sig_pair: 16 bytes per
// lane for float64); body(i, value) for every i, the odd last element by thread 0.
template <int U, class Sig, class Body>
__device__ __forceinline__ void block_stream2(i64 n, Sig x, Body body)
{
    const i64 tid = threadIdx.x, nt = blockDim.x, np = n >> 1;
    i64 base = 0;
    for (; base + (i64)U * nt <= np; base += (i64)U * nt) {
        double a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) sig_pair(x, 2 * (base + (i64)u * nt + tid), a[u], b[u]);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const i64 i = 2 * (base + (i64)u * nt + tid);
            body(i, a[u]); body(i + 1, b[u]);
        }
    }
    for (i64 j = base + tid; j < np; j += nt) {
        double a, b;
        sig_pair(x, 2 * j, a, b);
        body(2 * j, a); body(2 * j + 1, b);
    }
    if ((n & 1) && tid == 0) body(n - 1, x[n - 1]);
}
// ... and with the results written back pairwise: out[i] = f(x[i]) (16-byte stores)
template <int U, class Sig, class F>
__device__ __forceinline__ void block_map2(i64 n, Sig x, double *out, F f)
{
    const i64 tid = threadIdx.x, nt = blockDim.x, np = n >> 1;
    i64 base = 0;
    for (; base + (i64)U * nt <= np; base += (i64)U * nt) {
        double a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) sig_pair(x, 2 * (base + (i64)u * nt + tid), a[u], b[u]);
#pragma unroll
        for (int u = 0; u < U; u++) st2(out + 2 * (base + (i64)u * nt + tid), f(a[u]), f(b[u]));
    }
    for (i64 j = base + tid; j < np; j += nt) {
        double a, b;
        sig_pair(x, 2 * j, a, b);
        st2(out + 2 * j, f(a), f(b));
    }
    if ((n & 1) && tid == 0) out[n - 1] = f(x[n - 1]);
}

// wave_stage: a wavefront copies x[lo .. lo + span), span <= 128 * NP, into its LDS slice as
// f(value) -- and, with out != nullptr, to out[lo ..] as well -- two consecutive elements per lane
// and access (16-byte loads and stores); all loads of a lane go out before the first value is used
// (uniform branches: span is made an SGPR).  The pair that straddles the end of an odd span reads
// one element past it: every signal buffer of the engine carries a few bytes of slack for that.
template <int NP, class Sig>
__device__ __forceinline__ void wave_stage_load(Sig x, i64 lo, int spn, double (&a)[NP], double (&b)[NP])
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int u = 0; u < NP; u++) {
        if (128 * u < spn) {
            int k = 2 * lane + 128 * u;
            k = k < spn ? k : (spn - 1) & ~1;       // (lanes past the end re-read the last pair)
            sig_pair(x, lo + k, a[u], b[u]);
        }
    }
}
template <int NP, class F>
__device__ __forceinline__ void wave_stage_store(const double (&a)[NP], const double (&b)[NP], i64 lo, int spn,
                                                 double *lds, double *out, F f)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int u = 0; u < NP; u++) {
        if (128 * u < spn) {
            const int k = 2 * lane + 128 * u;
            if (k + 1 < spn) {
                const double va = f(a[u]), vb = f(b[u]);
                lds[k] = va; lds[k + 1] = vb;
                if (out) st2(out + lo + k, va, vb);
            } else if (k < spn) {
                const double va = f(a[u]);
                lds[k] = va;
                if (out) out[lo + k] = va;
            }
        }
    }
}
template <int NP, class Sig, class F>
__device__ __forceinline__ void wave_stage(Sig x, i64 lo, i64 span, double *lds, double *out, F f)
{
    const int spn = __builtin_amdgcn_readfirstlane((int)span);
    double a[NP], b[NP];
    wave_stage_load<NP>(x, lo, spn, a, b);
    wave_stage_store<NP>(a, b, lo, spn, lds, out, f);
}

#define SEL_NT 512 // threads per workgroup for every kernel that uses these helpers

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
        sm->hist[tid & 255] = 0; // (SEL_NT >= 256 threads clear the 256 bins)
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

// ---------------------------------------------------------------------------------------------
// Bucket select: exact k-th order statistic in ~2 passes for data whose range is roughly known.
// Pass 1 histograms a monotone (non-decreasing) value->bucket map over [lo, hi]; the bucket that
// holds rank k is gathered into LDS (pass 2) and resolved by rank counting.  Any range gives the
// right answer (values outside fall into the edge buckets); a bad range only costs refinement
// levels (re-bucket the members of the chosen bucket over their exact min/max), and after
// BS_LEVELS levels the radix select above takes over.  Comparisons and counting only: exact.
#define BS_NB 4096
#define BS_CAP 2048
#define BS_LEVELS 4

struct BucketSmem {
    union { // the 32 KB histogram + candidate area doubles as a raw tile buffer (k_peaks)
        struct { u32 hist[BS_NB]; double cand[BS_CAP]; };
        double raw8[BS_NB / 2 + BS_CAP];
    };
    double lo[BS_LEVELS], scale[BS_LEVELS];
    i32 bk[BS_LEVELS];
    i32 nlev, found;
    u32 n_cand;
    i64 k, cnt;
    double result, next; // next: smallest candidate above result (or +inf)
    i64 n_le;            // number of elements <= result (valid when has_le)
    double redd[2 * (SEL_NT / 64)];
    SelectSmem rad;      // fallback
#ifdef TBA_PHASE_DEBUG
    i64 stamp[4];        // cycle counter after: histogram pass, bucket search, gather pass, ranking
#endif
};
#ifdef TBA_PHASE_DEBUG
#define BS_STAMP(i_) do { if (threadIdx.x == 0) sm->stamp[i_] = (i64)__builtin_readcyclecounter(); } while (0)
#else
#define BS_STAMP(i_) do { } while (0)
#endif

__device__ __forceinline__ int bs_bucket(double v, double lo, double scale)
{
    double t = (v - lo) * scale;
    int b = t >= (double)(BS_NB - 1) ? BS_NB - 1 : (t > 0.0 ? (int)t : 0);
    return b;
}
__device__ __forceinline__ bool bs_member(const BucketSmem *sm, int nlev, double v)
{
    for (int l = 0; l < nlev; l++)
        if (bs_bucket(v, sm->lo[l], sm->scale[l]) != sm->bk[l]) return false;
    return true;
}

// wave 0 (all 64 lanes): the histogram bucket that holds rank sm->k (BS_NB/64 bins per lane);
// records it as level nlev and leaves the rank inside it / its member count in sm->k / sm->cnt.
// (k_pick's copy of the search inside block_kth_fe: sharing one function moves k_normalize's
// register allocation, which sits exactly on its 128-VGPR step, into spills)
__device__ __forceinline__ void bs_locate(BucketSmem *sm, int nlev, double lo, double scale)
{
    const int tid = threadIdx.x;
    const int per = BS_NB / 64;
    i64 c = 0;
    for (int q = 0; q < per; q++) c += sm->hist[tid * per + q];
    i64 inc = c;
    for (int d = 1; d < 64; d <<= 1) {
        i64 t = shfl_i64(inc, tid - d < 0 ? 0 : tid - d);
        if (tid >= d) inc += t;
    }
    i64 exc = inc - c, kk = sm->k;
    if (exc <= kk && kk < inc) {
        i64 acc = exc;
        for (int q = 0; q < per; q++) {
            u32 h = sm->hist[tid * per + q];
            if (kk < acc + h) {
                sm->bk[nlev] = tid * per + q;
                sm->lo[nlev] = lo; sm->scale[nlev] = scale;
                sm->k = kk - acc; sm->cnt = h;
                break;
            }
            acc += h;
        }
        sm->nlev = nlev + 1;
    }
}

// Element enumeration for block_kth: a functor fe(visit) must call visit(value, valid) for every
// element exactly once, with a trip count that is uniform across the workgroup's lanes (visit
// contains wave ballots); invalid visits pad the last iterations.  flat_elems adapts an indexed
// accessor val(i), 0 <= i < n.
template <class F>
struct FlatElems {
    F val; i64 n;
    template <class V> __device__ void operator()(V visit) const
    {
        constexpr int U = 4; // element loads in flight per thread (see block_stream)
        for (i64 base = 0; base < n; base += (i64)U * SEL_NT) {
            double v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const i64 i = base + (i64)u * SEL_NT + threadIdx.x;
                v[u] = val(i < n ? i : n - 1);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool ok = base + (i64)u * SEL_NT + threadIdx.x < n;
                visit(ok ? v[u] : 0.0, ok);
            }
        }
    }
};
template <class F> __device__ FlatElems<F> flat_elems(F val, i64 n) { return FlatElems<F>{val, n}; }

// k-th smallest (0-based) of the n elements enumerated by fe; all threads call, all return the
// value.  On return sm->next holds the (k+1)-th smallest if it could be determined cheaply
// (sm->found & 2), i.e. when it lies in the same final bucket or equals the k-th.
template <class FE>
__device__ double block_kth_fe(FE fe, i64 n, i64 k, double lo, double hi, BucketSmem *sm)
{
    const int tid = threadIdx.x;
    if (tid == 0) { sm->nlev = 0; sm->k = k; sm->found = 0; sm->cnt = n; }
    __syncthreads();
    // few elements: gather them all and count ranks directly (quadratic, so only when that is
    // cheaper than a histogram pass)
    const bool small = n <= 192;
    for (int level = 0; level < BS_LEVELS; level++) {
        double scale = (double)BS_NB / (hi - lo);
        if (!small && (!(hi > lo) || !(scale < 1e300))) break; // degenerate range -> radix
        if (!small) {
        for (int b = tid; b < BS_NB; b += SEL_NT) sm->hist[b] = 0;
        __syncthreads();
        const int nlev = sm->nlev;
        fe([&](double v, bool ok) {
            // 4096 fine buckets: lanes rarely share one, plain LDS atomics beat leader aggregation
            if (ok && bs_member(sm, nlev, v)) atomicAdd(&sm->hist[bs_bucket(v, lo, scale)], 1u);
        });
        __syncthreads();
        BS_STAMP(0);
        if (tid < 64) { // wave 0: locate the bucket of rank k (BS_NB/64 bins per lane)
            const int per = BS_NB / 64;
            i64 c = 0;
            for (int q = 0; q < per; q++) c += sm->hist[tid * per + q];
            i64 inc = c;
            for (int d = 1; d < 64; d <<= 1) {
                i64 t = shfl_i64(inc, tid - d < 0 ? 0 : tid - d);
                if (tid >= d) inc += t;
            }
            i64 exc = inc - c, kk = sm->k;
            if (exc <= kk && kk < inc) {
                i64 acc = exc;
                for (int q = 0; q < per; q++) {
                    u32 h = sm->hist[tid * per + q];
                    if (kk < acc + h) {
                        sm->bk[nlev] = tid * per + q;
                        sm->lo[nlev] = lo; sm->scale[nlev] = scale;
                        sm->k = kk - acc; sm->cnt = h;
                        break;
                    }
                    acc += h;
                }
                sm->nlev = nlev + 1;
            }
        }
        } // !small
        __syncthreads();
        BS_STAMP(1);
        const i64 cnt = sm->cnt;
        const int nl = sm->nlev;
        if (cnt <= BS_CAP) {
            // gather the bucket and pick rank k inside it by counting
            if (tid == 0) sm->n_cand = 0;
            __syncthreads();
            fe([&](double v, bool ok) {
                if (ok && bs_member(sm, nl, v)) { u32 p = atomicAdd(&sm->n_cand, 1u); if (p < BS_CAP) sm->cand[p] = v; }
            });
            __syncthreads();
            BS_STAMP(2);
            const int m = (int)cnt;
            const i64 kk = sm->k;
            for (int a = tid; a < m; a += SEL_NT) {
                const double va = sm->cand[a];
                int less = 0, eq_before = 0, le = 0;
                double nxt = INFINITY;
                for (int b2 = 0; b2 < m; b2++) {
                    const double vb = sm->cand[b2];
                    less += vb < va;
                    le += vb <= va;
                    eq_before += (vb == va) && (b2 < a);
                    nxt = (vb > va && vb < nxt) ? vb : nxt;
                }
                if (less + eq_before == kk) { // exactly one candidate has this rank
                    sm->result = va;
                    // the next order statistic: same value if another copy follows, else the
                    // smallest larger candidate of this bucket (if any)
                    if (le - 1 > kk) { sm->next = va; sm->found = 3; }
                    else if (nxt < INFINITY) { sm->next = nxt; sm->found = 3; }
                    else sm->found = 1;
                }
            }
            __syncthreads();
            BS_STAMP(3);
            return sm->result;
        }
        // too many members: re-bucket them over their exact min / max
        double mn = INFINITY, mx = -INFINITY;
        fe([&](double v, bool ok) {
            if (ok && bs_member(sm, nl, v)) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        });
        for (int mm = 32; mm >= 1; mm >>= 1) {
            double a = shfl_xor_f64(mn, mm), b2 = shfl_xor_f64(mx, mm);
            mn = a < mn ? a : mn; mx = b2 > mx ? b2 : mx;
        }
        if ((tid & 63) == 0) { sm->redd[2 * (tid >> 6)] = mn; sm->redd[2 * (tid >> 6) + 1] = mx; }
        __syncthreads();
        mn = sm->redd[0]; mx = sm->redd[1];
        for (int w = 1; w < SEL_NT / 64; w++) {
            mn = sm->redd[2 * w] < mn ? sm->redd[2 * w] : mn;
            mx = sm->redd[2 * w + 1] > mx ? sm->redd[2 * w + 1] : mx;
        }
        __syncthreads();
        if (mn == mx) { // every member equal: that is the answer, and the next one too if cnt > k+1
            if (tid == 0) { sm->result = mn; sm->next = mn; sm->found = sm->cnt - 1 > sm->k ? 3 : 1; }
            __syncthreads();
            return mn;
        }
        lo = mn; hi = mx;
    }
    // pathological input: exact radix select on the whole set (8 passes over 8-bit digits)
    {
        SelectSmem *rs_ = &sm->rad;
        if (tid == 0) { rs_->prefix = 0; rs_->k = k; rs_->n_less = 0; rs_->n_eq = 0; }
        __syncthreads();
        for (int pass = 7; pass >= 0; pass--) {
            rs_->hist[tid & 255] = 0;
            __syncthreads();
            const u64 prefix = rs_->prefix;
            const int sh = 8 * pass;
            const u64 himask = pass == 7 ? 0ull : (~0ull << (sh + 8));
            fe([&](double v, bool ok) {
                const u64 key = f64_key(v);
                hist_add(rs_->hist, ok && (key & himask) == prefix, (u32)((key >> sh) & 255));
            });
            __syncthreads();
            if (tid == 0) {
                i64 kk = rs_->k, acc = 0;
                for (int b2 = 0; b2 < 256; b2++) {
                    const u32 h = rs_->hist[b2];
                    if (kk < acc + h) { rs_->prefix = prefix | ((u64)b2 << sh); rs_->k = kk - acc; break; }
                    acc += h;
                }
            }
            __syncthreads();
        }
        const double res = key_f64(rs_->prefix);
        if (tid == 0) { sm->result = res; sm->found = 1; }
        __syncthreads();
        return res;
    }
}

template <class F>
__device__ double block_kth(F val, i64 n, i64 k, double lo, double hi, BucketSmem *sm)
{
    return block_kth_fe(flat_elems(val, n), n, k, lo, hi, sm);
}

// np.median through block_kth: (lower middle + upper middle) / 2.  The upper middle comes for
// free when it shares the final bucket with the lower one, else one min-greater pass.
// Returns the median; *lo_mid / *hi_mid (if not NULL) get the two middle order statistics.
template <class FE>
__device__ double block_median_fe(FE fe, i64 n, double lo, double hi, BucketSmem *sm,
                                  double *lo_mid = nullptr, double *hi_mid = nullptr)
{
    const i64 k_lo = (n - 1) / 2;
    double a = block_kth_fe(fe, n, k_lo, lo, hi, sm);
    double b = a;
    if (!(n & 1)) {
        const int found = sm->found;
        const double nxt = sm->next;
        __syncthreads();
        if (found & 2) {
            b = nxt;
        } else {
            // count elements <= a; if that exceeds k_lo + 1 the next one equals a, otherwise it
            // is the smallest element above a
            i64 le = 0;
            double mn = INFINITY;
            fe([&](double v, bool ok) {
                le += ok && v <= a;
                mn = (ok && v > a && v < mn) ? v : mn;
            });
            le = block_sum_i64(le, &sm->rad);
            for (int mm = 32; mm >= 1; mm >>= 1) { double t = shfl_xor_f64(mn, mm); mn = t < mn ? t : mn; }
            if ((threadIdx.x & 63) == 0) sm->redd[threadIdx.x >> 6] = mn;
            __syncthreads();
            mn = sm->redd[0];
            for (int w = 1; w < SEL_NT / 64; w++) mn = sm->redd[w] < mn ? sm->redd[w] : mn;
            __syncthreads();
            b = le > k_lo + 1 ? a : mn;
        }
    }
    if (lo_mid) *lo_mid = a;
    if (hi_mid) *hi_mid = b;
    return (n & 1) ? a : (a + b) / 2.0;
}
template <class F>
__device__ double block_median_fast(F val, i64 n, double lo, double hi, BucketSmem *sm,
                                    double *lo_mid = nullptr, double *hi_mid = nullptr)
{
    return block_median_fe(flat_elems(val, n), n, lo, hi, sm, lo_mid, hi_mid);
}

// ---------------------------------------------------------------------------------------------
// np.median through a sampled window: ONE pass over the data instead of the bucket select's two.
// Every thread holds WS_PER strided samples in registers; a histogram of the sample gives a value
// window [t1, t2) around the sample quantiles 0.5 -/+ dq that holds the two middle ranks of the
// full set with near certainty; the pass counts the elements below t1 and appends the ones inside
// the window to `list` (global scratch, >= cap doubles); the middle order statistics are then
// selected exactly among the listed values.  Count and list are exact, so the result is the
// reference's np.median bit for bit; when the window misses a middle rank, or the list
// overflows (massive ties), the caller falls back to block_median_fast.  All threads call;
// returns true on success with *lo_mid / *hi_mid = the two middle order statistics.
#ifndef WS_PER
#define WS_PER 4  // samples per thread (SEL_NT * WS_PER in all: one per ~3 cache lines of a 10 kb
                  // read; 16 per thread touch every line -- most of a pass -- for a window half
                  // as wide: 8.5 vs 6.5 ms on the 10k x 10 kb batch)
#endif
// The elements are of(x[i]), 0 <= i < n: x a signal (sig_pair: two consecutive samples per access),
// of the value looked at (x itself, |x - shift|, ...).
template <int WU = 4, class X, class OF> // WU: PAIR loads in flight per thread during the pass
__device__ __forceinline__ bool block_median_window(X x, OF of, i64 n, const double (&samp)[WS_PER], double *list,
                                    i64 cap, BucketSmem *sm, double *lo_mid, double *hi_mid)
{
    const int tid = threadIdx.x;
    __shared__ double s_t[2];
    __shared__ int s_ok;
    __shared__ u32 s_nl;
    // range of the sample
    double mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < WS_PER; q++) { mn = samp[q] < mn ? samp[q] : mn; mx = samp[q] > mx ? samp[q] : mx; }
    for (int mm = 32; mm >= 1; mm >>= 1) {
        double a = shfl_xor_f64(mn, mm), b2 = shfl_xor_f64(mx, mm);
        mn = a < mn ? a : mn; mx = b2 > mx ? b2 : mx;
    }
    __syncthreads();
    if ((tid & 63) == 0) { sm->redd[2 * (tid >> 6)] = mn; sm->redd[2 * (tid >> 6) + 1] = mx; }
    for (int b = tid; b < BS_NB; b += SEL_NT) sm->hist[b] = 0;
    if (tid == 0) { s_ok = 0; s_nl = 0; }
    __syncthreads();
    mn = sm->redd[0]; mx = sm->redd[1];
    for (int w = 1; w < SEL_NT / 64; w++) {
        mn = sm->redd[2 * w] < mn ? sm->redd[2 * w] : mn;
        mx = sm->redd[2 * w + 1] > mx ? sm->redd[2 * w + 1] : mx;
    }
    if (!(mx > mn)) return false;                    // constant sample: no window to speak of
    const double hsc = (double)BS_NB / (mx - mn);
    if (!(hsc < 1e300)) return false;
#pragma unroll
    for (int q = 0; q < WS_PER; q++) atomicAdd(&sm->hist[bs_bucket(samp[q], mn, hsc)], 1u);
    __syncthreads();
    if (tid < 64) { // wave 0: the buckets of the sample ranks (0.5 -/+ dq) m
        const double m = (double)(SEL_NT * WS_PER);
        const double dq = 4.0 * sqrt(0.25 / m) + 0.002;
        const i64 r1 = (i64)((0.5 - dq) * m), r2 = (i64)((0.5 + dq) * m);
        const int per = BS_NB / 64;
        i64 c = 0;
        for (int q = 0; q < per; q++) c += sm->hist[tid * per + q];
        i64 inc = c;
        for (int d = 1; d < 64; d <<= 1) {
            i64 t = shfl_i64(inc, tid - d < 0 ? 0 : tid - d);
            if (tid >= d) inc += t;
        }
        i64 acc = inc - c;
        for (int q = 0; q < per; q++) {
            const i64 nx = acc + sm->hist[tid * per + q];
            const int b = tid * per + q;
            // first bucket whose cumulative count exceeds the rank
            if (acc <= r1 && r1 < nx) s_t[0] = b == 0 ? -INFINITY : mn + (double)b / hsc;
            if (acc <= r2 && r2 < nx) { s_t[1] = b == BS_NB - 1 ? INFINITY : mn + (double)(b + 1) / hsc; s_ok = 1; }
            acc = nx;
        }
    }
    __syncthreads();
    if (!s_ok) return false;
    const double t1 = s_t[0], t2 = s_t[1];
    // the pass: count below the window, list the window (wave-aggregated append)
    i64 c_lo = 0;
    const int lane = tid & 63;
    // (WU pair loads in flight per thread, one list-counter bump per 2 WU * 64 elements; the odd
    // last element rides in the last pair slot of the pass)
    const i64 np = (n + 1) >> 1;
    for (i64 base = 0; base < np; base += (i64)WU * SEL_NT) {
        double v[2 * WU];
        bool okv[2 * WU];
#pragma unroll
        for (int u = 0; u < WU; u++) {
            const i64 j = base + (i64)u * SEL_NT + tid;
            const i64 i = 2 * j;
            okv[2 * u] = i < n; okv[2 * u + 1] = i + 1 < n;
            if (i + 1 < n) sig_pair(x, i, v[2 * u], v[2 * u + 1]);
            else { v[2 * u] = x[i < n ? i : n - 1]; v[2 * u + 1] = v[2 * u]; }
        }
        u64 m[2 * WU];
        u32 tot = 0;
#pragma unroll
        for (int u = 0; u < 2 * WU; u++) {
            v[u] = of(v[u]);
            c_lo += okv[u] && v[u] < t1;
            m[u] = __ballot(okv[u] && v[u] >= t1 && v[u] < t2);
            tot += (u32)__popcll(m[u]);
        }
        if (tot) {
            u32 b0 = 0;
            if (lane == 0) b0 = atomicAdd(&s_nl, tot);
            b0 = __shfl(b0, 0, 64);
#pragma unroll
            for (int u = 0; u < 2 * WU; u++) {
                const u32 pos = b0 + (u32)__popcll(m[u] & ((1ull << lane) - 1ull));
                if (((m[u] >> lane) & 1ull) && pos < cap) list[pos] = v[u];
                b0 += (u32)__popcll(m[u]);
            }
        }
    }
    c_lo = block_sum_i64(c_lo, &sm->rad);
    __threadfence_block();
    __syncthreads();
    const i64 n_c = s_nl;
    const i64 k_lo = (n - 1) / 2 - c_lo, k_hi = n / 2 - c_lo;
    if (n_c > cap || k_lo < 0 || k_hi >= n_c) return false;
    // bucket range for the list: the window, clamped to something finite
    const double span = mx - mn;
    const double lo = t1 > mn - span ? t1 : mn - span, hi = t2 < mx + span ? t2 : mx + span;
    const double a = block_kth([&](i64 k) { return list[k]; }, n_c, k_lo, lo, hi, sm);
    const int found = sm->found;
    const double nxt = sm->next;
    __syncthreads();
    double b = a;
    if (k_hi != k_lo) {
        if (found & 2) b = nxt;
        else { b = block_kth([&](i64 k) { return list[k]; }, n_c, k_hi, lo, hi, sm); __syncthreads(); }
    }
    *lo_mid = a; *hi_mid = b;
    return true;
}

// Exact medians of int16 samples from ONE counting pass: hist[v - base] over a window of
// BS_NB consecutive integer values placed around the sample (DAC values of a read span a few
// hundred to a thousand levels).  Returns false when a middle rank falls outside the window
// (the caller falls back to the generic select).  On success: *xlo / *xhi = middle order
// statistics of x, and -- given shift2 = 2 * median, an integer -- *dlo / *dhi = middle order
// statistics of |x - median| (from the same histogram: a deviation e/2 collects the bins at
// (shift2 -/+ e) / 2).  Counting only: exact.
template <class XS>
__device__ bool block_int_medians(XS x, i64 n, int vmin_s, int vmax_s, BucketSmem *sm,
                                  double *xlo, double *xhi, double *dlo, double *dhi)
{
    const int tid = threadIdx.x;
    __shared__ i64 s_k[4];
    __shared__ int s_okk;
    if (vmax_s - vmin_s >= BS_NB - 128) return false;
    const int base = vmin_s - (BS_NB - (vmax_s - vmin_s + 1)) / 2;  // sample range centred
    for (int b = tid; b < BS_NB; b += SEL_NT) sm->hist[b] = 0;
    if (tid == 0) s_okk = 0;
    __syncthreads();
    i64 below = 0, above = 0;
    block_stream<8>(n, [&](i64 i) { return (int)x[i]; }, [&](i64, int xv) {
        const int b = xv - base;
        if (b < 0) below++;
        else if (b >= BS_NB) above++;
        else atomicAdd(&sm->hist[b], 1u);
    });
    below = block_sum_i64(below, &sm->rad);
    above = block_sum_i64(above, &sm->rad);
    __syncthreads();
    const i64 k_lo = (n - 1) / 2, k_hi = n / 2;
    if (k_lo < below || k_hi >= n - above) return false;
    // inclusive prefix over the bins, kept in cand[] as doubles (counts < 2^53: exact); 8 bins
    // per thread + wave / block scan
    {
        const int per = BS_NB / SEL_NT; // 8
        i64 loc[BS_NB / SEL_NT];
        i64 c = 0;
#pragma unroll
        for (int q = 0; q < per; q++) { c += sm->hist[tid * per + q]; loc[q] = c; }
        i64 inc = c;
        for (int d = 1; d < 64; d <<= 1) {
            i64 t = shfl_i64(inc, (tid & 63) - d < 0 ? 0 : (tid & 63) - d);
            if ((tid & 63) >= d) inc += t;
        }
        __syncthreads();
        if ((tid & 63) == 63) sm->rad.red[tid >> 6] = (u64)inc;
        __syncthreads();
        i64 off = below;
        for (int w = 0; w < (tid >> 6); w++) off += (i64)sm->rad.red[w];
        const i64 exc = off + inc - c;
#pragma unroll
        for (int q = 0; q < per; q++) {
            const i64 lo_c = exc + (q ? loc[q - 1] : 0), hi_c = exc + loc[q];
            if (lo_c <= k_lo && k_lo < hi_c) s_k[0] = base + tid * per + q;
            if (lo_c <= k_hi && k_hi < hi_c) s_k[1] = base + tid * per + q;
        }
    }
    __syncthreads();
    const i64 vlo = s_k[0], vhi = s_k[1];
    *xlo = (double)vlo; *xhi = (double)vhi;
    // deviations from the median (vlo + vhi) / 2 (n even) or vlo (n odd), in half units:
    // e(v) = |2 v - shift2|, same parity as shift2; t-th candidate e_t = par + 2 t
    const i64 shift2 = (n & 1) ? 2 * vlo : vlo + vhi;
    const int par = (int)(shift2 & 1);
    // count of candidate t: bins (shift2 - e) / 2 and (shift2 + e) / 2; out-of-window values have
    // the largest deviations and only matter if a middle rank reaches them (-> fail)
    {
        const int per = BS_NB / SEL_NT;
        i64 loc[BS_NB / SEL_NT];
        i64 c = 0;
#pragma unroll
        for (int q = 0; q < per; q++) {
            const i64 e = par + 2 * (i64)(tid * per + q);
            const i64 a = (shift2 - e) / 2 - base, b2 = (shift2 + e) / 2 - base;
            i64 cnt = 0;
            if (a >= 0 && a < BS_NB) cnt += sm->hist[a];
            if (e != 0 && b2 >= 0 && b2 < BS_NB) cnt += sm->hist[b2];
            c += cnt; loc[q] = c;
        }
        i64 inc = c;
        for (int d = 1; d < 64; d <<= 1) {
            i64 t = shfl_i64(inc, (tid & 63) - d < 0 ? 0 : (tid & 63) - d);
            if ((tid & 63) >= d) inc += t;
        }
        __syncthreads();
        if ((tid & 63) == 63) sm->rad.red[tid >> 6] = (u64)inc;
        __syncthreads();
        i64 off = 0;
        for (int w = 0; w < (tid >> 6); w++) off += (i64)sm->rad.red[w];
        const i64 exc = off + inc - c;
        i64 tot = 0;
        for (int w = 0; w < SEL_NT / 64; w++) tot += (i64)sm->rad.red[w];
        if (tid == 0) s_okk = (tot == n - below - above) && k_hi < tot; // every in-window bin reached
#pragma unroll
        for (int q = 0; q < per; q++) {
            const i64 lo_c = exc + (q ? loc[q - 1] : 0), hi_c = exc + loc[q];
            const i64 e = par + 2 * (i64)(tid * per + q);
            if (lo_c <= k_lo && k_lo < hi_c) s_k[2] = e;
            if (lo_c <= k_hi && k_hi < hi_c) s_k[3] = e;
        }
    }
    __syncthreads();
    if (!s_okk) return false;
    // values outside the window deviate more than anything inside only up to the distance of
    // the nearer window edge: a middle rank beyond that cannot be ranked from the histogram
    if (below > 0 && (shift2 - s_k[3]) / 2 < (i64)base) return false;
    if (above > 0 && (shift2 + s_k[3]) / 2 > (i64)base + BS_NB - 1) return false;
    *dlo = (double)s_k[2] / 2.0; *dhi = (double)s_k[3] / 2.0;
    return true;
}


// sum of p[j0 .. j1) in index order (c_new_means adds sample by sample), the LDS loads issued eight
// at a time: the adds are the only dependent chain (a plain loop waits one LDS round trip per
// sample, which is what bounded the RNA segment kernels: 15-sample events, 43-sample bases)
__device__ __forceinline__ double seq_sum_lds(const double *p, int j0, int j1, double s = 0)
{
    int j = j0;
    for (; j + 8 <= j1; j += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = p[j + u];
#pragma unroll
        for (int u = 0; u < 8; u++) s += t[u];
    }
    for (; j < j1; j++) s += p[j];
    return s;
}

// c_new_means-style segment sums, wave-cooperative: a wavefront takes a group of consecutive
// segments, pulls the samples they span (one contiguous range) into its LDS slice with coalesced
// loads -- as f(sample), written to out[] as well when out != nullptr -- and every lane then sums its
// own segment out of LDS: sequentially, in sample order, like the reference.  (A thread-per-segment
// loop over global memory touches 64 different cache lines per load instead.)
// The loop over a wave's groups is software-pipelined: while the lanes add up group g out of LDS
// the samples of group g + 1 are on their way into registers and the boundaries of group g + 2
// (and whatever emit needs per segment: pre(i)) behind them, so a step waits one memory round trip
// where it used to wait three in a row (boundaries -> samples -> emit's operands; RNA, 10 000
// reads: k_rescale_absz 8.2 ms, k_event_means 4.5 ms, both bound by exactly that chain).
// A group whose span exceeds the slice (long dwell, RNA stalls) is staged SEGW_CAP samples at a
// time, every lane carrying its running sum from one piece to the next: the same additions in the
// same order.
// seg has n_segs + 1 ascending boundaries; emit(i, sum, length, pre(i)) per segment.
template <int SEGW_CAP, class Sig, class F, class Pre, class Emit>
__device__ __forceinline__ void wave_segment_sums(Sig x,
    const i64 *__restrict__ seg, i64 n_segs, i64 first_group, i64 group_stride, double *lds,
    double *out, F f, Pre pre, Emit emit, double mean_len)
{
    constexpr int NP = (SEGW_CAP + 127) / 128;
    const int lane = threadIdx.x & 63;
    // segments per wave step: 64, or fewer (a power of two) when the segments are long, so that
    // a step's samples still fit the slice (RNA: 15-sample events, 43-sample bases); the lanes
    // beyond the group only help loading
    // (mean_len: the caller's estimate of the mean segment length -- any value is correct, it only
    // picks the group size; reading it off seg[] would be two more dependent loads per workgroup)
    int gs = 64;
    while (gs > 4 && (double)gs * mean_len * 1.3 > (double)SEGW_CAP) gs >>= 1;
    i64 g = first_group;
    if (g * gs >= n_segs) return;
    // the boundaries of a group as loaded (one coalesced load + the group's last one, one address for
    // the whole wavefront), resolved -- neighbour by shuffle -- only when the group comes up
    struct Raw { i64 a, hi; };
    auto load_raw = [&](i64 gg) {
        const i64 i = gg * gs + lane;
        const bool ok = lane < gs && i < n_segs;
        const i64 i_end = gg * gs + gs < n_segs ? gg * gs + gs : n_segs;
        return Raw{seg[ok ? i : i_end], seg[i_end]};
    };
    // a group resolved: lo / span are the same for the whole wavefront (scalar registers), a lane's own
    // segment is [ja, jb) relative to lo
    struct Grp { i64 lo; int span, ja, jb; bool ok; };
    auto rfl64 = [](i64 v) {
        const u32 l = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffff)), h = __builtin_amdgcn_readfirstlane((int)(v >> 32));
        return (i64)(((u64)h << 32) | l);
    };
    auto resolve = [&](i64 gg, Raw w) {
        const i64 i = gg * gs + lane;
        const bool ok = lane < gs && i < n_segs;
        const i64 i_end = gg * gs + gs < n_segs ? gg * gs + gs : n_segs;
        const i64 lo = rfl64(w.a), hi = rfl64(w.hi);       // (lane 0 is always inside the group)
        const int ja = (int)(w.a - lo), hj = (int)(hi - lo);
        const int jn = __shfl(ja, lane + 1 < 64 ? lane + 1 : 63, 64);
        const int jb = ok ? (i + 1 < i_end && lane + 1 < 64 ? jn : hj) : ja;
        return Grp{lo, hj, ja, jb, ok};
    };
    double ra[NP], rb[NP];
    Grp cur = resolve(g, load_raw(g));
    auto pre_cur = pre(g * gs + lane < n_segs ? g * gs + lane : n_segs - 1);
    bool staged = cur.span <= SEGW_CAP;
    if (staged) wave_stage_load<NP>(x, cur.lo, cur.span, ra, rb);
    i64 gn = g + group_stride;
    bool has_next = gn * gs < n_segs;
    Raw nraw = has_next ? load_raw(gn) : Raw{0, 0};
    auto pre_nxt = pre(has_next && gn * gs + lane < n_segs ? gn * gs + lane : n_segs - 1);
    for (;;) {
        double s = 0;
        if (staged) {
            __builtin_amdgcn_wave_barrier(); // the previous group's lanes are done with the slice
            wave_stage_store<NP>(ra, rb, cur.lo, cur.span, lds, out, f);
            __builtin_amdgcn_wave_barrier();
        } else {
            for (int c0 = 0; c0 < cur.span; c0 += SEGW_CAP) {
                const int left = cur.span - c0, piece = left < SEGW_CAP ? left : SEGW_CAP;
                __builtin_amdgcn_wave_barrier();
                wave_stage_load<NP>(x, cur.lo + c0, piece, ra, rb);   // (ra / rb are free: the next group is not on its way yet)
                wave_stage_store<NP>(ra, rb, cur.lo + c0, piece, lds, out, f);
                __builtin_amdgcn_wave_barrier();
                const int j0 = (cur.ja > c0 ? cur.ja : c0) - c0, j1 = (cur.jb < c0 + piece ? cur.jb : c0 + piece) - c0;
                if (j1 > j0) s = seq_sum_lds(lds, j0, j1, s);
            }
        }
        // the next group's samples, the boundaries of the one after
        const Grp now = cur;
        const bool was_staged = staged;
        const auto pre_now = pre_cur;
        const i64 g_now = g;
        if (has_next) {
            cur = resolve(gn, nraw);
            staged = cur.span <= SEGW_CAP;
            if (staged) wave_stage_load<NP>(x, cur.lo, cur.span, ra, rb);
        }
        const i64 gnn = gn + group_stride;
        const bool has_nn = has_next && gnn * gs < n_segs;
        pre_cur = pre_nxt;
        if (has_nn) {
            nraw = load_raw(gnn);
            pre_nxt = pre(gnn * gs + lane < n_segs ? gnn * gs + lane : n_segs - 1);
        }
        if (was_staged) s = seq_sum_lds(lds, now.ja, now.jb);
        if (now.ok) emit(g_now * gs + lane, s, (i64)(now.jb - now.ja), pre_now);
        if (!has_next) break;
        g = gn; gn = gnn; has_next = has_nn;
    }
}

// ordered stream compaction over [0, n): emit(i, out_index) for every i with pred(i), output
// indices ascending in i.  All threads call; returns the number emitted.  s_w: >= SEL_NT/64 i64.
// A step covers CB * SEL_NT items and costs two workgroup barriers; a wavefront takes CB * 64
// consecutive items as CB rows of 64 (lane = column: coalesced loads), so the rank of an item
// inside the wave comes from the row ballots alone (popcounts, no shuffle scan).
// The predicate is split into load(i) -- the memory reads of item i, all CB of them issued
// before the first is looked at -- and pred(i, loaded), which may have side effects (counters,
// LDS histograms: those pin the loads of a one-piece predicate in program order, one memory
// round trip per item).
#define CB 8
template <class Load, class Pred, class Emit>
__device__ i64 block_compact(i64 n, Load load, Pred pred, Emit emit, i64 *s_w)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u64 below = (1ull << lane) - 1ull;
    i64 run = 0;
    for (i64 base = 0; base < n; base += (i64)CB * SEL_NT) {
        const i64 i0 = base + (i64)w * (CB * 64) + lane;
        decltype(load((i64)0)) ld[CB];
#pragma unroll
        for (int k = 0; k < CB; k++) ld[k] = load(i0 + 64 * k < n ? i0 + 64 * k : n - 1);
        u64 m[CB];
        int cw = 0;
#pragma unroll
        for (int k = 0; k < CB; k++) {
            const i64 i = i0 + 64 * k;
            m[k] = __ballot(i < n && pred(i, ld[k]));
            cw += __popcll(m[k]);
        }
        if (lane == 0) s_w[w] = cw;
        __syncthreads();
        i64 off = 0, tot = 0;
        for (int q = 0; q < SEL_NT / 64; q++) { i64 cc = s_w[q]; off += q < w ? cc : 0; tot += cc; }
        i64 o = run + off;
#pragma unroll
        for (int k = 0; k < CB; k++) {
            if ((m[k] >> lane) & 1ull) emit(i0 + 64 * k, o + __popcll(m[k] & below));
            o += __popcll(m[k]);
        }
        run += tot;
        __syncthreads();
    }
    return run;
}

// Ordered stream compaction with TWO workgroup barriers whatever n: every wavefront takes one
// contiguous chunk of [0, n) (compact_chunk), counts its hits (pass 1: row ballots), the chunk totals
// are exchanged once, and the hits are emitted in a second pass over the same chunk (the predicate is
// evaluated twice: it must be pure).  block_compact above costs two barriers per 4 096 items -- 12
// for the 23 k entries of a 10 kb read's taken list.  A caller that already has the wavefronts'
// hit counts (k_pick: from its pass that counts the scores above the threshold) calls
// compact_chunk_emit alone.
#define COMPACT_RB 16 // rows of 64 whose loads are in flight together
__device__ __forceinline__ void compact_chunk(i64 n, i64 *c0, i64 *c1)
{
    constexpr int NW = SEL_NT / 64;
    const i64 chunk = (((n + NW - 1) / NW) + 63) & ~(i64)63;
    *c0 = (i64)(threadIdx.x >> 6) * chunk;
    *c1 = *c0 + chunk < n ? *c0 + chunk : n;
}
// this wavefront's hits go to emit(i, o, item) with o = off, off + 1, ... in index order
template <class Load, class Pred, class Emit>
__device__ __forceinline__ void compact_chunk_emit(i64 n, i64 off, Load load, Pred pred, Emit emit)
{
    const int lane = threadIdx.x & 63;
    const u64 below = (1ull << lane) - 1ull;
    i64 c0, c1;
    compact_chunk(n, &c0, &c1);
    i64 o = off;
    for (i64 i0 = c0; i0 < c1; i0 += 64 * COMPACT_RB) {
        decltype(load((i64)0)) ld[COMPACT_RB];
#pragma unroll
        for (int k = 0; k < COMPACT_RB; k++) {
            const i64 i = i0 + 64 * k + lane;
            ld[k] = load(i < c1 ? i : (n > 0 ? n - 1 : 0));
        }
#pragma unroll
        for (int k = 0; k < COMPACT_RB; k++) {
            const i64 i = i0 + 64 * k + lane;
            const u64 m = __ballot(i < c1 && pred(i, ld[k]));
            if ((m >> lane) & 1ull) emit(i, o + __popcll(m & below), ld[k]);
            o += __popcll(m);
        }
    }
}
template <class Load, class Pred, class Emit>
__device__ i64 block_compact_chunks(i64 n, Load load, Pred pred, Emit emit, i64 *s_w)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int NW = SEL_NT / 64;
    i64 c0, c1;
    compact_chunk(n, &c0, &c1);
    i64 cnt = 0;
    for (i64 i0 = c0; i0 < c1; i0 += 64 * COMPACT_RB) {
        decltype(load((i64)0)) ld[COMPACT_RB];
#pragma unroll
        for (int k = 0; k < COMPACT_RB; k++) {
            const i64 i = i0 + 64 * k + lane;
            ld[k] = load(i < c1 ? i : (n > 0 ? n - 1 : 0));
        }
#pragma unroll
        for (int k = 0; k < COMPACT_RB; k++) {
            const i64 i = i0 + 64 * k + lane;
            cnt += __popcll(__ballot(i < c1 && pred(i, ld[k])));
        }
    }
    if (lane == 0) s_w[w] = cnt;
    __syncthreads();
    i64 off = 0, tot = 0;
    for (int q = 0; q < NW; q++) { const i64 cc = s_w[q]; off += q < w ? cc : 0; tot += cc; }
    compact_chunk_emit(n, off, load, pred, emit);
    __syncthreads();
    return tot;
}
