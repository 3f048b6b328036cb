// tba_common.h -- shared types + device helpers of the gfx950 resquiggle engine.
// All float64 arithmetic here is two-operand IEEE in the reference's source order; the TU is
// compiled with -ffp-contract=off (no FMA formation) and without fast-math.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tombo_amd.h"

typedef int64_t i64;
typedef uint64_t u64;
typedef int32_t i32;
typedef uint32_t u32;

#define MASK_BASES 50             // _default_parameters.py:69
#define MASK_FILL_Z_SCORE (-15.0) // _default_parameters.py:70
#define DEL_FIX_WINDOW 2          // _default_parameters.py:72
#define MAX_DEL_FIX_WINDOW 10     // _default_parameters.py:73
#define EXTRA_SIG_FACTOR 1.1      // _default_parameters.py:67
#define SHIFT_CHANGE_THRESH 0.1   // _default_parameters.py:169
#define SCALE_CHANGE_THRESH 0.1   // _default_parameters.py:170
#define MAX_TS_POINTS 1000        // _default_parameters.py:178

// optional per-phase cycle stamps into ReadState.dbg: build with -DTBA_PHASE_DEBUG=<kernel>
// (1 k_peaks, 2 k_normalize, 3 k_theil_sen: stamps cumulative from the kernel's start; 4: the
// three parts of k_peaks' tiles, summed over wave 0's tiles; 5: the parts of a k_dp row, summed
// over the rows of the read)
#ifdef TBA_PHASE_DEBUG
#define TBA_PHASE_DEBUG_OR0 TBA_PHASE_DEBUG
#define TBA_PHASE_T0(k_) const i64 tba_t0_ = (k_) == TBA_PHASE_DEBUG ? (i64)__builtin_readcyclecounter() : 0; \
    const i64 tba_w0_ = (k_) == TBA_PHASE_DEBUG ? (i64)__builtin_amdgcn_s_memrealtime() : 0
// dbg[7]: the same interval on the constant 100 MHz counter (gives the shader clock of the run)
#define TBA_PHASE_END(k_) do { if ((k_) == TBA_PHASE_DEBUG && threadIdx.x == 0) r.dbg[7] = (i64)__builtin_amdgcn_s_memrealtime() - tba_w0_; } while (0)
#define TBA_PHASE(k_, i_) do { if ((k_) == TBA_PHASE_DEBUG && threadIdx.x == 0) r.dbg[i_] = (i64)__builtin_readcyclecounter() - tba_t0_; } while (0)
#else
#define TBA_PHASE_DEBUG_OR0 0
#define TBA_PHASE_T0(k_) do { } while (0)
#define TBA_PHASE(k_, i_) do { } while (0)
#define TBA_PHASE_END(k_) do { } while (0)
#endif

enum { PATH_NONE = 0, PATH_ADAPTIVE = 1, PATH_STATIC = 2 };
enum { ST_NONE = 0, ST_TRY = 1, ST_OK = 2, ST_RETRY = 3, ST_STATIC = 4 };

// per-read state carried from kernel to kernel (device memory, one per read)
struct ReadState {
    i64 raw_off, n_raw;       // into raw / norm / score arrays (score arrays use raw_off + idx)
    i64 seq_off, seq_len;     // into seq codes
    i64 ref_off, B;           // into ref_means / ref_sds / band_starts / lo / hi (B per read)
    i64 seg_off;              // into segs-like arrays (B+1 per read) = ref_off + read index
    i64 ev_off, num_events;   // into valid_cpts / event_means (capacity num_events per read)
    i64 stall_off, n_stall;
    i32 status, path, start_state, sv_flags;
    i32 n_start_calls, changed, pad0;
    i32 is_long;              // more than TBA_LONG_RAW samples or TBA_LONG_BASES bases (k_long.h)
    double shift, scale, lower, upper; // scale values in force after segment_signal
    i32 has_lims;
    i32 tb_done;              // main traceback: 0 to be walked by the serial kernels, 2 walked chunk-parallel and awaiting
                              // k_tb_par_verify, 1 finished (k_tb_par.h)
    i32 ed_flag, dp_wg; // ed_flag: event detection: 1 = this read needs the kernels that keep the scores (k_detect.h); dp_wg: unused since round 6 (0)
    i32 ed_form, tb_form;     // which kernels produced this read's change points / main traceback
    i32 strip_s0;             // first band cell of the centre strip k_dp keeps beside the move rows (-1: none; k_dp.h)
    i32 bad_seq;              // k_ref_levels on the side stream found a base outside ACGT (applied in stage order: k_seq_status)
    i32 tb_verify_fail, pad1; // rows where k_tb_par_verify disagreed with the chunk-parallel traceback (k_tb_par.h; 0 expected)
                              // (TBA_ED_FORM_* / TBA_TB_FORM_*, include/tombo_amd.h: TBA_GET_ED_FORM / TBA_GET_TB_FORM)
    i64 n_taken;              // entries of the taken (score, position) list k_detect left
    double ed_min, ed_max;    // ... and the range of its scores
    i64 n_cpts, n_ev;
    double start_res[4];      // (loc, events_per_base) of start-discovery call 0 / 1
    i64 mapped_start; double epb;
    i64 clip, offset, W, n_static, moves_off, top_pos;
    i64 read_start, norm_len, dp_read_start;
    i64 n_win, skip_off;      // deletion windows of this read, scratch arena offset
    double ts[4]; double score;
    i64 dbg[8];               // phase cycle counters (profiling aid)
};

struct DevParams {
    tba_params p;
    tba_opts o;
    i64 kmer_width, central_pos;
    double fill_masked; // (MASK_FILL_Z_SCORE - z_shift) + z_shift, the round trip of
                        // resquiggle.py:665-668,678
    i32 dp_wg_mode, pad; // (unused since round 6: the workgroup-per-read forward pass is gone; kept for the struct's size)
};

// Two consecutive float64 as ONE 16-byte memory access at 8-byte alignment (global_load /
// global_store_dwordx4 only need dword alignment on gfx9).  An 8-byte access per lane runs at
// 0.54-0.70 x the rate of a 16-byte one (MI355X_MICROARCH.md), which is what held the streaming
// kernels of this pipeline at 2.7-3.6 TB/s through round 3: every pass over a float64 signal
// goes through these now.  Reads of a read's slice are ragged (8-byte aligned starts, odd
// lengths): the callers handle the odd last element.
typedef double f64x2_u __attribute__((ext_vector_type(2), aligned(8)));
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef short i16x2_u __attribute__((ext_vector_type(2), aligned(2)));
__device__ __forceinline__ void ld2(const double *p, double &a, double &b)
{
    const f64x2_u v = *(const f64x2_u *)p;
    a = v.x; b = v.y;
}
__device__ __forceinline__ void st2(double *p, double a, double b)
{
    f64x2_u v;
    v.x = a; v.y = b;
    *(f64x2_u *)p = v;
}

// raw samples of one read as float64 (exact widening of float / int16 input)
template <class RT>
struct RawSamples {
    const RT *p;
    __device__ __forceinline__ double operator[](i64 i) const { return (double)p[i]; }
    __device__ __forceinline__ RawSamples operator+(i64 k) const { return RawSamples{p + k}; }
};
// samples i and i + 1 of a signal in one access
__device__ __forceinline__ void sig_pair(const double *x, i64 i, double &a, double &b) { ld2(x + i, a, b); }
__device__ __forceinline__ void sig_pair(RawSamples<double> x, i64 i, double &a, double &b) { ld2(x.p + i, a, b); }
__device__ __forceinline__ void sig_pair(RawSamples<float> x, i64 i, double &a, double &b)
{
    const f32x2_u v = *(const f32x2_u *)(x.p + i);
    a = (double)v.x; b = (double)v.y;
}
__device__ __forceinline__ void sig_pair(RawSamples<int16_t> x, i64 i, double &a, double &b)
{
    const i16x2_u v = *(const i16x2_u *)(x.p + i);
    a = (double)v.x; b = (double)v.y;
}
template <class Sig> // anything else that can be indexed (FinalSignal: rescaled on the fly)
__device__ __forceinline__ void sig_pair(Sig x, i64 i, double &a, double &b) { a = x[i]; b = x[i + 1]; }

// a / b for a row-constant divisor, bit-identical to IEEE division: y = RN(1/b) comes from one
// true division per row, q0 = RN(a*y) is within 2 ulp, one residual correction makes it
// faithful, a second one rounds correctly (Markstein: q faithful, y = RN(1/b), r = a - b*q exact
// => RN(q + r*y) = RN(a/b)).  Results that underflow are far below the 1e-17 granularity of the
// z_shift they are subtracted from.  tests/test_gpu_kernel_abi.py checks it against true division.
__device__ __forceinline__ double div_by_recip(double a, double b, double y)
{
    double q = a * y;
    double e = __builtin_fma(-b, q, a);
    q = __builtin_fma(e, y, q);
    e = __builtin_fma(-b, q, a);
    return __builtin_fma(e, y, q);
}

// The same quotient in FOUR instructions, for a divisor whose reciprocal is kept as a pair (k_dp: per row):
// yh = RN(1/b), yl = RN(RN(1 - b yh) yh) -- 1 - b yh is exact in an fma, so yh + yl = 1/b to 2^-105.
// q0 = RN(a yh + RN(a yl)) is then within half an ulp + 2^-104 of a/b: faithful at once, and Markstein's
// correction (r = a - b q0 exact, RN(q0 + r yh) = RN(a/b)) finishes it -- one correction round less than
// div_by_recip.  Not three instructions: a quotient of two doubles can sit 2^-107 from a rounding
// boundary, closer than q0's error bound.  100 M random and edge quotients equal a / b on the host
// (gcc fma), and tests/test_gpu_kernel_abi.py::test_row_constant_division_is_ieee holds both forms.
// k_dp uses it under -DTBA_DP_DIV4 only: eight float64 instructions less per row and two v_readlane
// more came out 1.0 ms SLOWER on cfg2 (profiles/r06_k_dp_division_ab.txt).
__device__ __forceinline__ double div_by_recip2(double a, double b, double yh, double yl)
{
    const double p = a * yl;
    const double q = __builtin_fma(a, yh, p);
    const double e = __builtin_fma(-b, q, a);
    return __builtin_fma(e, yh, q);
}
__device__ __forceinline__ double recip_low(double b, double yh) { return __builtin_fma(-b, yh, 1.0) * yh; }

// Lanes of ONE wavefront handing values to each other through GLOBAL memory (k_main_tb_par: read_tb
// entries between the lanes of a read; k_skip_dp_wave: the boundaries lane 0 found).
// __threadfence_block() / __syncthreads() are not the fence for that in a workgroup of a single
// wavefront: the compiler narrows workgroup scope to wavefront scope and emits NOTHING -- no s_waitcnt
// between the stores and the loads that follow (seen in the ISA).  This one is explicit: all stores
// performed, then the vector L1 dropped.  It is hygiene, not a cure: the run-dependent traceback of
// round 5 was first blamed on this handoff, and round 6 showed that it was not -- the same failures with
// this fence, with a poisoned read_tb and with system-coherent loads; the lanes entered phase B with
// identical states and COMPUTED different rows, as a function of the kernel's VGPR allocation
// (profiles/r06_traceback_rootcause.txt).
__device__ __forceinline__ void wave_mem_fence()
{
#if defined(__gfx942__) || defined(__gfx950__)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tbuffer_inv sc1" ::: "memory");
#else
    __threadfence();
#endif
}

// whole-wave shift by one lane on the DPP crossbar (no LDS round trip): lane i <- lane i-1 /
// lane i <- lane i+1; the lane without a source takes the given value
__device__ __forceinline__ double wave_shr1_f64(double x, double lane0_val)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(__double2loint(lane0_val), lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(lane0_val), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// lane i <- lane i-1, lane 0 <- +0.0: bound_ctrl fills the lane without a source with zero bits, so
// the DPP move needs no prepared destination (the form above costs two v_mov of the fill value per
// call -- its `old` operand is tied to the destination register)
__device__ __forceinline__ double wave_shr1_f64_zero(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shl1_f64(double x, double lane63_val) // lane i <- lane i+1
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(__double2loint(lane63_val), lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(lane63_val), hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// order-preserving map double -> u64 (ascending); no NaNs on this path
__device__ __forceinline__ u64 f64_key(double x)
{
    u64 b = (u64)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(u64 k)
{
    u64 b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__device__ __forceinline__ double shfl_f64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_up_f64(double v, int d)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, d, 64);
    hi = __shfl_up(hi, d, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_xor_f64(double v, int m)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ i64 shfl_i64(i64 v, int src)
{
    int lo = (int)(v & 0xffffffff), hi = (int)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((i64)hi << 32) | (u32)lo;
}

// numpy pairwise summation of a contiguous float64 vector (DOUBLE_pairwise_sum); n <= 8192.
// Iterative form of the recursion n -> (n2 = n/2 - (n/2)%8, n - n2) with leaves of <= 128.
__device__ inline double np_pairwise_leaf(const double *a, i64 n)
{
    if (n < 8) {
        double res = 0.;
        for (i64 i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    i64 i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i + 0]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}
__device__ inline double np_pairwise_sum(const double *a, i64 n)
{
    if (n <= 128) return np_pairwise_leaf(a, n);
    // explicit post-order walk; depth <= 7 for n <= 8192
    const double *sa[16]; i64 sn[16]; double sv[16]; int sst[16];
    int sp = 0;
    sa[0] = a; sn[0] = n; sst[0] = 0;
    double ret = 0;
    while (sp >= 0) {
        if (sn[sp] <= 128) {
            ret = np_pairwise_leaf(sa[sp], sn[sp]);
            sp--;
            continue;
        }
        i64 n2 = sn[sp] / 2;
        n2 -= n2 % 8;
        if (sst[sp] == 0) {          // descend left
            sst[sp] = 1;
            sa[sp + 1] = sa[sp]; sn[sp + 1] = n2; sst[sp + 1] = 0;
            sp++;
        } else if (sst[sp] == 1) {   // left done -> descend right
            sv[sp] = ret;
            sst[sp] = 2;
            sa[sp + 1] = sa[sp] + n2; sn[sp + 1] = sn[sp] - n2; sst[sp + 1] = 0;
            sp++;
        } else {                     // both done
            ret = sv[sp] + ret;
            sp--;
        }
    }
    return ret;
}
// np.add.reduce of a contiguous float64 vector: 8192-element buffer chunks, each pairwise
// summed, accumulated left to right onto 0.0 (pinned against numpy in tests)
__device__ inline double np_sum(const double *a, i64 n)
{
    double acc = 0.0;
    for (i64 i = 0; i < n; i += 8192) acc += np_pairwise_sum(a + i, n - i < 8192 ? n - i : 8192);
    return acc;
}

// np.linspace(start, stop, num)[i]
__device__ __forceinline__ double np_linspace_at(double start, double stop, i64 num, i64 i)
{
    i64 div = num - 1;
    double delta = stop - start;
    if (div <= 0) return (0.0 * delta) + start;
    if (i == num - 1) return stop;
    double step = delta / (double)div;
    if (step == 0.0) return (((double)i / (double)div) * delta) + start;
    return ((double)i * step) + start;
}
