// k_dp.h -- event-to-sequence banded dynamic programming: one read per wavefront.
//
// Restates c_banded_forward_pass / c_adaptive_banded_forward_pass / c_process_band / c_argmax /
// c_banded_traceback (_c_dynamic_programming.pyx:186-412) and their callers
// find_seq_start_in_events, _get_masked_start_fwd_pass, find_static_base_assignment,
// find_adaptive_base_assignment (resquiggle.py:547-1050).
//
// Layout: each lane owns CPL contiguous band cells and keeps its cells of the previous row in
// registers; the cells a row needs from it ("previous row shifted by the band offset") are a
// compile-time register renaming per offset plus a few DPP moves for the cells of the next lane
// (cand_row), no LDS round trip.  LDS only holds a ring of the read's event means around the band.
// Cell update (pyx:213-234): v[b] = max(stay, diag, skip) with
// stay = (v[b-1] - stay_pen) + z[b] serial along the row.  The diag/skip candidates are
// independent per cell; the stay chain is resolved *exactly* by a monotone fixed-point sweep:
// every lane walks its CPL cells from a guessed incoming value (-inf first), lanes exchange
// chunk-exit values with a wave shuffle, and the sweep repeats until no incoming value changes.
// At the fixed point every cell was produced by the reference's own expression from the exact
// left neighbour (induction from lane 0), so rows, moves and the argmax-driven band placement
// are bit-identical to the sequential code.  No MFMA: this is a scalar max-plus recurrence.
#pragma once
#include "tba_common.h"
#include <math.h>

__host__ __device__ inline int cpl_class(i64 W)
{
    // 5: Tombo's default DNA bandwidth (300) without a third of the lanes' cells idle
    const int cls[] = {4, 5, 8, 12, 16, 24, 32, 48};
    for (int i = 0; i < 8; i++) if ((i64)cls[i] * 64 >= W) return cls[i];
    return 0;
}

// (Rounds 4-5 kept a second form of the main forward pass in the tree, a workgroup per read (k_dp_wgm.h): built as
// the latency form, bit-identical, 20.3 ms against this kernel's 12.9 on a 10 kb read -- profiles/r04_dp_workgroup_form.txt
// says why.  Removed in round 6 with its switch, tba_engine_set_dp_workgroup_batch: git history has it.)


enum { DP_START_TRY = 0, DP_START_RETRY = 1, DP_MAIN = 2, DP_DIRECT = 3 };

// Narrow adaptive bands are run several reads per wavefront (k_dp_multi.h): the class of a
// bandwidth there (cells per lane, reads per wavefront; cpl == 0: none, k_dp takes the read)
struct DpMultiClass { int cpl, rpw; };
__host__ __device__ inline DpMultiClass dp_multi_class(i64 W)
{
#ifdef TBA_NO_DP_MULTI
    (void)W;
    return {0, 0};
#else
    // Measured (MI355X, 10 k reads x 10 kb unless noted; profiles/r03_dp_multi_classes.txt):
    //   W = 100 (2 kb reads): 2 reads x 32 lanes x 4 cells 7.6 ms, 4 x 16 x 8 8.0 ms, k_dp<4> 8.4 ms
    //           (40 k reads per launch: 24.7 / 28.9 / 28.8 ms)
    //   W = 200: 2 x 32 x 8 cells 74 ms (LDS bank conflicts: a lane stride of 64 bytes puts 32 lanes
    //           on 4 bank pairs, SQ_LDS_BANK_CONFLICT = 82 % of the LDS cycles), k_dp<4> 42 ms
    //   W = 300: 2 x 32 x 10 cells 62 ms (508 VALU instructions per 2-read row = 254 per read against
    //           ~290, but 2.75 waves per SIMD instead of 4 and an LDS round trip on the row chain:
    //           66 % VALU utilisation), k_dp<5> 57 ms
    // so only the narrowest class is dispatched; -DTBA_DPM_WIDE builds the other two for A/B runs.
#ifdef TBA_DPM_128_84
    if (W <= 128) return {8, 4};
#else
    if (W <= 128) return {4, 2};
#endif
#ifdef TBA_DPM_WIDE
    if (W <= 256) return {8, 2};
    if (W <= 320) return {10, 2};
#endif
    return {0, 0};
#endif
}

// one forward pass described explicitly (per-kernel C ABI entry points, tba_c_*): same row
// engine, geometry taken from here instead of ReadState
struct DpJob {
    i64 W, n_rows, row0, n_static, n_ev;
    const double *ev, *mu, *sd;   // event means, per-row level mean / sd (unused with zmat)
    const double *zmat;           // precomputed shifted z-scores [n_rows][W] or NULL
    i64 *starts;                  // band starts [n_rows] (static rows in, adaptive rows out)
    const double *init_row;       // forward row `row0` [W] or NULL (zeros)
    double *fwd_out;              // all forward rows [(n_rows+1)][W] or NULL
    double *z_out;                // shifted z-scores of rows row0.. [(n_rows-row0)][W] or NULL (return_z_scores, pyx:339,387)
    unsigned char *mv;            // moves, rows of 64*CPL bytes
    double z_shift, skip_pen, stay_pen, max_half_z, fill;
    i32 winsor, status;
    i64 top_pos;
};

// wave-level helpers on the DPP crossbar (no LDS round trip): lane i <- lane i-1, and a full
// wave max (quad_perm / row_ror / row_bcast ladder, result broadcast from lane 63)
// (wave_shr1_f64 / wave_shl1_f64: tba_common.h)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float wave_max_f32(float v)
{
    // v_max_f32 with a DPP source, in place: one instruction per ladder step (the builtin form
    // costs a copy, a DPP move and a canonicalising self-max besides the max).  s_nop 1: a DPP
    // read of a VGPR needs two wait states after the VALU write of it.
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_max_f64(double v)
{
    // no NaNs reach this point (a NaN row is caught by the sweep bound), so v_max_f64 == select
    v = __builtin_fmax(v, dpp_mov_f64<0xb1, 0xf>(v));   // quad_perm:[1,0,3,2]
    v = __builtin_fmax(v, dpp_mov_f64<0x4e, 0xf>(v));   // quad_perm:[2,3,0,1]
    v = __builtin_fmax(v, dpp_mov_f64<0x124, 0xf>(v));  // row_ror:4
    v = __builtin_fmax(v, dpp_mov_f64<0x128, 0xf>(v));  // row_ror:8
    v = __builtin_fmax(v, dpp_mov_f64<0x142, 0xa>(v));  // row_bcast:15 -> rows 1,3
    v = __builtin_fmax(v, dpp_mov_f64<0x143, 0xc>(v));  // row_bcast:31 -> rows 2,3
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// bytes of packed 2-bit moves per lane per row: 4 cells per byte, so a row is plainly linear
// (cell b -> byte b/4, bits 2*(b%4)) and 16*CPL bytes long
__host__ __device__ constexpr int mv_bpl(int cpl) { return cpl / 4; } // (whole bytes: cpl % 4 == 0)
// bytes of a packed row of a band class: 64 lanes x cpl cells x 2 bits
__host__ __device__ constexpr int mv_class_rowb(int cpl) { return cpl * 16; }
// bytes of one packed 2-bit move row of a W-cell band: the band class' row (64 lanes x cpl/4
// bytes) or, for bands wider than every class (k_dp_wide), W rounded up to 256 cells
__host__ __device__ inline i64 mv_row_bytes(i64 W)
{
    const int c = cpl_class(W);
    return c ? (i64)mv_class_rowb(c) : ((W + 255) / 256) * 64;
}

// The centre strip.  An adaptive row's band is placed around the best cell of the row before
// (pyx:342-358), and the traceback never strays far from band cell W / 2 (synthetic reads: sd 2.3
// cells; 99.9 % within -20 .. +8).  A move row is a whole 128-byte cache line at W = 500, of which the
// traceback needs 16 bytes: 12.8 GB of lines per 10 000 x 10 kb reads, the whole cost of k_main_tb_par
// (2.4 TB/s of line-granular reads; rotating the row inside its line changes nothing, measured).  So
// the forward pass stores the 64 cells [s0, s0 + 64) around the centre a second time, 16 bytes per row,
// rows back to back behind the read's move rows (same arena: strip row rr at mv + (B + 1) rowb + 16 rr).
// The traceback reads eight rows per line from there and falls back to the full row whenever its
// position leaves the strip's comfortable middle (and in the static rows at the read's start, which
// have no strip).  Classes whose lanes hold whole bytes and divide the strip: 4, 8, 16, 32 cells.
#define MV_STRIP_CELLS 64
#define MV_STRIP_BYTES 16
__host__ __device__ inline int mv_strip_s0(i64 W)
{
#ifdef TBA_NO_MV_STRIP
    (void)W;
    return -1;
#else
    const int c = cpl_class(W);
    if (!(c == 4 || c == 8 || c == 16 || c == 32)) return -1;
    const int unit = c > 16 ? c : 16;                 // dword- and lane-aligned
    const int s0 = (((int)(W / 2) - 40) / unit) * unit; // the centre (W / 2 - 1) sits ~41 cells into the strip
    return s0 >= 0 && s0 + MV_STRIP_CELLS <= 64 * c ? s0 : -1;
#endif
}

// wave-uniform values the compiler cannot prove uniform (they come from vector loads or
// shuffles) are moved to scalar registers explicitly, so that control flow on them is scalar
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ i64 uni(i64 x)
{
    const int lo = __builtin_amdgcn_readfirstlane((int)(x & 0xffffffff));
    const int hi = __builtin_amdgcn_readfirstlane((int)(x >> 32));
    return ((i64)hi << 32) | (u32)lo;
}
template <class T> __device__ __forceinline__ T *uni(T *p) { return (T *)uni((i64)p); }
__device__ __forceinline__ double uni(double x) { return __longlong_as_double((long long)uni((i64)__double_as_longlong(x))); }
__device__ __forceinline__ bool uni(bool x) { return __builtin_amdgcn_readfirstlane((int)x) != 0; }
// lane `sel` (wave-uniform) of a double, as a scalar
__device__ __forceinline__ double readlane_f64(double x, int sel)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), sel);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), sel);
    return __hiloint2double(hi, lo);
}

// v_max_f64 without the canonicalising self-max the fmax builtin puts in front of it (both
// operands are results of adds / maxes here, never signalling NaNs)
__device__ __forceinline__ double max_f64_raw(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Previous-row access without memory: every lane keeps its CPL cells of the previous row in
// registers; the cells a row needs are that row shifted by the (wave-uniform) band offset, i.e. a
// compile-time register renaming per offset plus a few wave_shl DPP moves for the cells that come
// from the next lane.  A[k] = previous-row cell (lane*CPL + D + k - 1), k = 0..CPL.
template <int CPL, int D>
__device__ __forceinline__ void shifted_row(const double (&Q)[CPL], double left, double (&A)[CPL + 1])
{
#pragma unroll
    for (int k = 0; k <= CPL; k++) {
        const int idx = k - 1 + D;
        if (idx < 0) A[k] = left;
        else if (idx < CPL) A[k] = Q[idx];
        else A[k] = wave_shl1_f64(Q[idx - CPL], -INFINITY);
    }
}
// diag / skip candidates of one row for band offset D (pyx:220-231, first cell pyx:392-401):
// A[j] is cell j's diagonal source and cell j-1's skip source.  Instantiated per offset so the
// previous row is read straight out of its registers (no renaming moves).  tk bit j: skip taken.
// -DTBA_DP_SWEEP1_FUSED (A/B switch, off): the FIRST sweep of the stay chain inside this block, so that
// its dependent adds are scheduled between the independent candidate arithmetic.  Measured round 5:
// main_dp 59.97 against 59.63 ms -- with four wavefronts per SIMD the f64 pipe is busy whatever the
// order inside one of them (tools/valu_rates.hip), and nine copies of the sweep cost instruction cache.
template <int CPL, int D>
__device__ __forceinline__ void cand_row(const double (&Q)[CPL], double left,
    const double (&z)[CPL], double skip_pen, double stay_pen, bool first_is_skip, bool lane0, double (&cv)[CPL],
    bool (&tk)[CPL], double &exit0)
{
    double A[CPL + 1];
    shifted_row<CPL, D>(Q, left, A);
    double x = -INFINITY;
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        const double d = A[j] + z[j];
        const double s = A[j + 1] - skip_pen;
        bool take_s = s > d;
        if (j == 0) {
            take_s = lane0 ? first_is_skip : take_s; // band cell 0: skip xor diag
            cv[j] = take_s ? s : d;
        } else {
            // the larger of the two IS the selected one (equal values: either), so one v_max_f64
            // replaces the two 32-bit selects; the move flag still comes from the strict compare
            cv[j] = max_f64_raw(s, d);
        }
        tk[j] = take_s;
#ifdef TBA_DP_SWEEP1_FUSED
        x = max_f64_raw(cv[j], (x - stay_pen) + z[j]);
#endif
    }
    exit0 = x;
}
// Q <- Q shifted by S cells (S <= CPL); returns the cell just left of the new Q[0]
template <int CPL, int S>
__device__ __forceinline__ double shift_cells(double (&Q)[CPL])
{
    double T[CPL];
    const double left = Q[S - 1];
#pragma unroll
    for (int k = 0; k < CPL; k++) T[k] = k + S < CPL ? Q[k + S] : wave_shl1_f64(Q[k + S - CPL], -INFINITY);
#pragma unroll
    for (int k = 0; k < CPL; k++) Q[k] = T[k];
    return left;
}

// Row body: the only branches are wave-uniform (static vs adaptive row, masked row, band offset
// switch, sweep loop).  Data flow per row:
//   events   : linear LDS ring of the read's event means around the band, refilled by coalesced
//              prefetches one chunk ahead (no global-load latency on the row chain)
//   prev row : registers + DPP (above); cells >= W hold -inf so out-of-band candidates need no guards
//   mu/sd    : prefetched one row ahead
template <bool B> struct BoolTag { static constexpr bool value = B; };
template <int I> struct IntTag { static constexpr int value = I; };

template <int CPL, bool DIRECT>
// Waves per SIMD the register allocation of the classes up to 8 cells per lane is held to: 4 (128
// VGPRs; k_dp<8> comes out at 133 without it since the row loop exists twice).  -DTBA_DP_WAVES=5 (96
// VGPRs) was measured SLOWER, 70.4 against 65.6 ms at W = 500: the squeeze costs instructions and
// the LDS ring allows 19 workgroups per CU, not 20.
#ifndef TBA_DP_WAVES
#define TBA_DP_WAVES 4
#endif
// (The 12-cell class -- start discovery at DNA's start_bw = 750 -- comes out at 196 registers = two
// wavefronts per SIMD; held to three (168, no spills: -DTBA_DP12_WAVES=3) it measured 3.94 against 3.85 ms.)
#ifndef TBA_DP12_WAVES
#define TBA_DP12_WAVES 1
#endif
#define TBA_DP_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(CPL <= 8 ? TBA_DP_WAVES : (CPL == 12 ? TBA_DP12_WAVES : 1))))
__device__ __forceinline__ void dp_body(ReadState *rs, const DevParams *dp, int mode,
    const double *event_means, const double *ref_means, const double *ref_sds,
    i64 *band_starts, const i32 *lo_arr, const i32 *hi_arr,
    unsigned char *moves, i64 start_moves_stride, double *last_row, DpJob *job)
{
    constexpr int S = CPL < 8 ? CPL : 8;     // band offsets 0..S take the register-renaming path
    constexpr int RING = CPL <= 4 ? 512 : CPL <= 8 ? 1024 : CPL <= 16 ? 2048 : CPL <= 32 ? 4096 : 8192; // power of two >= 128*CPL
    constexpr int BPL = mv_bpl(CPL);
#ifdef TBA_DP_LDS_PAD
    // experiment: cap the kernel at fewer waves per SIMD through its LDS footprint, so that the
    // memory-bound kernels of another sub-batch (other stream) find registers beside it
    __shared__ double ring[RING + CPL + TBA_DP_LDS_PAD];
#else
    __shared__ double ring[RING + CPL];      // + mirror of the first CPL slots: reads never wrap
#endif
#ifdef TBA_DP_VGPR_PAD
    // experiment (profiles/r06_coresidency_ab.txt): a named clobber raises the kernel's VGPR allocation -- "v135": 136
    // registers, three wavefronts on a SIMD and 104 registers left for another stream's kernels, LDS untouched
    asm volatile("" ::: TBA_DP_VGPR_PAD);
#endif
    ReadState &r = rs[DIRECT ? 0 : blockIdx.x];
    if (!DIRECT && r.status != TBA_OK) return;
    const tba_params &P = dp->p;
    const int lane = threadIdx.x;

    // row / event indices fit 32 bits: scalar compares and adds instead of 64-bit VALU ones
    int W, n_rows, n_static, n_ev, row0 = 0;
    bool identity;
    unsigned char *mv;
    const double *ev, *rmu, *rsd, *zmat = nullptr;
    i64 *bst;
    const i32 *lo_a = lo_arr, *hi_a = hi_arr;
    double stay_pen, skip_pen, z_shift, max_half_z, fill_masked;
    bool winsor;
    if constexpr (DIRECT) {
        W = (int)job->W; n_rows = (int)job->n_rows; n_static = (int)job->n_static; n_ev = (int)job->n_ev;
        row0 = (int)job->row0;
        identity = false;
        mv = job->mv;
        ev = job->ev; rmu = job->mu; rsd = job->sd; zmat = job->zmat; bst = job->starts;
        stay_pen = job->stay_pen; skip_pen = job->skip_pen; z_shift = job->z_shift;
        max_half_z = job->max_half_z; fill_masked = job->fill; winsor = job->winsor != 0;
    } else {
        i64 ev_base;
        if (mode == DP_MAIN) {
            if (r.path == PATH_NONE) return;
            W = (int)r.W;
            if (cpl_class(W) != CPL) return;
            // an adaptive read at a narrow batch bandwidth belongs to k_dp_multi
            if (r.path == PATH_ADAPTIVE && r.W == P.bandwidth && dp_multi_class(P.bandwidth).cpl != 0) return;
            n_rows = (int)r.B;
            n_static = (int)r.n_static;
            ev_base = r.ev_off + r.clip;
            n_ev = (int)(r.n_ev - r.clip);
            identity = false;
            mv = moves + uni(r.moves_off);
        } else {
            if (r.start_state != (mode == DP_START_TRY ? ST_TRY : ST_RETRY)) return;
            W = (int)(mode == DP_START_TRY ? P.start_bw : P.start_save_bw);
            n_rows = (int)P.start_n_bases;
            n_static = n_rows;
            ev_base = r.ev_off;
            n_ev = (int)r.n_ev;
            identity = true;
            mv = moves + (i64)blockIdx.x * start_moves_stride;
        }
        const i64 ro = uni(r.ref_off);
        ev = event_means + uni(ev_base);
        rmu = ref_means + ro;
        rsd = ref_sds + ro;
        bst = band_starts + ro;
        lo_a = lo_arr + ro;
        hi_a = hi_arr + ro;
        stay_pen = P.stay_pen; skip_pen = P.skip_pen; z_shift = P.z_shift;
        max_half_z = P.max_half_z_score;
        winsor = P.do_winsorize_z != 0;
        fill_masked = dp->fill_masked;
    }
#ifndef TBA_NO_DP_PRIO
    // The SIMD's arbiter serves the oldest wavefront first.  Workgroups are dispatched in index
    // order, so the last ones of the launch -- the wavefronts its end waits for -- are the youngest
    // on their SIMDs for all of their lives: they wait while the older ones run at lone speed and
    // finish long after them (4 096 equal reads started together end between 16 and 30 ms).
    // Priority by dispatch order over the last four wavefronts per SIMD (s_setprio, 4 levels; 1 024
    // SIMDs on an MI355X): a late starter runs at lone speed, the rest of the launch is as it was.
    // (Priority by PROGRESS -- fewer rows done, served first -- was measured too: it bunches the
    // wavefronts of a SIMD, which then drains in steps of four or five reads: -2 % at W = 500,
    // +4 % at W = 300.)
    if (!DIRECT && mode == DP_MAIN) {
        const unsigned from_end = gridDim.x - 1u - blockIdx.x;
#ifndef TBA_DP_PRIO_LEVELS
#define TBA_DP_PRIO_LEVELS 3
#endif
        if (from_end < 1024u) __builtin_amdgcn_s_setprio(TBA_DP_PRIO_LEVELS);
        else if (TBA_DP_PRIO_LEVELS > 1 && from_end < 2048u) __builtin_amdgcn_s_setprio(TBA_DP_PRIO_LEVELS - 1);
        else if (TBA_DP_PRIO_LEVELS > 2 && from_end < 3072u) __builtin_amdgcn_s_setprio(TBA_DP_PRIO_LEVELS - 2);
    }
#endif
    // everything above came through vector loads: make it scalar once
    W = uni(W); n_rows = uni(n_rows); n_static = uni(n_static); n_ev = uni(n_ev); row0 = uni(row0);
    identity = uni(identity); winsor = uni(winsor);
    stay_pen = uni(stay_pen); skip_pen = uni(skip_pen); z_shift = uni(z_shift);
    max_half_z = uni(max_half_z); fill_masked = uni(fill_masked);
    const bool use_z = DIRECT && zmat != nullptr;
    const int Wi = W;
    const int half_bw = W / 2; // integer division, pyx:327
    const double NEG_INF = -INFINITY;
    const i64 mv_stride = mv_class_rowb(CPL);
    // centre strip of the adaptive rows (main forward pass of the batch pipeline only)
    constexpr bool STRIP_CLASS = !DIRECT && (CPL == 4 || CPL == 8 || CPL == 16 || CPL == 32);
    int strip_l0 = -1;           // first lane of the strip (-1: this read keeps none)
    if constexpr (STRIP_CLASS) {
        if (mode == DP_MAIN) { const int s0 = uni(r.strip_s0); strip_l0 = s0 >= 0 ? s0 / CPL : -1; }
    }
    const int b0 = lane * CPL;   // my first band cell
    int nvalid = Wi - b0;        // how many of my cells are inside the band
    nvalid = nvalid < 0 ? 0 : (nvalid > CPL ? CPL : nvalid);
    const u64 has_cells = __ballot(nvalid > 0);         // lanes that hold band cells
    const double zcap = winsor ? max_half_z : INFINITY; // no winsorising: clamp at +inf
    double zs[CPL]; // z_shift per cell, -inf for the cells past the band (folds the band-end mask
                    // into the subtraction that forms z)
#pragma unroll
    for (int j = 0; j < CPL; j++) zs[j] = j < nvalid ? z_shift : NEG_INF;

    double v[CPL]; // my cells of the previous row (cells past the band: -inf for good)
    int prev_start = 0;
    int am = 0; // argmax of the previous row (row 0: all zeros -> 0)
    if (DIRECT && job->init_row != nullptr) {
        // resume from a given forward row (c_adaptive_banded_forward_pass is handed rows
        // 0..start_seq_pos): load it, take its argmax
        double lmax = NEG_INF;
        int lidx = 0;
#pragma unroll
        for (int j = 0; j < CPL; j++) {
            const int b = b0 + j;
            double x = job->init_row[j < nvalid ? b : Wi - 1];
            x = j < nvalid ? x : NEG_INF;
            v[j] = x;
            const bool better = x > lmax;
            lmax = better ? x : lmax;
            lidx = better ? b : lidx;
        }
        const double wm = wave_max_f64(lmax);
        u64 eq = __ballot(lmax == wm && nvalid > 0);
        am = uni(__shfl(lidx, __ffsll((unsigned long long)eq) - 1, 64));
        if (row0 > 0) prev_start = uni((int)bst[row0 - 1]);
    } else {
#pragma unroll
        for (int j = 0; j < CPL; j++) v[j] = j < nvalid ? 0.0 : NEG_INF; // row 0: zeros (pyx:253-254)
    }
    if (DIRECT && job->fwd_out != nullptr && row0 == 0) {
#pragma unroll
        for (int j = 0; j < CPL; j++) job->fwd_out[b0 + j] = 0.0; // rows padded to 64*CPL
    }

    // event ring: absolute event index a lives at slot (a + RING) & (RING - 1) (plus a mirror of
    // slots 0..CPL-1 behind the end); filled = first absolute index not yet loaded
    int filled = 0;
    auto ring_store = [&](int a, double x) {
        const int sl = (a + RING) & (RING - 1);
        ring[sl] = x;
        if (sl < CPL) ring[RING + sl] = x;
    };
    auto ev_load = [&](int a) { // clamped load + select
        const int ac = a < 0 ? 0 : (a >= n_ev ? n_ev - 1 : a);
        const double x = ev[ac];
        return (a >= 0 && a < n_ev) ? x : 0.0;
    };
    double pf = 0.0;      // one prefetched chunk (event pf_at + lane), in flight
    int pf_at = 0;
    bool pf_pending = false;
    if (!use_z) {
        // the first band start positions the ring: [start, start + RING - 128) is loaded up front
        int first_start = row0 < n_static ? (identity ? row0 : uni((int)bst[row0])) : prev_start;
        filled = first_start;
        for (int c = 0; c < RING / 64 - 2; c++) { ring_store(filled + lane, ev_load(filled + lane)); filled += 64; }
    }
    __syncthreads();

    // expected level, sd and its reciprocal of 64 rows at a time: lane l holds row blk0 + l
    // (one coalesced load and one true division per 64 rows), a row reads its lane
#ifndef TBA_DP_DIV4
    double mu_v = 0, sd_v = 1, y_v = 1;
#else
    double mu_v = 0, sd_v = 1, y_v = 1, yl_v = 0;   // (yl_v: the low word of the reciprocal, div_by_recip2)
#endif
    auto load_levels = [&](int first) {
        int rc = first + lane;
        rc = rc < n_rows ? rc : n_rows - 1;
        mu_v = rmu[rc]; sd_v = rsd[rc];
        y_v = 1.0 / sd_v;
#ifdef TBA_DP_DIV4
        yl_v = recip_low(sd_v, y_v);
#endif
        // the loads end HERE, once per 64 rows: left pending, the rows' read of these registers
        // sits behind a conditional load and the compiler guards it with s_waitcnt vmcnt(0) in
        // EVERY row (which on gfx9 also waits for the previous row's stores)
        asm volatile("" : "+v"(mu_v));
    };
    if (!use_z) load_levels(row0);
    // Per-row band geometry of the static rows, loaded one row ahead.  The three registers are
    // touched by nothing but these loads and the static rows' read of them: an adaptive row that
    // as much as selects on one of them gets an s_waitcnt vmcnt(0) from the compiler (the register
    // MAY have a load in flight), which on gfx9 also waits for the previous row's stores.
    int st_n = 0, lo_n = 0, hi_n = Wi;
    auto fetch_row = [&](int rr) {
        const int rc = rr < n_rows ? rr : n_rows - 1;
        if (rc < n_static && !identity) {
            st_n = (int)bst[rc];
            if (!DIRECT) { lo_n = lo_a[rc]; hi_n = hi_a[rc]; }
        }
    };
    fetch_row(row0);

#ifdef TBA_SWEEP_STATS
    i64 sw_total = 0;
#endif
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 5
    // cycles of wave 0's rows by part: 0 z-scores (ring reads + arithmetic), 1 candidates, 2 first
    // scan + sweeps, 3 cells / flags / stores / ring upkeep, 4 arg-max, 5 band placement; 6 rows
    // dbg[7]: where the wavefront ran: HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13])
    // | XCC_ID << 32
    i64 ph[7] = {0, 0, 0, 0, 0, 0, 0};
    i64 ph_t = (i64)__builtin_readcyclecounter();
    u32 hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const i64 ph_first = (i64)hw_id | ((i64)(xcc_id & 15) << 32);
#define DP_PH(i_) do { const i64 t_ = (i64)__builtin_readcyclecounter(); ph[i_] += t_ - ph_t; ph_t = t_; } while (0)
#else
#define DP_PH(i_) do { } while (0)
#endif
    // One row of the forward pass.  ADAPT: the row is past the static rows (compile-time: no
    // band-geometry loads, no masked-start test -- a wavefront on its own issues ONE instruction
    // of any kind per >= 4 cycles, so the scalar bookkeeping of the general row costs it as much
    // as the arithmetic).  Returns true when the read has failed.
    auto row_step = [&](const int row, auto adapt_tag) __attribute__((always_inline)) -> bool {
        constexpr bool ADAPT = decltype(adapt_tag)::value;
        double mu = 0, sd = 1, y = 1;
#ifdef TBA_DP_DIV4
        double yl = 0;
#endif
        if (!use_z) {
            const int sel = (row - row0) & 63;
            if (sel == 0 && row != row0) load_levels(row);
            mu = readlane_f64(mu_v, sel); sd = readlane_f64(sd_v, sel); y = readlane_f64(y_v, sel);
#ifdef TBA_DP_DIV4
            yl = readlane_f64(yl_v, sel);
#endif
        }
        int cur_start;
        int lo, hi;
        double fill;
        if (!ADAPT) {
            if (identity) { cur_start = row; lo = 0; hi = Wi; }                 // start discovery
            else if (DIRECT) { cur_start = uni(st_n); lo = 0; hi = Wi; }
            else { cur_start = uni(st_n); lo = uni(lo_n); hi = uni(hi_n); }
            fill = fill_masked;
        } else {
            // adaptive band placement, pyx:342-358
            cur_start = prev_start + am - half_bw + 1;
            if (cur_start < prev_start) cur_start = prev_start;
            if (cur_start >= n_ev) {
                if (row < n_rows - 2) {
                    if (lane == 0) { if (DIRECT) job->status = TBA_ADAPT_BEYOND; else r.status = TBA_ADAPT_BEYOND; }
                    return true;
                }
                cur_start = n_ev - 1;
            }
            if (lane == 0) bst[row] = cur_start;
            lo = 0;
            hi = cur_start + W <= n_ev ? Wi : n_ev - cur_start;
            fill = DIRECT ? fill_masked : MASK_FILL_Z_SCORE; // literal -15, pyx:385-386
        }
        if (!ADAPT) fetch_row(row + 1); // next row's inputs travel while this row computes
        const int diff_i = row > 0 ? cur_start - prev_start : 0;
        DP_PH(5);

        // shifted half z-scores of my cells (pyx:361-372 / resquiggle.py:574-582,712-720)
        double z[CPL];
        if (use_z) {
#pragma unroll
            for (int j = 0; j < CPL; j++) {
                const double zz = zmat[(i64)row * W + (j < nvalid ? b0 + j : Wi - 1)];
                z[j] = j < nvalid ? zz : NEG_INF;
            }
        } else {
            // make sure the ring covers [cur_start, cur_start + 64*CPL) (only a large band jump
            // gets here; the prefetch below normally stays ahead)
            if (cur_start + 64 * CPL > filled) {
                if (pf_pending) { ring_store(pf_at + lane, pf); filled = pf_at + 64; pf_pending = false; }
                while (cur_start + 64 * CPL > filled) { ring_store(filled + lane, ev_load(filled + lane)); filled += 64; }
            }
            const double *er = ring + ((cur_start + RING + b0) & (RING - 1)); // + j < RING + CPL
#pragma unroll
            for (int j = 0; j < CPL; j++) {
#ifndef TBA_DP_DIV4
                double pz = fabs(div_by_recip(er[j] - mu, sd, y));
#else
                double pz = fabs(div_by_recip2(er[j] - mu, sd, y, yl));
#endif
                pz = __builtin_fmin(pz, zcap);
                z[j] = zs[j] - pz; // cells past the band: -inf - pz = -inf, stays -inf for good
            }
            if (__builtin_expect((!ADAPT && lo != 0) || hi != Wi, 0)) { // masked start rows / band past the last event
#pragma unroll
                for (int j = 0; j < CPL; j++) {
                    z[j] = (b0 + j >= lo && b0 + j < hi) ? z[j] : fill;
                    z[j] = j < nvalid ? z[j] : NEG_INF;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (DIRECT) {
            if (job->z_out != nullptr) { // all_shifted_z_scores[seq_pos - start_seq_pos, :] (pyx:387-388)
#pragma unroll
                for (int j = 0; j < CPL; j++) if (j < nvalid) job->z_out[(i64)(row - row0) * W + b0 + j] = z[j];
            }
        }
        DP_PH(0);
        // diag / skip candidates from the previous row (pyx:220-231), first cell pyx:392-401:
        // pp[j] is cell j's diagonal source and cell j-1's skip source
        double cv[CPL];
        bool tk[CPL];
        double exit0 = NEG_INF;
        {
            int rem = diff_i;
            double left = NEG_INF;
            while (rem > S) { left = shift_cells<CPL, S>(v); rem -= S; } // rare: big band jump
            const bool fs = diff_i == 0, l0 = lane == 0;
            switch (rem) {
            case 0: left = diff_i == 0 ? wave_shr1_f64(v[CPL - 1], NEG_INF) : left;
                    cand_row<CPL, 0>(v, left, z, skip_pen, stay_pen, fs, l0, cv, tk, exit0); break;
            case 1: cand_row<CPL, 1>(v, left, z, skip_pen, stay_pen, fs, l0, cv, tk, exit0); break;
            case 2: if constexpr (S >= 2) cand_row<CPL, 2>(v, left, z, skip_pen, stay_pen, fs, l0, cv, tk, exit0); break;
            case 3: if constexpr (S >= 3) cand_row<CPL, 3>(v, left, z, skip_pen, stay_pen, fs, l0, cv, tk, exit0); break;
            case 4: if constexpr (S >= 4) cand_row<CPL, 4>(v, left, z, skip_pen, stay_pen, fs, l0, cv, tk, exit0); break;
            case 5: if constexpr (S >= 5) cand_row<CPL, 5>(v, left, z, skip_pen, stay_pen, fs, l0, cv, tk, exit0); break;
            case 6: if constexpr (S >= 6) cand_row<CPL, 6>(v, left, z, skip_pen, stay_pen, fs, l0, cv, tk, exit0); break;
            case 7: if constexpr (S >= 7) cand_row<CPL, 7>(v, left, z, skip_pen, stay_pen, fs, l0, cv, tk, exit0); break;
            default: if constexpr (S >= 8) cand_row<CPL, 8>(v, left, z, skip_pen, stay_pen, fs, l0, cv, tk, exit0); break;
            }
        }
        // stay chain: monotone fixed-point sweeps, chunk exit values shifted one lane up.
        // Sweep 1 starts every lane from -inf and gives v0[j], the value of cell j without anything
        // coming in from the left.  x -> (x - stay_pen) + z is monotone, and a monotone map commutes
        // with max, so with an incoming value `in` cell j holds max(v0[j], c_j(in)), c_j = the pure
        // stay chain from `in` through cells 0..j: the later sweeps only carry that chain (2
        // instructions per cell instead of 3) to get the lane's exit value max(v0[CPL-1], c_{CPL-1}),
        // and the cells themselves are written once, in the pass that also derives the move flags.
        // The sequence of incoming values is the same as with full sweeps, so is the sweep count.
        DP_PH(1);
#ifndef TBA_DP_SHR_FILL
        // Band cell 0 has no stay move (pyx:392-401): nothing may come into lane 0.  Instead of
        // handing lane 0 an incoming -inf with every lane shift (two v_mov per sweep to prepare the
        // DPP destination), lane 0's z[0] is -inf for the chain from here on (the candidates above were
        // the last readers of its true value): whatever comes in -- the shift's zero fill -- dies
        // there, (x - stay_pen) + -inf = -inf, exactly the value the reference's missing move has.
        z[0] = lane == 0 ? NEG_INF : z[0];
        double in = lane == 0 ? 0.0 : NEG_INF;
#define DP_SHR(x_) wave_shr1_f64_zero(x_)
#else
        double in = NEG_INF;
#define DP_SHR(x_) wave_shr1_f64(x_, NEG_INF)
#endif
        bool converged = false;
#ifndef TBA_DP_SWEEP1_FUSED
        {
            double x = NEG_INF;
#pragma unroll
            for (int j = 0; j < CPL; j++) x = max_f64_raw(cv[j], (x - stay_pen) + z[j]);
            exit0 = x;
        }
#endif
#ifdef TBA_SWEEP_STATS
        i64 sw_row = 1;
#endif
        {
            double nin = DP_SHR(exit0);
            for (int it = 0; it < 66; it++) { // <= 64 sweeps by induction over lanes (NaN-proof bound)
                if (__ballot(nin != in) == 0) { converged = true; break; }
                in = nin;
                double c = in;
#pragma unroll
                for (int j = 0; j < CPL; j++) c = (c - stay_pen) + z[j];
                nin = DP_SHR(max_f64_raw(exit0, c));
#ifdef TBA_SWEEP_STATS
                sw_row++;
#endif
            }
        }
#ifdef TBA_SWEEP_STATS
        sw_total += sw_row;
#endif
#undef DP_SHR
        if (!converged) { // only reachable with NaNs in the signal
            if (lane == 0) { if (DIRECT) job->status = TBA_INTERNAL; else r.status = TBA_INTERNAL; }
            return true;
        }
        DP_PH(2);
        // the cells, their move codes (0 stay, 1 skip, 2 diag; pyx:216-231) packed 2 bits per cell,
        // lane-local argmax (pyx:186-197; -inf cells never win)
        u32 mvw[(CPL + 15) / 16];
#pragma unroll
        for (int q = 0; q < (CPL + 15) / 16; q++) mvw[q] = 0;
        double lmax = NEG_INF;
        {
            double x = in;
#pragma unroll
            for (int j = 0; j < CPL; j++) {
                const double s = (x - stay_pen) + z[j];
                const u32 f = cv[j] > s ? (tk[j] ? 1u : 2u) : 0u;
                mvw[j / 16] |= f << (2 * (j % 16));
                x = max_f64_raw(cv[j], s);
                v[j] = x;
                lmax = max_f64_raw(lmax, x);
            }
        }
        {
            unsigned char *mrow = mv + (row + 1) * mv_stride + lane * BPL;
            if constexpr (CPL % 4 != 0) {
                // 5 cells = 10 bits per lane: a quad's 40 bits are 5 whole bytes, assembled by
                // its first lane (quad_perm broadcasts) -- the row stays linear in the cell index
                static_assert(CPL == 5, "odd class: 5 cells per lane");
                const u32 q0 = mvw[0];
                const u32 q1 = (u32)__builtin_amdgcn_update_dpp(0, (int)q0, 0x55, 0xf, 0xf, false); // quad_perm:[1,1,1,1]
                const u32 q2 = (u32)__builtin_amdgcn_update_dpp(0, (int)q0, 0xaa, 0xf, 0xf, false); // [2,2,2,2]
                const u32 q3 = (u32)__builtin_amdgcn_update_dpp(0, (int)q0, 0xff, 0xf, 0xf, false); // [3,3,3,3]
                if ((lane & 3) == 0) {
                    const u64 bits = (u64)q0 | ((u64)q1 << 10) | ((u64)q2 << 20) | ((u64)q3 << 30);
                    unsigned char *p = mv + (row + 1) * mv_stride + (lane >> 2) * 5;
#pragma unroll
                    for (int q = 0; q < 5; q++) p[q] = (unsigned char)(bits >> (8 * q));
                }
            } else
            if constexpr (BPL == 1) *mrow = (unsigned char)mvw[0];
            else if constexpr (BPL == 2) *(unsigned short *)mrow = (unsigned short)mvw[0];
            else if constexpr (BPL % 4 == 0) {
#pragma unroll
                for (int q = 0; q < BPL / 4; q++) ((u32 *)mrow)[q] = mvw[q];
            } else { // 3 or 6 bytes per lane
#pragma unroll
                for (int q = 0; q < BPL; q++) mrow[q] = (unsigned char)(mvw[q / 4] >> (8 * (q % 4)));
            }
        }
        if constexpr (STRIP_CLASS && ADAPT) {
            // the strip's lanes store their bytes a second time, rows 16 bytes apart behind the move rows
            if (strip_l0 >= 0) {
                const unsigned rel = (unsigned)(lane - strip_l0);
                if (rel < (unsigned)(MV_STRIP_CELLS / CPL)) {
                    unsigned char *sp = mv + (i64)(n_rows + 1) * mv_stride + (i64)(row + 1) * MV_STRIP_BYTES + rel * BPL;
                    if constexpr (BPL == 1) *sp = (unsigned char)mvw[0];
                    else if constexpr (BPL == 2) *(unsigned short *)sp = (unsigned short)mvw[0];
                    else {
#pragma unroll
                        for (int q = 0; q < BPL / 4; q++) ((u32 *)sp)[q] = mvw[q];
                    }
                }
            }
        }
        if (DIRECT && job->fwd_out != nullptr) {
#pragma unroll
            for (int j = 0; j < CPL; j++) job->fwd_out[(row + 1) * (i64)(64 * CPL) + b0 + j] = v[j];
        }
        // event ring upkeep: land the chunk that was in flight, ask for the next one
        if (!use_z) {
            if (pf_pending) { ring_store(pf_at + lane, pf); filled = pf_at + 64; pf_pending = false; }
            if (filled < cur_start + 64 * CPL + 192 && filled < n_ev + 64 * CPL) {
                pf_at = filled; pf = ev_load(filled + lane); pf_pending = true;
            }
        }
        DP_PH(3);
        // wave argmax, first index among equal maxima (c_argmax, pyx:186-197): first lane that
        // holds the maximum, first of its cells that equals it
        // The wave maximum is first located in float32 (conversion is monotone, so the lanes
        // whose rounded maximum equals the rounded wave maximum include the true one; the DPP
        // ladder on 32-bit values is one instruction per step instead of three); one candidate
        // lane is the common case, several fall back to the float64 ladder.  The cell index
        // inside the winning lane comes from one compare + ballot per cell, resolved on the
        // scalar unit.
        double wm;
        {
            const float fl = (float)lmax;
            const float fm = wave_max_f32(fl);
            const u64 cm = __ballot(fl == fm);
            if (__popcll(cm) == 1) wm = readlane_f64(lmax, __ffsll((unsigned long long)cm) - 1);
            else wm = wave_max_f64(lmax);
        }
        const u64 eq = __ballot(lmax == wm) & has_cells; // (the AND on the scalar unit: a lane predicate in the
                                                         // ballot went through a 0/1 register and a second compare)
        const int wl = __ffsll((unsigned long long)eq) - 1; // first lane holding the maximum
        int wj = CPL - 1;
#pragma unroll
        for (int j = CPL - 2; j >= 0; j--) wj = ((__ballot(v[j] == wm) >> wl) & 1ull) ? j : wj;
        am = wl * CPL + wj;
        prev_start = cur_start;
        DP_PH(4);
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 5
        ph[6]++;
#endif
        return false;
    };
    {
        int row = row0;
        const int n_stat_rows = n_static < n_rows ? n_static : n_rows;
        for (; row < n_stat_rows; row++) if (row_step(row, BoolTag<false>{})) return;
        for (; row < n_rows; row++) if (row_step(row, BoolTag<true>{})) return;
    }
#if defined(TBA_PHASE_DEBUG) && TBA_PHASE_DEBUG == 5
    if (!DIRECT && mode == DP_MAIN && lane == 0) {
        for (int i = 0; i < 7; i++) r.dbg[i] = ph[i];
        r.dbg[7] = ph_first;
    }
#endif
#undef DP_PH
#ifdef TBA_SWEEP_STATS
    if (!DIRECT && mode == DP_MAIN && lane == 0) {
        r.dbg[0] = n_rows - row0; r.dbg[1] = sw_total;
    }
#endif
    // last row + traceback start (np.argmax of the last row, resquiggle.py:728,1032)
    if constexpr (DIRECT) {
        if (lane == 0) job->top_pos = am;
    } else {
        double *lr = last_row + (i64)blockIdx.x * TBA_MAX_BAND; // 64*CPL <= TBA_MAX_BAND
#pragma unroll
        for (int j = 0; j < CPL; j++) lr[b0 + j] = v[j];
        if (lane == 0) r.top_pos = am;
    }
}

#define TBA_DP_ARGS ReadState *rs, const DevParams *dp, int mode, \
    const double *event_means, const double *ref_means, const double *ref_sds, \
    i64 *band_starts, const i32 *lo_arr, const i32 *hi_arr, \
    unsigned char *moves, i64 start_moves_stride, double *last_row, DpJob *job
#define TBA_DP_PASS rs, dp, mode, event_means, ref_means, ref_sds, band_starts, lo_arr, hi_arr, \
    moves, start_moves_stride, last_row, job
template <int CPL, bool DIRECT>
__global__ __launch_bounds__(64) TBA_DP_WAVES_ATTR void k_dp(TBA_DP_ARGS)
{
    dp_body<CPL, DIRECT>(TBA_DP_PASS);
}
// The batch form of the 8-cell class (W = 500) under a register budget of 112 instead of 128
// (amdgpu_num_vgpr counts in units of two on gfx90a+: 56; it takes no template-dependent value,
// hence a kernel of its own).  Four wavefronts of the 128-register kernel fill a SIMD's register
// file; at 112 a wavefront of another stream's kernel (event detection, normalisation: <= 64
// registers) still fits beside them.  Costs 4 spilled registers (main_dp 59.9 -> 64.1 ms at cfg2)
// and wins when several engines stream on one device AND the DP is the smaller part of the work:
// RNA 3 kb reads in -> results out 95.6 k -> 107.6 k reads/s, DNA 10 kb 95.5 k -> 88.4 k
// (profiles/r04_k_dp_vgpr_budget_ab.txt) -- the engine picks per batch (tba_engine_set_sharing).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TBA_DP_WAVES), amdgpu_num_vgpr(56)))
void k_dp8_lowreg(TBA_DP_ARGS)
{
    dp_body<8, false>(TBA_DP_PASS);
}

// Static whole-read DP (find_static_base_assignment, resquiggle.py:547-600 over
// c_banded_forward_pass, pyx:240-279) for bands wider than the widest register class: a short
// read with a lot of signal gets a band of n_events - mask_len cells.  Rare and small in rows
// (< start_n_bases bases, or a failed start discovery), so: one wavefront per read, the two
// forward rows in a global scratch slice, 64 cells per step with the stay chain resolved by
// monotone sweeps (cell = max(non-stay candidate, (left - stay_pen) + z), iterated until no
// lane changes; the carry crosses steps through lane 63).  Same cell arithmetic, tie rules and
// packed 2-bit move rows as k_dp.  Blocks pick reads grid-strided; scratch = 2 * wide_w
// doubles per block.
__global__ __launch_bounds__(64) void k_dp_wide(ReadState *rs, i64 n_reads, const DevParams *dp,
    const double *event_means, const double *ref_means, const double *ref_sds,
    const i64 *band_starts, unsigned char *moves, double *scratch, i64 wide_w)
{
    const int lane = threadIdx.x;
    const tba_params &P = dp->p;
    const double stay_pen = P.stay_pen, skip_pen = P.skip_pen, z_shift = P.z_shift;
    const double zcap = P.do_winsorize_z ? P.max_half_z_score : INFINITY;
    const double NEG_INF = -INFINITY;
    double *rowA = scratch + (i64)blockIdx.x * 2 * wide_w, *rowB = rowA + wide_w;
    for (i64 ri = blockIdx.x; ri < n_reads; ri += gridDim.x) {
        ReadState &r = rs[ri];
        if (r.status != TBA_OK || r.path != PATH_STATIC || cpl_class(r.W) != 0) continue;
        const i64 W = r.W, n_rows = r.B;
        if (W > wide_w) { if (lane == 0) r.status = TBA_UNSUPPORTED; continue; }
        const double *ev = event_means + r.ev_off + r.clip;
        const double *rmu = ref_means + r.ref_off, *rsd = ref_sds + r.ref_off;
        const i64 *bst = band_starts + r.ref_off;
        unsigned char *mv = moves + r.moves_off;
        const i64 rowb = mv_row_bytes(W);
        double *prev = rowA, *cur = rowB;
        for (i64 b = lane; b < W; b += 64) prev[b] = 0.0; // row 0: zeros
        __threadfence();
        double best_v = NEG_INF; i64 best_i = 0;
        for (i64 row = 0; row < n_rows; row++) {
            const i64 st = bst[row];
            const i64 diff = row > 0 ? st - bst[row - 1] : 0;
            const double mu = rmu[row], sd = rsd[row];
            const bool last = row == n_rows - 1;
            unsigned char *mrow = mv + (row + 1) * rowb;
            double carry = NEG_INF; // value of the cell left of this step's first cell
            best_v = NEG_INF; best_i = 0;
            for (i64 b0 = 0; b0 < W; b0 += 64) {
                const i64 b = b0 + lane;
                const bool in = b < W;
                const i64 bc = in ? b : W - 1;
                double pz = fabs((ev[st + bc] - mu) / sd);
                pz = __builtin_fmin(pz, zcap);
                const double z = z_shift - pz;
                // non-stay candidate and its move: diag (2) wins ties against skip (1)
                const i64 pb = bc + diff;
                double v0; int m0;
                if (bc == 0) { // first cell, pyx:259-270
                    if (row == 0 || diff == 0) { v0 = prev[0] - skip_pen; m0 = 1; }
                    else { v0 = prev[diff - 1] + z; m0 = 2; }
                } else {
                    const double d = pb - 1 < W ? prev[pb - 1] + z : NEG_INF;
                    const double sk = pb < W ? prev[pb] - skip_pen : NEG_INF;
                    v0 = sk > d ? sk : d;
                    m0 = sk > d ? 1 : 2;
                }
                v0 = in ? v0 : NEG_INF;
                // stay chain: monotone sweeps to the fixed point
                double v = v0, sv = NEG_INF;
                for (;;) {
                    double left = __shfl_up(v, 1, 64);
                    left = lane == 0 ? carry : left;
                    sv = bc == 0 ? NEG_INF : (left - stay_pen) + z;
                    const double nv = (in && sv > v) ? sv : v;
                    const bool ch = nv != v;
                    v = nv;
                    if (!__any(ch)) break;
                }
                // a stay is replaced only by a strictly better candidate (pyx:213-234)
                const int m = (bc != 0 && !(v0 > sv)) ? 0 : m0;
                carry = __shfl(v, 63, 64);
                if (in) cur[b] = v;
                // pack four cells per byte
                int mm = in ? m : 0;
                const int m1 = __shfl_down(mm, 1, 64), m2 = __shfl_down(mm, 2, 64), m3 = __shfl_down(mm, 3, 64);
                if ((lane & 3) == 0 && in) mrow[b >> 2] = (unsigned char)(mm | (m1 << 2) | (m2 << 4) | (m3 << 6));
                if (last && in && v > best_v) { best_v = v; best_i = b; } // first max per lane
            }
            __threadfence(); // the next row reads this one across lanes
            double *t = prev; prev = cur; cur = t;
        }
        // np.argmax of the last row: first index of the maximum
        const double wm = wave_max_f64(best_v);
        i64 cand = best_v == wm ? best_i : (i64)0x7fffffffffffffffll;
        for (int o = 32; o >= 1; o >>= 1) { const i64 t = __shfl_xor(cand, o, 64); cand = t < cand ? t : cand; }
        if (lane == 0) r.top_pos = cand;
    }
}

// one 2-bit move code out of the packed rows written by k_dp
__device__ __forceinline__ int mv_get(const unsigned char *mv, i64 row, int cpl, int bpl, i64 b)
{
    (void)cpl;
    (void)bpl;
    return (mv[row * (i64)mv_class_rowb(cpl) + (b >> 2)] >> (2 * (b & 3))) & 3;
}

// c_banded_traceback (pyx:281-310); python wrap-around indexing of a negative band position
// kept.  packed != 0: 2-bit moves in k_dp's row layout (cpl = cells per lane); else one byte per
// cell with row stride `stride`.  Returns a TBA status.
__device__ inline int dev_banded_traceback(const unsigned char *mv, i64 stride, int cpl,
    i64 n_bases, i64 bw, const i64 *starts, bool identity, i64 band_pos, i64 thresh,
    i64 *seq_poss)
{
    const int bpl = cpl ? mv_bpl(cpl) : 0;
#define ST_AT(r_) (identity ? (i64)(r_) : starts[(r_)])
#define MV_AT(r_, b_) (cpl ? mv_get(mv, (r_), cpl, bpl, (b_) < 0 ? (b_) + bw : (b_)) \
                           : (int)mv[(r_) * stride + ((b_) < 0 ? (b_) + bw : (b_))])
    i64 cur_ev = band_pos + ST_AT(n_bases - 1);
    seq_poss[n_bases] = cur_ev + 1;
    for (i64 rr = n_bases; rr > 0; rr--) {
        const i64 st = ST_AT(rr - 1);
        band_pos = cur_ev - st;
        if (band_pos >= bw || band_pos < -bw) return TBA_INTERNAL;
        while (MV_AT(rr, band_pos) == 0) {
            band_pos--;
            if (band_pos < -bw) return TBA_INTERNAL;
        }
        if (MV_AT(rr, band_pos) == 2) band_pos--;
        if (thresh >= 0) {
            i64 a = band_pos, b2 = bw - band_pos - 1;
            if ((a < b2 ? a : b2) < thresh) return TBA_BEYOND_BANDWIDTH;
        }
        cur_ev = st + band_pos;
        seq_poss[rr - 1] = cur_ev + 1;
    }
#undef ST_AT
#undef MV_AT
    return TBA_OK;
}

// (k_start_tb, the start discovery's epilogue, lives in k_tb_par.h: it walks with that file's row blocks)

// the branch of find_adaptive_base_assignment before start discovery (resquiggle.py:984-989)
__global__ __launch_bounds__(64) void k_path0(ReadState *rs, i64 n_reads, const DevParams *dp)
{
    i64 ri = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (ri >= n_reads) return;
    ReadState &r = rs[ri];
    if (r.status != TBA_OK) return;
    const tba_params &P = dp->p;
    if (dp->kmer_width - dp->central_pos - 1 <= 0) { r.status = TBA_DISCORDANT; return; }
    if (r.n_ev < P.start_bw + P.start_n_bases || r.B < P.start_n_bases) r.start_state = ST_STATIC;
    else r.start_state = ST_TRY;
}

// Everything between start discovery and the forward pass: open-pore check, clip / offset,
// static fallback decision (resquiggle.py:1008-1027), the static-band row descriptors of
// _get_masked_start_fwd_pass (resquiggle.py:607-683) or find_static_base_assignment
// (resquiggle.py:561-571).  One thread per read; writes band_starts / lo / hi for the static
// rows and the moves size (in moves_off, turned into an offset by k_scan_moves).
__global__ __launch_bounds__(64) void k_prep(ReadState *rs, i64 n_reads, const DevParams *dp, i64 *band_starts,
    i32 *lo_arr, i32 *hi_arr)
{
    i64 ri = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (ri >= n_reads) return;
    ReadState &r = rs[ri];
    r.moves_off = 0;
    r.tb_done = 0;
    r.tb_verify_fail = 0;
    r.strip_s0 = -1;
    r.tb_form = TBA_TB_FORM_NONE;
    r.dp_wg = 0;
    if (r.status != TBA_OK) return;
    const tba_params &P = dp->p;
    i64 *bst = band_starts + r.ref_off;
    i32 *lo_a = lo_arr + r.ref_off, *hi_a = hi_arr + r.ref_off;
    bool use_static = r.start_state == ST_STATIC;
    const i64 bw = P.bandwidth, half_bw = P.bandwidth / 2;
    i64 clip = 0, offset = 0;
    if (!use_static) {
        if (r.start_state != ST_OK) { r.status = TBA_INTERNAL; return; }
        if (r.epb == 0) { r.status = TBA_OPEN_PORE; return; }
        if (r.mapped_start < half_bw) { clip = 0; offset = r.mapped_start; }
        else { clip = r.mapped_start - half_bw; offset = half_bw; }
        if ((i64)((double)(half_bw + 1) / r.epb) >= r.B || r.n_ev - offset - clip < bw)
            use_static = true;
    }
    if (use_static) {
        // find_static_base_assignment, resquiggle.py:561-571
        const i64 seq_len = r.B, n_ev = r.n_ev;
        const i64 mask_len = (seq_len < n_ev ? seq_len : n_ev) / 4;
        const i64 Ws = n_ev - mask_len;
        if (Ws <= 0 || seq_len - 2 * mask_len < 0) { r.status = TBA_INTERNAL; return; }
        const i64 nz = seq_len - 2 * mask_len;
        for (i64 i = 0; i < seq_len; i++) {
            i64 s = 0;
            if (i >= nz) s = (i64)np_linspace_at(0.0, (double)mask_len, 2 * mask_len, i - nz);
            bst[i] = s; lo_a[i] = 0; hi_a[i] = (i32)Ws;
        }
        r.path = PATH_STATIC; r.clip = 0; r.offset = 0; r.W = Ws; r.n_static = seq_len;
        r.moves_off = (seq_len + 1) * mv_row_bytes(Ws); // wider than every band class: k_dp_wide
        return;
    }
    // _get_masked_start_fwd_pass on event_means[clip:]
    const i64 n_ev = r.n_ev - clip;
    if (n_ev - offset < bw) { r.status = TBA_STARTS_TOO_FAR; return; }
    if (cpl_class(bw) == 0) { r.status = TBA_UNSUPPORTED; return; }
    const double epb = r.epb;
    const i64 start_pos = half_bw <= offset ? 0 : offset - half_bw;
    i64 tmp_seq_len = half_bw > MASK_BASES ? half_bw : MASK_BASES;
    const i64 t3 = (i64)((double)(half_bw + 1) / epb);
    if (t3 > tmp_seq_len) tmp_seq_len = t3;
    tmp_seq_len += 1;
    const double ls0 = (double)start_pos;
    const double ls1 = (double)start_pos + ((double)tmp_seq_len * epb);
    i64 first = -1;
    for (i64 i = 0; i < tmp_seq_len; i++)
        if ((i64)np_linspace_at(ls0, ls1, tmp_seq_len, i) >= offset) { first = i + 2; break; }
    if (first < 0) { r.status = TBA_INTERNAL; return; }
    i64 msl = first > MASK_BASES ? first : MASK_BASES;
    if (msl > tmp_seq_len) msl = tmp_seq_len;
    if (msl > r.B) { r.status = TBA_INTERNAL; return; }
    for (i64 i = 0; i < msl; i++) bst[i] = (i64)np_linspace_at(ls0, ls1, tmp_seq_len, i);
    const double m0 = (double)(offset + 1), m1 = (double)(bst[MASK_BASES - 1] + bw);
    for (i64 sp = 0; sp < msl; sp++) {
        const i64 ev_pos = bst[sp];
        const i64 sml = offset - ev_pos > 0 ? offset - ev_pos : 0;
        i64 eml = sp >= MASK_BASES ? 0
                                   : bw - ((i64)np_linspace_at(m0, m1, MASK_BASES, sp) - ev_pos);
        if (ev_pos + bw - eml > n_ev) eml = ev_pos + bw - n_ev;
        const i64 lo = ev_pos + sml, hi = ev_pos + bw - eml;
        i64 hi_c = hi > n_ev ? n_ev : hi;
        if (hi < 0) { hi_c = hi + n_ev; if (hi_c < 0) hi_c = 0; }
        const i64 lo_c = lo > n_ev ? n_ev : lo;
        const i64 nzs = hi_c > lo_c ? hi_c - lo_c : 0;
        const i64 eml_n = eml > 0 ? eml : 0;
        if (sml + nzs + eml_n != bw) { r.status = TBA_MASK_TOO_FEW; return; }
        lo_a[sp] = (i32)sml;
        hi_a[sp] = (i32)(sml + nzs);
    }
    r.path = PATH_ADAPTIVE; r.clip = clip; r.offset = offset; r.W = bw; r.n_static = msl;
    // (k_dp writes the centre strip; the reads k_dp_multi takes have none)
    r.strip_s0 = dp_multi_class(bw).cpl != 0 ? -1 : mv_strip_s0(bw);
    r.moves_off = (r.B + 1) * ((i64)mv_class_rowb(cpl_class(bw)) + (r.strip_s0 >= 0 ? MV_STRIP_BYTES : 0));
}

// exclusive scan of per-read arena sizes (held in `field`) into arena offsets: one workgroup,
// every thread scans a contiguous chunk of reads.  A read that does not fit the arena gets
// TBA_UNSUPPORTED (its space is still counted, so later reads may fail with it).
template <int WHICH> // 0: moves_off (also clears path on failure), 1: skip_off
__global__ __launch_bounds__(256) void k_scan_arena(ReadState *rs, i64 n_reads, i64 arena)
{
    __shared__ i64 s_sum[256];
    const int tid = threadIdx.x;
    const i64 chunk = (n_reads + 255) / 256;
    const i64 a = (i64)tid * chunk, b = a + chunk < n_reads ? a + chunk : n_reads;
    i64 acc = 0;
    for (i64 i = a; i < b; i++) {
        ReadState &r = rs[i];
        i64 sz = WHICH == 0 ? r.moves_off : r.skip_off;
        if (r.status != TBA_OK) sz = 0;
        acc += sz;
    }
    s_sum[tid] = acc;
    __syncthreads();
    if (tid == 0) { i64 run = 0; for (int t = 0; t < 256; t++) { i64 c = s_sum[t]; s_sum[t] = run; run += c; } }
    __syncthreads();
    acc = s_sum[tid];
    for (i64 i = a; i < b; i++) {
        ReadState &r = rs[i];
        i64 sz = WHICH == 0 ? r.moves_off : r.skip_off;
        if (r.status != TBA_OK) sz = 0;
        i64 off = acc;
        acc += sz;
        if (sz != 0 && off + sz > arena) {
            r.status = TBA_UNSUPPORTED;
            if (WHICH == 0) r.path = PATH_NONE;
            off = 0;
        }
        if (sz == 0) off = 0;
        if (WHICH == 0) r.moves_off = off; else r.skip_off = off;
    }
}

// main traceback (c_banded_traceback, pyx:281-310) + _trim_traceback (resquiggle.py:754-764).
// One LANE per read; launched with only TB_LANES reads per wavefront: the walk is a chain of
// dependent memory round trips, so what counts is the number of wavefronts in flight, not
// the lanes in use.  The pointer chase is made
// latency-tolerant by prefetching: the path stays near the same band position from row to row
// (the band follows it), so for the next TBR rows a 64-cell window of packed moves around the
// current band position (one 16-byte load per row) and the band starts are fetched together,
// then the rows are walked out of registers; a position outside its window falls back to
// direct loads.  A run of stays is resolved with clz on the 2-bit fields.
#ifndef TBR
#define TBR 16
#endif
__device__ __forceinline__ int mv_find_le(u64 x, int top) // highest non-zero 2-bit field <= top, or -1
{
    const u64 m = top >= 31 ? x : (x & ((1ull << (2 * top + 2)) - 1ull));
    const u64 nz = (m | (m >> 1)) & 0x5555555555555555ull;
    return nz ? (63 - __clzll((long long)nz)) >> 1 : -1;
}
__global__ __launch_bounds__(64) void k_main_tb(ReadState *rs, i64 n_reads, const DevParams *dp,
    const unsigned char *moves, const i64 *band_starts, i64 *read_tb)
{
    const i64 ri = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (ri >= n_reads) return;
    ReadState &r = rs[ri];
    if (r.status != TBA_OK) return;
    if (r.path == PATH_NONE) { r.status = TBA_INTERNAL; return; }
    if (r.tb_done) return;                              // walked chunk-parallel (k_tb_par.h)
    if (r.is_long && r.path == PATH_ADAPTIVE && r.W <= 1024) return; // k_main_tb_long (k_long.h)
    r.tb_form = TBA_TB_FORM_LANE;
    const i64 B = r.B;
    const int Wi = (int)r.W;
    const int rowb = (int)mv_row_bytes(r.W);            // bytes per packed row (multiple of 64)
    const int roww = rowb / 4;                          // dwords per row
    const unsigned char *mv = moves + r.moves_off;
    const i64 *st = band_starts + r.ref_off;
    i64 *tb = read_tb + r.seg_off;
    const bool adaptive = r.path == PATH_ADAPTIVE;
    const int thresh = adaptive ? (int)dp->p.band_bound_thresh : -1;
    i64 cur_ev = r.top_pos + st[B - 1];
    tb[B] = cur_ev + 1;
    int rc = TBA_OK;
    int bp_guess = (int)r.top_pos;
    for (i64 r0 = B; r0 >= 1 && rc == TBA_OK; r0 -= TBR) {
        // fetch: band starts and a 4-dword window of row r0-k around the expected position
        i64 stv[TBR];
        uint4 win[TBR];
        int wb = (bp_guess >> 4) - 2;                   // first dword of the window
        wb = wb < 0 ? 0 : (wb > roww - 4 ? roww - 4 : wb);
#pragma unroll
        for (int k = 0; k < TBR; k++) {
            const i64 rr = r0 - k;
            const i64 rc_ = rr >= 1 ? rr : 1;
            stv[k] = st[rc_ - 1];
            win[k] = *(const uint4 *)(mv + rc_ * rowb + 4 * wb);
        }
        // Per row, off the dependent chain (it only needs the fetched window): one bit per cell
        // for "not a stay" and "is a diagonal move" in the two 32-cell halves, and the answer for
        // "everything at or below the position in the upper half was a stay" (the highest
        // non-stay cell of the lower half).
        u64 nzl[TBR], nzh[TBR], d2l[TBR], d2h[TBR];
        int fl_full[TBR], m2_full[TBR];
#pragma unroll
        for (int k = 0; k < TBR; k++) {
            const u64 lo = ((u64)win[k].y << 32) | win[k].x, hi = ((u64)win[k].w << 32) | win[k].z;
            const u64 E = 0x5555555555555555ull;
            nzl[k] = (lo | (lo >> 1)) & E; nzh[k] = (hi | (hi >> 1)) & E;
            d2l[k] = (lo >> 1) & ~lo & E;  d2h[k] = (hi >> 1) & ~hi & E;
            const int cf = __clzll((long long)(nzl[k] | 1ull)); // (| 1: defined for an empty half)
            fl_full[k] = nzl[k] ? 31 - (cf >> 1) : -1;
            m2_full[k] = (int)((d2l[k] >> (2 * (fl_full[k] & 31))) & 1ull);
        }
#pragma unroll
        for (int k = 0; k < TBR; k++) {
            const i64 rr = r0 - k;
            const bool act = rr >= 1 && rc == TBA_OK;
            // Fast form, branch-free on the dependent chain (a wavefront of 16 lanes on its own
            // SIMD pays ~10 cycles per dependent instruction and much more per exec-mask
            // branch): the position lies in the 64-cell window; the highest non-stay move at or
            // below it is one shift + count-leading-zeros in its 32-cell half, else the
            // precomputed answer of the lower half.
            const i64 bp64 = cur_ev - stv[k];
            int bp = (int)bp64;
            const int lc = bp - 16 * wb;                // position inside the window
            const bool in_win = bp64 < Wi && bp64 >= 0 && (unsigned)lc < 64u;
            const bool up = (lc & 32) != 0;
            const int amt = 62 - 2 * (lc & 31);
            const u64 sn = (up ? nzh[k] : nzl[k]) << amt, s2 = (up ? d2h[k] : d2l[k]) << amt;
            const bool hit = sn != 0;
            const int c = __clzll((long long)(sn | 1ull)); // 1 + 2 (lc - f); 63 without a hit
            const bool low = !hit && up && fl_full[k] >= 0;
            const int m2_hit = (int)((s2 >> (63 - c)) & 1ull);
            const int f = hit ? lc - (c >> 1) : fl_full[k];
            const int m2 = hit ? m2_hit : m2_full[k];
            const bool fast = in_win && (hit || low);
            int m = m2 ? 2 : 1;
            if (fast) bp = 16 * wb + f;
            if (__builtin_expect(act && !fast, 0)) {
                // outside the window, or nothing but stays down to its start: the reference's
                // cell-by-cell walk with direct loads (python wrap-around of a negative index kept)
                if (bp64 >= Wi || bp64 < -Wi) { rc = TBA_INTERNAL; continue; }
                if (in_win) bp = wb > 0 ? 16 * wb - 1 : -1; // everything in the window was a stay
                const unsigned char *row = mv + rr * rowb;
#define MVG(b_) ({ int bb_ = (b_) < 0 ? (b_) + Wi : (b_); (int)((row[bb_ >> 2] >> (2 * (bb_ & 3))) & 3); })
                m = MVG(bp);
                while (m == 0) {
                    bp--;
                    if (bp < -Wi) { rc = TBA_INTERNAL; break; }
                    m = MVG(bp);
                }
#undef MVG
                if (rc != TBA_OK) continue;
            }
            if (m == 2) bp--;
            const int edge = bp < Wi - bp - 1 ? bp : Wi - bp - 1;
            const bool beyond = thresh >= 0 && edge < thresh;
            if (act) {
                if (beyond) rc = TBA_BEYOND_BANDWIDTH;
                else {
                    cur_ev = stv[k] + bp;
                    tb[rr - 1] = cur_ev + 1;
                    bp_guess = bp;
                }
            }
        }
    }
    if (rc != TBA_OK) { r.status = rc; return; }
    const i64 n_ev = r.n_ev - r.clip;
    if (adaptive) { // _trim_traceback
        i64 i = 0;
        while (tb[i] < 0) { tb[i] = 0; i++; if (i > B) { r.status = TBA_INTERNAL; return; } }
        i64 j = 1;
        while (tb[B + 1 - j] > n_ev) { tb[B + 1 - j] = n_ev; j++; if (j > B + 1) { r.status = TBA_INTERNAL; return; } }
    }
    i64 t0 = tb[0];
    if (!adaptive && (t0 < -(n_ev + 1) || t0 > n_ev)) { r.status = TBA_INTERNAL; return; }
    if (t0 < 0) t0 += n_ev + 1;
    r.top_pos = t0; // index of the first base's change point, consumed by k_tb_gather
}

// get_rel_raw_coords (resquiggle.py:858-864): segs[i] = valid_cpts[clip:][read_tb[i]] - first.
// grid: (blocks, reads); the last thread block of a read also records read_start / norm_len.
__global__ __launch_bounds__(256) void k_tb_gather(ReadState *rs, const i64 *valid_cpts,
    const i64 *read_tb, i64 *dp_segs)
{
    ReadState &r = rs[blockIdx.y];
    if (r.status != TBA_OK) return;
    const i64 B = r.B;
    const i64 n_ev = r.n_ev - r.clip, n_c = n_ev + 1;
    const i64 *c = valid_cpts + r.ev_off + r.clip;
    const i64 *tb = read_tb + r.seg_off;
    i64 *sg = dp_segs + r.seg_off;
    const i64 first = c[r.top_pos];
    const bool adaptive = r.path == PATH_ADAPTIVE;
    bool bad = false;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i <= B; i += (i64)gridDim.x * 256) {
        i64 t = tb[i];
        if (!adaptive && (t < -(n_ev + 1) || t > n_ev)) { bad = true; t = 0; }
        if (t < 0) t += n_c;
        const i64 v = c[t] - first;
        sg[i] = v;
        if (i == B) {
            r.dp_read_start = first;
            r.read_start = first;
            r.norm_len = v;
            if (first < 0 || v < 0 || first + v > r.n_raw) bad = true;
        }
    }
    if (bad) r.status = TBA_INTERNAL;
}
