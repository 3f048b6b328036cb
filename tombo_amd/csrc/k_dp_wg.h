// k_dp_wg.h -- the start-discovery RETRY (find_seq_start_in_events with the "save" band,
// resquiggle.py:992-1006 over c_banded_forward_pass, pyx:240-279) with one WORKGROUP per read.
//
// The retry band is 2 500 (DNA) / 3 000 (RNA) cells wide and only the few reads whose first try
// failed run it.  As a single wavefront (k_dp<48>: 48 cells per lane, 512 VGPRs and spills) one such
// read costs the whole batch 1.55 ms of latency -- 6 % of a 2 kb / W = 100 step, and most of a
// batch-of-one resquiggle_read that needs it.  Here the band is spread over the 256 lanes of a
// workgroup (4 waves on the 4 SIMDs of a CU, CPL <= 12 cells per lane): the rows are static
// (band start = row index, offset 1), so everything a row needs is staged in LDS once -- the
// n_rows + W event means, the levels of all rows -- and the previous row lives in LDS as in
// k_dp_multi.  The stay chain is resolved by the same exact fixed-point iteration as in k_dp; the
// exit values cross lanes by DPP inside a wavefront and through LDS between wavefronts (two
// barriers per sweep).  Moves go out in the packed layout of the band's k_dp class, so k_start_tb
// does not know which kernel wrote them.
#pragma once
#include "k_dp.h"

#define WG_MAX_ROWS 256
__host__ __device__ inline int dp_wg_cpl(i64 W) { return W <= 1024 ? 4 : W <= 2048 ? 8 : W <= 3072 ? 12 : 0; }

template <int CPL>
__global__ __launch_bounds__(256) void k_dp_wg(ReadState *rs, const DevParams *dp,
    const double *event_means, const double *ref_means, const double *ref_sds,
    unsigned char *moves, i64 start_moves_stride, double *last_row)
{
    constexpr int NT = 256, CELLS = NT * CPL;
    constexpr int PROW = 8 + CELLS + 8 + CPL + 1;
    __shared__ double E[WG_MAX_ROWS + CELLS];   // event means under all rows of the band
    __shared__ double LV[3 * WG_MAX_ROWS];      // level, sd, 1 / sd of every row
    __shared__ double PR[PROW];                 // previous row: 8 x -inf, cells, -inf pad
    __shared__ double xch[4];                   // exit value of each wave's last lane
    __shared__ int s_changed[2];
    ReadState &r = rs[blockIdx.x];
    if (r.status != TBA_OK || r.start_state != ST_RETRY) return;
    const tba_params &P = dp->p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Wi = (int)P.start_save_bw, n_rows = (int)P.start_n_bases, n_ev = (int)r.n_ev;
    const double *ev = event_means + r.ev_off;
    const double *rmu = ref_means + r.ref_off, *rsd = ref_sds + r.ref_off;
    unsigned char *mv = moves + (i64)blockIdx.x * start_moves_stride;
    const i64 mv_stride = mv_row_bytes(Wi);
    const double stay_pen = P.stay_pen, skip_pen = P.skip_pen, z_shift = P.z_shift;
    const double zcap = P.do_winsorize_z ? P.max_half_z_score : INFINITY;
    const double NEG_INF = -INFINITY;
    const int b0 = tid * CPL;
    int nvalid = Wi - b0;
    nvalid = nvalid < 0 ? 0 : (nvalid > CPL ? CPL : nvalid);
    // stage: events [0, n_rows + W) (zeros past the last event, as k_dp's clamped load), levels,
    // row 0 of the forward pass (zeros inside the band, pyx:253-254)
    for (int i = tid; i < n_rows + CELLS; i += NT) E[i] = i < n_ev ? ev[i] : 0.0;
    for (int i = tid; i < n_rows; i += NT) {
        const double sd = rsd[i];
        LV[3 * i] = rmu[i]; LV[3 * i + 1] = sd; LV[3 * i + 2] = 1.0 / sd;
    }
    for (int k = tid; k < PROW; k += NT) PR[k] = (k >= 8 && k < 8 + Wi) ? 0.0 : NEG_INF;
    if (tid < 2) s_changed[tid] = 0;
    __syncthreads();
    double v[CPL];
    for (int row = 0; row < n_rows; row++) {
        const double mu = LV[3 * row], sd = LV[3 * row + 1], y = LV[3 * row + 2];
        const int diff = row > 0 ? 1 : 0; // identity band starts
        // shifted half z-scores (resquiggle.py:712-720)
        double z[CPL];
        {
            const double *er = E + row + b0;
#pragma unroll
            for (int j = 0; j < CPL; j++) {
                double pz = fabs(div_by_recip(er[j] - mu, sd, y));
                pz = __builtin_fmin(pz, zcap);
                z[j] = j < nvalid ? z_shift - pz : NEG_INF;
            }
        }
        // diag / skip candidates from the previous row (pyx:259-270 first cell, 213-234 the rest)
        double cv[CPL];
        u32 tk = 0;
        {
            const double *pa = PR + 8 + b0 + diff - 1;
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
                    take_s = tid == 0 ? fs : take_s;
                    cv[j] = take_s ? sk : d;
                } else {
                    cv[j] = max_f64_raw(sk, d);
                }
                tk |= (take_s ? 1u : 2u) << (2 * j);
            }
        }
        // stay chain: the fixed-point iteration of k_dp over 256 lanes.  Exit values move one lane
        // up by DPP; lane 0 of a wave takes the exit of the previous wave's lane 63 from LDS.
        double exit0;
        {
            double x = NEG_INF;
#pragma unroll
            for (int j = 0; j < CPL; j++) x = max_f64_raw(cv[j], (x - stay_pen) + z[j]);
            exit0 = x;
        }
        double in = NEG_INF, ex = exit0;
        bool ok = false;
        for (int it = 0; it < NT + 2; it++) {
            if (lane == 63) xch[wave] = ex;
            __syncthreads();
            double nin = wave_shr1_f64(ex, NEG_INF);
            if (lane == 0 && wave > 0) nin = xch[wave - 1];
            const bool ch = __ballot(nin != in) != 0;
            if (ch && lane == 0) s_changed[it & 1] = 1;
            if (tid == 0) s_changed[(it + 1) & 1] = 0; // (next iteration's flag; nobody reads it before the barrier after next)
            __syncthreads();
            if (s_changed[it & 1] == 0) { ok = true; break; }
            in = nin;
            double c = in;
#pragma unroll
            for (int j = 0; j < CPL; j++) c = (c - stay_pen) + z[j];
            ex = max_f64_raw(exit0, c);
        }
        if (!ok) { if (tid == 0) r.status = TBA_INTERNAL; return; } // NaNs in the signal
        // the cells, their move codes, the new previous row
        u32 keep = 0;
        {
            double x = in;
#pragma unroll
            for (int j = 0; j < CPL; j++) {
                const double s = (x - stay_pen) + z[j];
                keep |= cv[j] > s ? (3u << (2 * j)) : 0u;
                x = max_f64_raw(cv[j], s);
                v[j] = x;
            }
        }
        const u32 mvw = tk & keep;
        {
            unsigned char *mrow = mv + (row + 1) * mv_stride + tid * (CPL / 4);
            if (tid * (CPL / 4) < mv_stride) { // (a narrow band's row is shorter than 256 lanes' bytes)
#pragma unroll
                for (int q = 0; q < CPL / 4; q++) mrow[q] = (unsigned char)(mvw >> (8 * q));
            }
        }
        // (every lane has read its candidates before the first barrier of the sweep loop)
#pragma unroll
        for (int j = 0; j < CPL; j++) if (j < nvalid) PR[8 + b0 + j] = v[j];
        __syncthreads();
    }
    // last row + traceback start: np.argmax of the last row (first index of the maximum)
    double *lr = last_row + (i64)blockIdx.x * TBA_MAX_BAND;
    double lmax = NEG_INF;
    int lidx = 0;
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        if (b0 + j < TBA_MAX_BAND) lr[b0 + j] = v[j];
        if (j < nvalid && v[j] > lmax) { lmax = v[j]; lidx = b0 + j; }
    }
    __shared__ double s_max[4];
    __shared__ int s_idx[4];
    double wm = lmax;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const double t = shfl_xor_f64(wm, o); wm = t > wm ? t : wm; }
    const u64 eq = __ballot(lmax == wm && nvalid > 0);
    const int wl = eq ? __builtin_ctzll(eq) : 0;
    const int widx = __shfl(lidx, wl, 64);
    if (lane == 0) { s_max[wave] = eq ? wm : NEG_INF; s_idx[wave] = widx; }
    __syncthreads();
    if (tid == 0) {
        int best = 0;
        for (int w = 1; w < 4; w++) if (s_max[w] > s_max[best]) best = w; // first wave holding the maximum
        r.top_pos = s_idx[best];
    }
}
