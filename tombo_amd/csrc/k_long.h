// k_long.h -- the serial-per-read stages for LONG reads (ReadState.is_long: more than
// TBA_LONG_RAW samples or TBA_LONG_BASES bases; set on the host, plan_batch).
//
// k_cumsum_scores and k_main_tb give a read one LANE (the cumulative sum must be accumulated
// left to right, the traceback is a pointer chase) and hide the latency behind the other reads of
// the wavefront / workgroup.  That is right for a batch of like reads and wrong for the tail of a
// real run: a 200 kb read (1.8 M samples) holds its workgroup for 14 000 pipeline steps of the
// scan (65 ms) and its wavefront for 200 000 rows of a one-lane traceback (75 ms), on the critical
// path of its batch, while the machine idles.  Here a long read gets a workgroup (scan) or a
// wavefront (traceback) of its own and the serial part is cut to what is inherently serial: one
// dependent float64 add per sample, one short dependent chain per row.
#pragma once
#include "tba_common.h"
#include "k_dp.h"

#define TBA_LONG_RAW 262144  // samples
#define TBA_LONG_BASES 24576 // bases

// ---- np.cumsum + change-point scores (c_valid_cpts_w_cap, _c_helper.pyx:94-98), one workgroup
// per read.  Tiles of CL_T samples rotate through four LDS buffers: while lane 0 of wave 0 adds
// up tile i (in place, left to right: the bits of np.cumsum), waves 1..3 turn tile i-1 into scores
// (MODE 0) or store its sums (MODE 1: the stall detector's cumulative sum, k_prep_raw.h), and
// drop tile i+2 -- fetched during the previous step -- into the buffer tile i-2 has left.  One
// barrier per step; a step lasts as long as the 1856 dependent adds (~8 us), so the loads and
// stores are free.  grid: one block per entry of `long_idx`.
#define CL_T 1856 // (4 buffers of 64 + 1856 doubles = 61 KB: the static LDS limit is 64 KB)
#define CL_H 64 // halo: the 2w sums before a tile (2 * running_stat_width <= 64, else k_cumsum path)
template <class RT, int MODE>
__global__ __launch_bounds__(256) void k_cumsum_scores_long(const ReadState *rs, const i32 *long_idx,
    const DevParams *dp, const RT *__restrict__ sig, double *__restrict__ out)
{
    __shared__ double buf[4][CL_H + CL_T];
    const i64 ri = long_idx[blockIdx.x];
    const ReadState &r = rs[ri];
    if (r.status != TBA_OK) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const i64 n = r.n_raw;
    const RT *x = sig + r.raw_off;
    const int w = (int)dp->p.running_stat_width, w2 = 2 * w;
    const i64 n_tiles = (n + CL_T - 1) / CL_T;
    double *o = MODE == 1 ? out + r.raw_off + ri : out + r.raw_off; // csum has one more entry per read
    if (MODE == 1 && tid == 0) o[0] = 0.0;
    // loaders: threads 64..255 (192 of them) move CL_T samples per step: 10 per thread, strided
    constexpr int NLD = 192, PER = (CL_T + NLD - 1) / NLD;
    double pre[PER];
    const int lt = tid - 64;
    auto fetch = [&](i64 tile) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int c = lt + u * NLD;
            const i64 k = tile * CL_T + c;
            pre[u] = (c < CL_T && k < n) ? (double)x[k] : 0.0;
        }
    };
    auto drop = [&](double *b) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int c = lt + u * NLD;
            if (c < CL_T) b[CL_H + c] = pre[u];
        }
    };
    for (int k = tid; k < CL_H; k += 256) buf[0][k] = 0.0; // c[<= 0] = 0
    if (wave > 0) { // tiles 0 and 1 straight into their buffers, tile 2 in flight
        fetch(0); drop(buf[0]);
        if (n_tiles > 1) { fetch(1); drop(buf[1]); }
        if (n_tiles > 2) fetch(2);
    }
    __syncthreads();
    double acc = 0.0;
    for (i64 i = 0; i <= n_tiles; i++) { // one extra step for the outputs of the last tile
        if (wave == 0) {
            if (i < n_tiles && lane == 0) {
                double *t = buf[i & 3] + CL_H;
                const i64 left = n - i * CL_T;
                const int m = left >= CL_T ? CL_T : (int)left;
                int k = 0;
                for (; k + 16 <= m; k += 16) { // 16 loads, 16 dependent adds, 16 stores
                    double v[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) v[u] = t[k + u];
#pragma unroll
                    for (int u = 0; u < 16; u++) { acc = acc + v[u]; v[u] = acc; }
#pragma unroll
                    for (int u = 0; u < 16; u++) t[k + u] = v[u];
                }
                for (; k < m; k++) { acc = acc + t[k]; t[k] = acc; }
                // the halo of the next tile: this tile's last CL_H sums
                double *h = buf[(i + 1) & 3];
                for (int u = 0; u < CL_H; u++) h[u] = t[CL_T - CL_H + u];
            }
        } else {
            if (i >= 1) {
                // tile j = i - 1 holds c[jT + 1 .. jT + T] (column t <-> c index jT + 1 + t); its
                // halo the CL_H sums before it
                const double *t = buf[(i - 1) & 3] + CL_H;
                const i64 m0 = (i - 1) * CL_T + 1;
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const int c = lt + u * NLD;
                    if (c >= CL_T) continue;
                    if (MODE == 1) {
                        const i64 ci = m0 + c; // c index
                        if (ci <= n) o[ci] = t[c];
                    } else {
                        // column c closes the window of position k = c index - 2w (pyx:94-98)
                        const i64 k = m0 + c - w2;
                        if (k >= 0 && k < n + 1 - w2)
                            o[k] = fabs(((2 * t[c - w]) - t[c - w2]) - t[c]);
                    }
                }
            }
            // tile i + 2 (in registers since the previous step) takes the buffer of tile i - 2,
            // whose outputs went out in step i - 1; then tile i + 3 is asked for
            if (i + 2 < n_tiles) drop(buf[(i + 2) & 3]);
            if (i + 3 < n_tiles) fetch(i + 3);
        }
        __syncthreads();
    }
}

// ---- c_banded_traceback (pyx:281-310) + _trim_traceback (resquiggle.py:754-764) of a long read on
// the adaptive path, one WAVEFRONT per read.  k_main_tb walks a read with one lane (~95 dependent
// instructions per row); here the 64 lanes hold the packed moves of a whole row (one dword = 16
// cells per lane: bands up to 1024 cells) and the walk itself runs on the SCALAR unit: the two
// dwords around the position are read into scalar registers (v_readlane with a scalar lane index),
// "the highest non-stay cell at or below the position" is a mask and a find-first-bit there (a
// dependent scalar operation costs a few cycles, a vector one in a lone wavefront ~8 and a
// cross-lane one more); stay runs longer than the 16-31 cells below the position fall back to a
// masked compare + ballot over the whole row.  The rows of a block of 16 are fetched together (16
// coalesced row loads in flight, the band starts in one).  grid: one block of 64 per entry of `long_idx`; reads it does not take
// (status, path, band wider than 1024) are left to k_main_tb.
#define TBL_R 16
__device__ __forceinline__ bool tb_long_takes(const ReadState &r)
{
    return r.is_long && r.path == PATH_ADAPTIVE && r.W <= 1024;
}
__global__ __launch_bounds__(64) void k_main_tb_long(ReadState *rs, const i32 *long_idx, const DevParams *dp,
    const unsigned char *moves, const i64 *band_starts, i64 *read_tb)
{
    const i64 ri = long_idx[blockIdx.x];
    ReadState &r = rs[ri];
    if (r.status != TBA_OK || !tb_long_takes(r) || r.tb_done) return;
    const int lane = threadIdx.x;
    if (lane == 0) r.tb_form = TBA_TB_FORM_LONG;
    const int B = (int)uni(r.B), Wi = (int)uni(r.W);
    const int rowb = (int)mv_row_bytes(Wi), roww = rowb / 4;
    const unsigned char *mv = uni(moves + r.moves_off);
    const i64 *st = uni(band_starts + r.ref_off);
    i64 *tb = uni(read_tb + r.seg_off);
    const int thresh = (int)uni(dp->p.band_bound_thresh);
    const int n_ev = (int)uni(r.n_ev - r.clip);
    int cur_ev = (int)uni(r.top_pos) + (int)uni(st[B - 1]);
    int rc = TBA_OK;
    bool back_active = true; // _trim_traceback from the back: values above n_ev are clamped until the
                             // first one that is not
    int first_nonneg = B + 1; // smallest index whose value is >= 0 (the front rule clamps below it)
    int v_top = cur_ev + 1;
    if (v_top > n_ev) v_top = n_ev; else back_active = false;
    if (cur_ev + 1 >= 0) first_nonneg = B;
    if (lane == 0) tb[B] = v_top;
    int last_val = v_top;
    // a block: my dword of each of its 16 rows and the band starts (lane k: row r0 - k); the next
    // block is fetched while this one is walked
    u32 d[TBL_R], dn[TBL_R];
    int st_v, st_n;
    auto fetch = [&](int r0, u32 (&dd)[TBL_R], int &sv) {
#pragma unroll
        for (int k = 0; k < TBL_R; k++) {
            const int rr = r0 - k >= 1 ? r0 - k : 1;
            dd[k] = lane < roww ? *(const u32 *)(mv + (i64)rr * rowb + 4 * lane) : 0u;
        }
        const int rs_k = r0 - lane >= 1 ? r0 - lane : 1;
        sv = lane < TBL_R ? (int)st[rs_k - 1] : 0;
    };
    fetch(B, dn, st_n);
    for (int r0 = B; r0 >= 1 && rc == TBA_OK; r0 -= TBL_R) {
#pragma unroll
        for (int k = 0; k < TBL_R; k++) d[k] = dn[k];
        st_v = st_n;
        if (r0 - TBL_R >= 1) fetch(r0 - TBL_R, dn, st_n);
        int res = 0; // lane k: the value of tb[r0 - k - 1]
#pragma unroll
        for (int k = 0; k < TBL_R; k++) {
            const int rr = r0 - k;
            if (rr < 1 || rc != TBA_OK) continue;
            const int stv = __builtin_amdgcn_readlane(st_v, k);
            int bp = cur_ev - stv;
            int m;
            bool found = false;
            if (__builtin_expect(bp >= 0 && bp < Wi, 1)) {
                // scalar fast path: the dword holding the position and the one below it (32 cells)
                // are read into scalar registers; "highest non-stay cell at or below" is a mask and
                // a find-first-bit on the scalar unit, whose dependent operations cost a few cycles
                const int q = bp >> 4;
                const u32 hi = (u32)__builtin_amdgcn_readlane((int)d[k], q);
                const u32 lo = q > 0 ? (u32)__builtin_amdgcn_readlane((int)d[k], q - 1) : 0u;
                const u64 win = ((u64)hi << 32) | lo; // cells [16 (q - 1), 16 (q + 1))
                u64 nz = (win | (win >> 1)) & 0x5555555555555555ull;
                const int top = 34 + 2 * (bp & 15); // bits of the fields at or below the position
                nz = top >= 64 ? nz : (nz & ((1ull << top) - 1ull));
                if (__builtin_expect(nz != 0, 1)) {
                    const int f = (63 - __builtin_clzll(nz)) >> 1;
                    bp = 16 * (q - 1) + f;
                    m = (int)((win >> (2 * f)) & 3ull);
                    found = true;
                }
            }
            if (found) {
            } else if (__builtin_expect(bp >= 0 && bp < Wi, 1)) { // a stay run of more than 16-31 cells: the whole row
                const int q = bp >> 4, s2 = 2 * (bp & 15) + 2;
                const u32 e = (d[k] | (d[k] >> 1)) & 0x55555555u;
                const u32 em = lane < q ? e : (lane == q ? (s2 >= 32 ? e : (e & ((1u << s2) - 1u))) : 0u);
                const u64 bal = __ballot(em != 0);
                if (__builtin_expect(bal != 0, 1)) {
                    const int hl = 63 - __builtin_clzll(bal);
                    const u32 eh = (u32)__builtin_amdgcn_readlane((int)em, hl);
                    const u32 dh = (u32)__builtin_amdgcn_readlane((int)d[k], hl);
                    const int f = (31 - __builtin_clz(eh)) >> 1;
                    bp = hl * 16 + f;
                    m = (int)((dh >> (2 * f)) & 3u);
                } else {
                    m = -1; // nothing but stays down to cell 0: the reference walks on (wrap-around)
                }
            } else {
                m = -1;
            }
            if (__builtin_expect(m < 0, 0)) {
                // the reference's cell-by-cell walk with python's wrap-around of a negative index
                if (bp >= Wi || bp < -Wi) { rc = TBA_INTERNAL; continue; }
                if (bp >= 0) bp = -1; // (everything from the position down to cell 0 was a stay)
                const unsigned char *row = mv + (i64)rr * rowb;
                int mm = 0;
                for (;;) {
                    const int bb = bp < 0 ? bp + Wi : bp;
                    mm = uni((int)((row[bb >> 2] >> (2 * (bb & 3))) & 3)); // (every lane reads the same byte)
                    if (mm != 0) break;
                    bp--;
                    if (bp < -Wi) { rc = TBA_INTERNAL; break; }
                }
                // (every lane walked the same cells: tell the compiler, or the whole loop state is
                // kept in vector registers under exec masks)
                rc = uni(rc); bp = uni(bp); mm = uni(mm);
                if (rc != TBA_OK) continue;
                m = mm;
            }
            if (m == 2) bp--;
            const int edge = bp < Wi - bp - 1 ? bp : Wi - bp - 1;
            if (thresh >= 0 && edge < thresh) { rc = TBA_BEYOND_BANDWIDTH; continue; }
            cur_ev = stv + bp;
            int val = cur_ev + 1;
            if (back_active) { if (val > n_ev) val = n_ev; else back_active = false; }
            if (cur_ev + 1 >= 0) first_nonneg = rr - 1;
            last_val = val;
            res = lane == k ? val : res;
        }
        if (rc == TBA_OK) {
            const int idx = r0 - lane - 1;
            if (lane < TBL_R && idx >= 0) tb[idx] = res;
        }
    }
    if (rc != TBA_OK) { if (lane == 0) r.status = rc; return; }
    // front rule of _trim_traceback: indices below the first non-negative value become 0
    if (first_nonneg > B) { if (lane == 0) r.status = TBA_INTERNAL; return; }
    for (int i = lane; i < first_nonneg; i += 64) tb[i] = 0;
    int t0 = first_nonneg > 0 ? 0 : last_val;
    if (lane == 0) r.top_pos = t0; // index of the first base's change point (k_tb_gather)
}
