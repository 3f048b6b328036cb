// k_cabi.h -- small kernels behind the per-kernel C ABI entry points (tba_c_*): each is the
// device form of one Cython function, run as a batch of one.
#pragma once
#include "tba_common.h"
#include "k_dp.h"

// c_base_z_scores, _c_dynamic_programming.pyx:17-32
__global__ void k_c_base_z_scores(const double *sig, i64 n, double mean, double sd, int winsor,
                                  double mh, double *out)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        double z = (sig[i] - mean) / sd;
        if (z > 0) z = -z;
        if (winsor && z < -mh) z = -mh;
        out[i] = z;
    }
}

// (c_new_means, _c_helper.pyx:59-71: tba_c_new_means runs the batch pipeline's k_event_means, k_segment.h)

// c_apply_outlier_thresh, _c_helper.pyx:73-87
__global__ void k_c_clip(const double *sig, i64 n, double lo, double hi, double *out)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        double v = sig[i];
        out[i] = v > hi ? hi : (v < lo ? lo : v);
    }
}

// c_banded_traceback, pyx:281-310 (one thread)
__global__ void k_c_traceback(const unsigned char *mv, i64 stride, i64 n_bases, i64 bw,
                              const i64 *starts, i64 band_pos, i64 thresh, i64 *seq_poss,
                              i32 *status)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *status = dev_banded_traceback(mv, stride, 0, n_bases, bw, starts, false, band_pos, thresh,
                                       seq_poss);
}

// self-test of the row-constant division used by k_dp (div_by_recip vs the IEEE quotient)
__global__ void k_c_div_check(const double *a, const double *b, i64 n, double *out)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const double y = 1.0 / b[i];
        const double q5 = div_by_recip(a[i], b[i], y), q4 = div_by_recip2(a[i], b[i], y, recip_low(b[i], y));
        // (both forms: a disagreement between them comes out as a NaN, which equals no quotient)
        out[i] = q5 == q4 || (q5 != q5 && q4 != q4) ? q4 : __longlong_as_double(0x7ff8000000000bad);
    }
}

// self-test of the approximate quotient k_theil_sen classifies pairs with: a * r, r =
// v_rcp_f64(b) refined by one Newton step
__global__ void k_c_rcp_check(const double *a, const double *b, i64 n, double *out)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        double rr = __builtin_amdgcn_rcp(b[i]);
        rr = __builtin_fma(__builtin_fma(-b[i], rr, 1.0), rr, rr);
        out[i] = a[i] * rr;
    }
}

// c_new_mean_stds, _c_helper.pyx:38-57 (population sd around the segment mean)
__global__ void k_c_new_mean_stds(const double *sig, const i64 *segs, i64 n_segs, double *means,
                                  double *stds)
{
    for (i64 s = (i64)blockIdx.x * blockDim.x + threadIdx.x; s < n_segs; s += (i64)gridDim.x * blockDim.x) {
        const double len = (double)(segs[s + 1] - segs[s]);
        double acc = 0;
        for (i64 j = segs[s]; j < segs[s + 1]; j++) acc += sig[j];
        const double m = acc / len;
        means[s] = m;
        double v = 0;
        for (i64 j = segs[s]; j < segs[s + 1]; j++) { const double d = sig[j] - m; v += d * d; }
        stds[s] = sqrt(v / len);
    }
}

// c_compute_slopes, _c_helper.pyx:362-377: itertools.combinations order, row i of the upper
// triangle starts at i*(2n-i-1)/2.  grid.x = n rows.
__global__ void k_c_compute_slopes(const double *ev, const double *md, i64 n, double max_slope,
                                   double *slopes)
{
    const i64 i = blockIdx.x;
    const double ei = ev[i], mi = md[i];
    double *row = slopes + i * (2 * n - i - 1) / 2 - (i + 1);
    for (i64 j = i + 1 + threadIdx.x; j < n; j += blockDim.x) {
        const double ej = ev[j];
        row[j] = ei == ej ? max_slope : (mi - md[j]) / (ei - ej);
    }
}

// c_reg_z_scores, _c_dynamic_programming.pyx:34-97, bounds part (a serial clip chain in both
// directions): absolute [start, end) of each base of the region and the z-score offsets.
__global__ void k_c_reg_bounds(const i64 *r_b_starts, i64 reg_start, i64 reg_end,
    i64 max_base_shift, i64 m, i64 *sig_starts, i64 *sig_ends, i64 *z_off)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const i64 reg_len = reg_end - reg_start;
    i64 prev = 0;
    for (i64 idx = 0; idx < reg_len; idx++) {
        const i64 base_i = reg_start + idx;
        const i64 lo = base_i - max_base_shift;
        i64 b = r_b_starts[lo > reg_start ? lo : reg_start];
        if (idx > 0 && b < prev + m) b = prev + m;
        sig_starts[idx] = b;
        prev = b;
    }
    for (i64 idx = 0; idx < reg_len; idx++) {
        const i64 base_i = reg_start + (reg_len - idx - 1);
        const i64 hi = base_i + max_base_shift + 1;
        i64 b = r_b_starts[hi < reg_end ? hi : reg_end];
        if (idx > 0 && b > prev - m) b = prev - m;
        sig_ends[reg_len - idx - 1] = b;
        prev = b;
    }
    i64 acc = 0;
    for (i64 idx = 0; idx < reg_len; idx++) {
        z_off[idx] = acc;
        const i64 l = sig_ends[idx] - sig_starts[idx];
        acc += l > 0 ? l : 0;
    }
    z_off[reg_len] = acc;
}

// z-score part: block b = base reg_start + b over r_sig[sig_starts[b]:sig_ends[b]]
__global__ void k_c_reg_z(const double *r_sig, i64 n_sig, const double *means, const double *sds,
    i64 reg_start, const i64 *sig_starts, const i64 *sig_ends, const i64 *z_off, i64 z_cap,
    int winsor, double mh, double *z, i32 *status)
{
    const i64 b = blockIdx.x;
    const i64 s = sig_starts[b], e = sig_ends[b];
    if (e <= s) return;
    if (s < 0 || e > n_sig || z_off[b] + (e - s) > z_cap) { if (threadIdx.x == 0) *status = TBA_INTERNAL; return; }
    const double mu = means[reg_start + b], sd = sds[reg_start + b];
    double *out = z + z_off[b];
    for (i64 k = threadIdx.x; k < e - s; k += blockDim.x) {
        double v = (r_sig[s + k] - mu) / sd;
        if (v > 0) v = -v;
        if (winsor && v < -mh) v = -mh;
        out[k] = v;
    }
}

// c_base_forward_pass, _c_dynamic_programming.pyx:99-163 (one serial chain; one thread).
// Negative indices wrap like the Cython buffer accesses, other out-of-range accesses are the
// reference's IndexError (TBA_INTERNAL).
__global__ void k_c_base_forward_pass(const double *b_data, i64 b_start, i64 b_end,
    const double *prev_b_data, i64 prev_b_start, i64 prev_b_end, const double *prev_fwd,
    const i64 *prev_last_diag, i64 m, double *cum, double *b_fwd, i64 *b_last_diag, i32 *status)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const i64 b_len = b_end - b_start, plen = prev_b_end - prev_b_start;
    { double acc = 0; for (i64 k = 0; k < plen; k++) { acc = k == 0 ? prev_b_data[k] : acc + prev_b_data[k]; cum[k] = acc; } }
    i64 i0 = b_start - prev_b_start - 1;
    if (i0 < 0) i0 += plen;
    if (i0 < 0 || i0 >= plen) { *status = TBA_INTERNAL; return; }
    b_fwd[0] = b_data[0] + prev_fwd[i0];
    b_last_diag[0] = 1;
    for (i64 pos = b_start + 1; pos < prev_b_end + 1; pos++) {
        if (pos - b_start >= b_len) { *status = TBA_INTERNAL; return; }
        i64 lag = 1;
        for (;;) {
            i64 idx = pos - prev_b_start - lag;
            if (idx < 0) idx += plen;
            if (idx < 0 || idx >= plen) { *status = TBA_INTERNAL; return; }
            if (prev_last_diag[idx] + lag <= m) lag++;
            else break;
        }
        i64 di = pos - prev_b_start - lag;
        if (di < 0) di += plen;
        double diag = prev_fwd[di];
        if (lag > 1) diag += cum[pos - prev_b_start - 1] - cum[di];
        const double stay = b_fwd[pos - b_start - 1];
        double best;
        i64 dv;
        if (diag > stay) { best = diag; dv = 1; }
        else { best = stay; dv = b_last_diag[pos - b_start - 1] + 1; }
        b_fwd[pos - b_start] = b_data[pos - b_start] + best;
        b_last_diag[pos - b_start] = dv;
    }
    if (b_end > prev_b_end + 1) {
        i64 at = prev_b_end - b_start;
        if (at < 0) at += b_len;
        if (at < 0 || at >= b_len) { *status = TBA_INTERNAL; return; }
        double fv = b_fwd[at];
        i64 cl = b_last_diag[at];
        for (i64 k = 0; k < b_end - prev_b_end - 1; k++) {
            fv += b_data[k + prev_b_end - b_start + 1];
            cl += 1;
            b_fwd[k + prev_b_end - b_start + 1] = fv;
            b_last_diag[k + prev_b_end - b_start + 1] = cl;
        }
    }
}

// c_base_traceback, _c_dynamic_programming.pyx:165-182; *out = -1 where the reference runs off
// the loop (returns None)
__global__ void k_c_base_traceback(const double *curr, i64 curr_len, i64 curr_start,
    const double *next, i64 next_len, i64 next_start, i64 next_end, i64 sig_start, i64 m,
    i64 *out, i32 *status)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    i64 cnt = 1;
    *out = -1;
    for (i64 sp = sig_start; sp >= 0; sp--) {
        cnt += 1;
        if (cnt <= m || sp - 1 >= next_end) continue;
        if (sp <= curr_start) { *out = sp; return; }
        i64 ni = sp - next_start - 1, ci = sp - curr_start - 1;
        if (ni < 0) ni += next_len;
        if (ci < 0) ci += curr_len;
        if (ni < 0 || ni >= next_len || ci < 0 || ci >= curr_len) { *status = TBA_INTERNAL; return; }
        if (next[ni] > curr[ci]) { *out = sp; return; }
    }
}

// Row N4: the per-position log-likelihood ratio kernels of the model-comparison
// statistics, c_calc_llh_ratio / c_calc_llh_ratio_const_var / c_calc_scaled_llh_ratio_const_var
// (_c_helper.pyx:277-358), over many windows at once: window i covers width values from
// starts[i] of the per-base arrays, as tombo_stats.py:4042-4074 slices them; one thread per
// window, terms accumulated in index order like the reference.  kind 0: per-base variances;
// 1: constant variance ref_vars[starts[i]]; 2: the scaled form (par = scale, height, power).
// log / exp / pow are the device library's (not glibc's): parity is a stated tolerance here.
__global__ void k_c_llh_windows(int kind, const double *means, const double *ref_means,
    const double *alt_means, const double *ref_vars, const double *alt_vars, i64 width,
    const i64 *starts, i64 n_windows, double par0, double par1, double par2, double *out)
{
    for (i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x; w < n_windows; w += (i64)gridDim.x * blockDim.x) {
        const i64 s = starts[w];
        if (kind == 0) {
            double ref_z = 0.0, ref_lv = 0.0, alt_z = 0.0, alt_lv = 0.0;
            for (i64 i = s; i < s + width; i++) {
                const double rd = means[i] - ref_means[i];
                ref_z += (rd * rd) / ref_vars[i];
                ref_lv += log(ref_vars[i]);
                const double ad = means[i] - alt_means[i];
                alt_z += (ad * ad) / alt_vars[i];
                alt_lv += log(alt_vars[i]);
            }
            out[w] = alt_z + alt_lv - ref_z - ref_lv;
        } else if (kind == 1) {
            const double cv = ref_vars[s];
            double run = 0.0;
            for (i64 i = s; i < s + width; i++) {
                const double obs = means[i], rd = obs - ref_means[i], ad = obs - alt_means[i];
                run += ((ad * ad) - (rd * rd)) / cv;
            }
            out[w] = run;
        } else {
            const double cv = ref_vars[s];
            double run = 0.0;
            for (i64 i = s; i < s + width; i++) {
                const double rm = ref_means[i], am = alt_means[i];
                if (rm == am) continue;
                const double obs = means[i];
                const double sm = (am + rm) / 2;
                const double rd = obs - rm, ad = obs - am, sd = obs - sm;
                double md = am - rm;
                if (md < 0) md = md * -1;
                run += exp(-(sd * sd) / (par0 * cv)) * ((ad * ad) - (rd * rd)) / (cv * pow(md, par2) * par1);
            }
            out[w] = run;
        }
    }
}


// ---- row N4: per-read test statistics ---------------------------------------------------------
// compute_sample_compare_read_stats / compute_de_novo_read_stats (tombo_stats.py:3675-3873):
//   z = |mean - ref_mean| / ref_sd;  p = norm.cdf(-z) * 2  (NaN where z is NaN);
//   fm_offset > 0: Fisher's method over windows of 2 * fm_offset + 1 p-values
//   (calc_window_fishers_method, tombo_stats.py:2252-2271): p floored at `smallest`, logs summed in
//   index order, chi2.sf(-2 * sum, 2 * width) in its closed form for even degrees of freedom
//   exp(-x/2) * sum_{i < width} (x/2)^i / i!; the first / last fm_offset positions are NaN;
//   floor_out (de novo): the result is floored at `smallest` once more (np.maximum keeps NaN).
// Reads are CSR slices off[r]..off[r+1]; one thread per base.  erfc / log / exp are the device
// library's: parity with scipy is a stated tolerance (tests: 1e-12 relative).
__global__ void k_read_pvals(const double *means, const double *ref_means, const double *ref_sds,
    const i64 *off, i64 n_reads, i64 total, i64 fm, int floor_out, double smallest, double *out)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        // read of position i: binary search in off[]
        i64 lo = 0, hi = n_reads - 1;
        while (lo < hi) { const i64 mid = (lo + hi + 1) >> 1; if (off[mid] <= i) lo = mid; else hi = mid - 1; }
        const i64 a = off[lo], b = off[lo + 1];
        auto pval = [&](i64 k) {
            const double z = fabs(means[k] - ref_means[k]) / ref_sds[k];
            return z != z ? z : erfc(z * 0.70710678118654752440);
        };
        double res;
        if (fm <= 0) res = pval(i);
        else if (i - a < fm || b - i <= fm) res = NAN;
        else {
            double ls = 0.0;
            for (i64 k = i - fm; k <= i + fm; k++) {
                double p = pval(k);
                p = p < smallest ? smallest : p; // np.maximum: NaN stays NaN (comparison false)
                ls += log(p);
            }
            const double hx = -ls;                // x / 2 with x = -2 * log_sum
            double term = 1.0, acc = 1.0;
            for (i64 q = 1; q < 2 * fm + 1; q++) { term = term * hx / (double)q; acc += term; }
            res = exp(-hx) * acc;
        }
        if (floor_out && res < smallest) res = smallest;
        out[i] = res;
    }
}
