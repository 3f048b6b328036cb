/* oracle/tombo_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT (see tombo_oracle.h).
 *
 * CPU restatement of the reference resquiggle hot path.  Every function cites the reference
 * lines it follows (paths relative to /root/reference/tombo/).  Arithmetic is IEEE double,
 * two-operand, in the reference's source order; build with -O2 -ffp-contract=off.
 * Parity: PINNED against tests/golden/ (npz files) (generated from the live reference).
 */
#include "tombo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MASK_BASES 50              /* _default_parameters.py:69 */
#define MASK_FILL_Z_SCORE (-15.0)  /* _default_parameters.py:70 */
#define DEL_FIX_WINDOW 2           /* _default_parameters.py:72 */
#define MAX_DEL_FIX_WINDOW 10      /* _default_parameters.py:73 */
#define EXTRA_SIG_FACTOR 1.1       /* _default_parameters.py:67 */
#define SHIFT_CHANGE_THRESH 0.1    /* _default_parameters.py:169 */
#define SCALE_CHANGE_THRESH 0.1    /* _default_parameters.py:170 */

static inline i64 imin(i64 a, i64 b) { return a < b ? a : b; }
static inline i64 imax(i64 a, i64 b) { return a > b ? a : b; }

/* ---------------------------------------------------------------- numpy restatements ---- */

/* numpy pairwise summation (numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum) */
static double np_pairwise_sum(const double *a, i64 n)
{
    if (n < 8) {
        double res = 0.;
        for (i64 i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        i64 i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8) {
            r[0] += a[i + 0]; r[1] += a[i + 1]; r[2] += a[i + 2]; r[3] += a[i + 3];
            r[4] += a[i + 4]; r[5] += a[i + 5]; r[6] += a[i + 6]; r[7] += a[i + 7];
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        i64 n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

/* np.add.reduce / np.sum / np.mean numerator of a contiguous float64 vector: the ufunc
 * machinery feeds the inner loop in chunks of the default buffer size (8192 elements), each
 * chunk pairwise-summed and accumulated left to right onto 0.0 (measured against numpy 2.2.6,
 * see tests/test_oracle_golden.py::test_numpy_restatements). */
double orc_np_sum(const double *a, i64 n)
{
    double acc = 0.0;
    for (i64 i = 0; i < n; i += 8192) acc += np_pairwise_sum(a + i, n - i < 8192 ? n - i : 8192);
    return acc;
}

/* np.linspace(start, stop, num) (numpy/_core/function_base.py): arange*step + start, last
 * element forced to stop */
void orc_linspace(double start, double stop, i64 num, double *out)
{
    if (num <= 0) return;
    i64 div = num - 1;
    double delta = stop - start;
    if (div > 0) {
        double step = delta / (double)div;
        if (step == 0.0) {
            for (i64 i = 0; i < num; i++) out[i] = (((double)i / (double)div) * delta) + start;
        } else {
            for (i64 i = 0; i < num; i++) out[i] = ((double)i * step) + start;
        }
        out[num - 1] = stop;
    } else {
        out[0] = (0.0 * delta) + start;
    }
}

static void swapd(double *a, double *b) { double t = *a; *a = *b; *b = t; }

/* k-th smallest (0-based) by quickselect on a scratch copy; partitions v so v[k] is in place
 * and everything right of k is >= v[k] */
static double select_kth(double *v, i64 n, i64 k)
{
    i64 lo = 0, hi = n - 1;
    while (lo < hi) {
        i64 mid = lo + (hi - lo) / 2;
        if (v[mid] < v[lo]) swapd(&v[mid], &v[lo]);
        if (v[hi] < v[lo]) swapd(&v[hi], &v[lo]);
        if (v[hi] < v[mid]) swapd(&v[hi], &v[mid]);
        double piv = v[mid];
        i64 i = lo, j = hi;
        while (i <= j) {
            while (v[i] < piv) i++;
            while (v[j] > piv) j--;
            if (i <= j) { swapd(&v[i], &v[j]); i++; j--; }
        }
        if (k <= j) hi = j;
        else if (k >= i) lo = i;
        else break;
    }
    return v[k];
}

/* np.median of a float64 vector: middle order statistic, or (lo + hi) / 2 for even n
 * (numpy/lib/_function_base_impl.py _median: partition, then mean of the two middle values) */
double orc_median(const double *x, i64 n)
{
    if (n <= 0) return NAN;
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(v, x, sizeof(double) * (size_t)n);
    double res;
    if (n & 1) {
        res = select_kth(v, n, n / 2);
    } else {
        double hi = select_kth(v, n, n / 2);
        double lo = v[0];
        for (i64 i = 1; i < n / 2; i++) if (v[i] > lo) lo = v[i];
        res = (lo + hi) / 2.0;
    }
    free(v);
    return res;
}

/* ---------------------------------------------------------------- _c_helper.pyx ---------- */

/* c_new_means, _c_helper.pyx:59-71: sequential sum, one divide */
void orc_new_means(const double *sig, const i64 *segs, i64 n_segs, double *means)
{
    for (i64 idx = 0; idx < n_segs; idx++) {
        double s = 0;
        for (i64 j = segs[idx]; j < segs[idx + 1]; j++) s += sig[j];
        means[idx] = s / (double)(segs[idx + 1] - segs[idx]);
    }
}

/* c_new_mean_stds, _c_helper.pyx:38-57 */
void orc_new_mean_stds(const double *sig, const i64 *segs, i64 n_segs, double *means,
                       double *stds)
{
    for (i64 idx = 0; idx < n_segs; idx++) {
        i64 len = segs[idx + 1] - segs[idx];
        double s = 0;
        for (i64 j = segs[idx]; j < segs[idx + 1]; j++) s += sig[j];
        double m = s / (double)len;
        means[idx] = m;
        double v = 0;
        for (i64 j = segs[idx]; j < segs[idx + 1]; j++) {
            double d = sig[j] - m;
            v += d * d;
        }
        stds[idx] = sqrt(v / (double)len);
    }
}

/* c_apply_outlier_thresh, _c_helper.pyx:73-87 */
void orc_apply_outlier_thresh(const double *sig, i64 n, double lo, double hi, double *out)
{
    for (i64 i = 0; i < n; i++) {
        double v = sig[i];
        out[i] = v > hi ? hi : (v < lo ? lo : v);
    }
}

/* descending (score, index) order: what np.argsort(score)[::-1] yields for tie-free scores;
 * ties (which the reference leaves to numpy's unstable sort) are broken by higher index first,
 * i.e. a stable ascending sort reversed.  DESIGN.md documents the rule. */
typedef struct { double s; i64 i; } cand_t;
static int cand_cmp(const void *a, const void *b)
{
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return x->i > y->i ? -1 : (x->i < y->i ? 1 : 0);
}

/* the greedy pick shared by c_valid_cpts_w_cap (_c_helper.pyx:100-120) and
 * c_valid_cpts_w_cap_t_test (:185-202); num_cands is the early-stop bound each one uses */
static int greedy_pick(cand_t *cand, i64 n_scores, i64 num_cands, i64 min_base_obs, i64 width,
                       i64 num_cpts, i64 *cpts)
{
    if (n_scores <= 0) return ORC_INTERNAL;
    qsort(cand, (size_t)n_scores, sizeof(cand_t), cand_cmp);
    unsigned char *black = (unsigned char *)calloc((size_t)(n_scores + 2 * min_base_obs + 2), 1);
    i64 off = min_base_obs;
    i64 p0 = cand[0].i;
    cpts[0] = p0 + width;
    for (i64 q = p0 - min_base_obs + 1; q < p0 + min_base_obs; q++) black[q + off] = 1;
    i64 cand_idx = 1, added = 1;
    int rc = ORC_OK;
    while (added < num_cpts) {
        if (cand_idx >= n_scores) { rc = ORC_INTERNAL; break; } /* IndexError upstream */
        i64 cp = cand[cand_idx].i;
        if (!black[cp + off]) {
            cpts[added++] = cp + width;
            for (i64 q = cp - min_base_obs + 1; q < cp + min_base_obs; q++) black[q + off] = 1;
        }
        cand_idx++;
        if (cand_idx >= num_cands) { rc = ORC_FEWER_CPTS; break; }
    }
    free(black);
    return rc;
}

static int i64_cmp(const void *a, const void *b)
{
    i64 x = *(const i64 *)a, y = *(const i64 *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* c_valid_cpts_w_cap, _c_helper.pyx:89-120, plus the .sort() of tombo_helper.py:76-82 */
int orc_valid_cpts_w_cap(const double *sig, i64 n, i64 min_base_obs, i64 width, i64 num_cpts,
                         i64 *cpts)
{
    i64 n_scores = n + 1 - 2 * width;
    if (n_scores <= 0 || num_cpts <= 0) return ORC_INTERNAL;
    double *cs = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    cs[0] = 0.0;
    for (i64 i = 0; i < n; i++) cs[i + 1] = cs[i] + sig[i]; /* np.cumsum: left to right */
    cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)n_scores);
    for (i64 k = 0; k < n_scores; k++) {
        /* |(2*c[k+w]) - c[k] - c[k+2w]|, evaluated left to right */
        cand[k].s = fabs(((2 * cs[k + width]) - cs[k]) - cs[k + 2 * width]);
        cand[k].i = k;
    }
    int rc = greedy_pick(cand, n_scores, n_scores - 2 * width, min_base_obs, width, num_cpts,
                         cpts);
    free(cand);
    free(cs);
    if (rc == ORC_OK) qsort(cpts, (size_t)num_cpts, sizeof(i64), i64_cmp);
    return rc;
}

/* c_valid_cpts_w_cap_t_test, _c_helper.pyx:144-202 (+ sort) */
int orc_valid_cpts_w_cap_t_test(const double *sig, i64 n, i64 min_base_obs, i64 width,
                                i64 num_cpts, i64 *cpts)
{
    i64 num_cands = n - 2 * width;
    if (num_cands <= 0 || num_cpts <= 0) return ORC_INTERNAL;
    cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)num_cands);
    for (i64 pos = 0; pos < num_cands; pos++) {
        double m1 = 0, m2 = 0, var1 = 0, var2 = 0, d;
        for (i64 j = 0; j < width; j++) m1 += sig[pos + j];
        m1 /= (double)width;
        for (i64 j = 0; j < width; j++) m2 += sig[pos + width + j];
        m2 /= (double)width;
        for (i64 j = 0; j < width; j++) { d = sig[pos + j] - m1; var1 += d * d; }
        for (i64 j = 0; j < width; j++) { d = sig[pos + width + j] - m2; var2 += d * d; }
        double t;
        if (var1 + var2 == 0) t = 0.0;
        else if (m1 > m2) t = (m1 - m2) / sqrt(var1 + var2);
        else t = (m2 - m1) / sqrt(var1 + var2);
        cand[pos].s = t;
        cand[pos].i = pos;
    }
    int rc = greedy_pick(cand, num_cands, num_cands, min_base_obs, width, num_cpts, cpts);
    free(cand);
    if (rc == ORC_OK) qsort(cpts, (size_t)num_cpts, sizeof(i64), i64_cmp);
    return rc;
}

/* c_compute_slopes, _c_helper.pyx:362-377: itertools.combinations order (i < j) */
void orc_compute_slopes(const double *ev, const double *model, i64 n, double max_slope,
                        double *slopes)
{
    i64 s = 0;
    for (i64 i = 0; i < n; i++)
        for (i64 j = i + 1; j < n; j++, s++)
            slopes[s] = (ev[i] == ev[j]) ? max_slope
                                         : (model[i] - model[j]) / (ev[i] - ev[j]);
}

/* c_calc_llh_ratio, _c_helper.pyx:277-296 */
double orc_calc_llh_ratio(const double *means, const double *ref_means, const double *alt_means,
                          const double *ref_vars, const double *alt_vars, i64 n)
{
    double ref_z_sum = 0.0, ref_log_var_sum = 0.0, alt_z_sum = 0.0, alt_log_var_sum = 0.0;
    for (i64 i = 0; i < n; i++) {
        double ref_diff = means[i] - ref_means[i];
        ref_z_sum += (ref_diff * ref_diff) / ref_vars[i];
        ref_log_var_sum += log(ref_vars[i]);
        double alt_diff = means[i] - alt_means[i];
        alt_z_sum += (alt_diff * alt_diff) / alt_vars[i];
        alt_log_var_sum += log(alt_vars[i]);
    }
    return alt_z_sum + alt_log_var_sum - ref_z_sum - ref_log_var_sum;
}

/* c_calc_llh_ratio_const_var, _c_helper.pyx:298-311 */
double orc_calc_llh_ratio_const_var(const double *means, const double *ref_means,
                                    const double *alt_means, i64 n, double const_var)
{
    double run = 0.0;
    for (i64 i = 0; i < n; i++) {
        double obs = means[i];
        double ref_diff = obs - ref_means[i];
        double alt_diff = obs - alt_means[i];
        run += ((alt_diff * alt_diff) - (ref_diff * ref_diff)) / const_var;
    }
    return run;
}

/* c_calc_scaled_llh_ratio_const_var, _c_helper.pyx:313-358 */
double orc_calc_scaled_llh_ratio_const_var(const double *means, const double *ref_means,
    const double *alt_means, i64 n, double const_var, double scale_factor,
    double density_height_factor, double density_height_power)
{
    double run = 0.0;
    for (i64 i = 0; i < n; i++) {
        double ref_mean = ref_means[i], alt_mean = alt_means[i];
        if (ref_mean == alt_mean) continue;
        double obs = means[i];
        double scale_mean = (alt_mean + ref_mean) / 2;
        double ref_diff = obs - ref_mean, alt_diff = obs - alt_mean;
        double scale_diff = obs - scale_mean;
        double means_diff = alt_mean - ref_mean;
        if (means_diff < 0) means_diff = means_diff * -1;
        run += exp(-(scale_diff * scale_diff) / (scale_factor * const_var)) *
               ((alt_diff * alt_diff) - (ref_diff * ref_diff)) /
               (const_var * pow(means_diff, density_height_power) * density_height_factor);
    }
    return run;
}

/* ------------------------------------------------------- _c_dynamic_programming.pyx ------ */

/* c_base_z_scores, _c_dynamic_programming.pyx:17-32 (negative half z-score) */
void orc_base_z_scores(const double *sig, i64 n, double mean, double sd, int winsor,
                       double max_half_z, double *out)
{
    for (i64 i = 0; i < n; i++) {
        double z = (sig[i] - mean) / sd;
        if (z > 0) z = -z;
        if (winsor && z < -max_half_z) z = -max_half_z;
        out[i] = z;
    }
}

/* c_argmax, pyx:186-197: first index of the maximum (strict >) */
static i64 orc_argmax(const double *v, i64 n)
{
    double mv = v[0];
    i64 mp = 0;
    for (i64 i = 1; i < n; i++) if (v[i] > mv) { mv = v[i]; mp = i; }
    return mp;
}

/* c_process_band, pyx:202-236: band positions 1..bw-1 of row seq_pos+1 */
static void orc_process_band(double *fwd, int8_t *tb, const double *z, double stay_pen,
                             double skip_pen, i64 bw, i64 diff, i64 seq_pos)
{
    const double *prev = fwd + seq_pos * bw;
    double *cur = fwd + (seq_pos + 1) * bw;
    int8_t *ctb = tb + (seq_pos + 1) * bw;
    for (i64 b = 1; b < bw; b++) {
        double pz = z[b];
        i64 pb = b + diff;
        double best = (cur[b - 1] - stay_pen) + pz;
        int8_t from = 0;
        if (pb - 1 < bw) {
            double d = prev[pb - 1] + pz;
            if (d > best) { best = d; from = 2; }
            if (pb < bw) {
                double s = prev[pb] - skip_pen;
                if (s > best) { best = s; from = 1; }
            }
        }
        cur[b] = best;
        ctb[b] = from;
    }
}

/* c_banded_forward_pass, pyx:240-279.  fwd/tb are (n_bases+1) x bw, row 0 = zeros. */
void orc_banded_forward_pass(const double *z, i64 n_bases, i64 bw, const i64 *event_starts,
                             double skip_pen, double stay_pen, double *fwd, int8_t *tb)
{
    for (i64 i = 0; i < bw; i++) { fwd[i] = 0.0; tb[i] = 0; }
    for (i64 sp = 0; sp < n_bases; sp++) {
        i64 diff = sp > 0 ? event_starts[sp] - event_starts[sp - 1] : 0;
        if (sp == 0 || diff == 0) {
            fwd[(sp + 1) * bw] = fwd[sp * bw] - skip_pen;
            tb[(sp + 1) * bw] = 1;
        } else {
            fwd[(sp + 1) * bw] = fwd[sp * bw + diff - 1] + z[sp * bw];
            tb[(sp + 1) * bw] = 2;
        }
        orc_process_band(fwd, tb, z + sp * bw, stay_pen, skip_pen, bw, diff, sp);
    }
}

/* c_adaptive_banded_forward_pass, pyx:314-412 (fwd/tb/event_starts updated in place) */
int orc_adaptive_banded_forward_pass(double *fwd, int8_t *tb, i64 n_bases, i64 bw,
    i64 *event_starts, const double *event_means, i64 n_events, const double *ref_means,
    const double *ref_sds, double z_shift, double skip_pen, double stay_pen, i64 start_seq_pos,
    double mask_fill_z, int winsor, double max_half_z)
{
    i64 half_bw = bw / 2;
    double *z = (double *)malloc(sizeof(double) * (size_t)bw);
    for (i64 sp = start_seq_pos; sp < n_bases; sp++) {
        i64 prev_start = event_starts[sp - 1];
        i64 cur_start = prev_start + orc_argmax(fwd + sp * bw, bw) - half_bw + 1;
        if (cur_start < prev_start) cur_start = prev_start;
        if (cur_start >= n_events) {
            if (sp < n_bases - 2) { free(z); return ORC_ADAPT_BEYOND; }
            cur_start = n_events - 1;
        }
        event_starts[sp] = cur_start;
        double mu = ref_means[sp], sd = ref_sds[sp];
        i64 n_real = cur_start + bw <= n_events ? bw : n_events - cur_start;
        for (i64 b = 0; b < n_real; b++) {
            double pz = (event_means[cur_start + b] - mu) / sd;
            if (pz < 0) pz = -pz;
            if (winsor) pz = pz < max_half_z ? pz : max_half_z; /* C++ std::min(pz, mh) */
            z[b] = z_shift - pz;
        }
        for (i64 b = n_real; b < bw; b++) z[b] = mask_fill_z;
        i64 diff = cur_start - prev_start;
        if (diff == 0) {
            fwd[(sp + 1) * bw] = fwd[sp * bw] - skip_pen;
            tb[(sp + 1) * bw] = 1;
        } else {
            fwd[(sp + 1) * bw] = fwd[sp * bw + diff - 1] + z[0];
            tb[(sp + 1) * bw] = 2;
        }
        orc_process_band(fwd, tb, z, stay_pen, skip_pen, bw, diff, sp);
    }
    free(z);
    return ORC_OK;
}

/* c_banded_traceback, pyx:281-310.  The reference indexes with Python wrap-around enabled;
 * a negative band position reads from the end of the row. */
int orc_banded_traceback(const int8_t *tb, i64 n_bases, i64 bw, const i64 *event_starts,
                         i64 band_pos, i64 band_boundary_thresh, i64 *seq_poss)
{
#define TB_AT(r, b) tb[(r) * bw + ((b) < 0 ? (b) + bw : (b))]
    i64 cur_ev = band_pos + event_starts[n_bases - 1];
    seq_poss[n_bases] = cur_ev + 1;
    for (i64 r = n_bases; r > 0; r--) {
        band_pos = cur_ev - event_starts[r - 1];
        if (band_pos >= bw || band_pos < -bw) return ORC_INTERNAL;
        while (TB_AT(r, band_pos) == 0) {
            band_pos--;
            if (band_pos < -bw) return ORC_INTERNAL;
        }
        if (TB_AT(r, band_pos) == 2) band_pos--;
        if (band_boundary_thresh >= 0 &&
            imin(band_pos, bw - band_pos - 1) < band_boundary_thresh)
            return ORC_BEYOND_BANDWIDTH;
        cur_ev = event_starts[r - 1] + band_pos;
        seq_poss[r - 1] = cur_ev + 1;
    }
#undef TB_AT
    return ORC_OK;
}

/* ---------------------------------------------------------------- tombo_stats.py --------- */

/* ts.normalize_raw_signal, tombo_stats.py:482-573, for the modes the hot path uses:
 * 'median' / 'median_const_scale' / given scale_values.  sv_out = shift, scale, lower, upper
 * (NaN for None). */
int orc_normalize_raw_signal(const double *raw, i64 n, const orc_opts *o, int use_sv,
                             double sv_shift, double sv_scale, int sv_has_lims, double sv_lo,
                             double sv_hi, double *norm, double *sv_out)
{
    double shift, scale;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (use_sv) {
        shift = sv_shift;
        scale = sv_scale;
    } else {
        shift = orc_median(raw, n);
        if (o->has_const_scale) {
            scale = o->const_scale;
        } else {
            for (i64 i = 0; i < n; i++) tmp[i] = fabs(raw[i] - shift);
            scale = orc_median(tmp, n);
        }
    }
    /* the reference runs under np.seterr(all='raise') (resquiggle.py:29, tombo_stats.py:19): a scale
     * of exactly 0 -- the MAD of a flat signal -- makes the division below raise FloatingPointError
     * ('divide by zero' / 'invalid value'), an unexpected error of the read
     * (tests/golden/gen_golden_degenerate.py records the live reference doing so) */
    if (scale == 0.0) { free(tmp); return ORC_INTERNAL; }
    for (i64 i = 0; i < n; i++) norm[i] = (raw[i] - shift) / scale;
    double lo = NAN, hi = NAN;
    int have = 0;
    if (!use_sv && o->has_outlier_thresh) {
        double med = orc_median(norm, n);
        for (i64 i = 0; i < n; i++) tmp[i] = fabs(norm[i] - med);
        double mad = orc_median(tmp, n);
        lo = med - (mad * o->outlier_thresh);
        hi = med + (mad * o->outlier_thresh);
        have = 1;
    } else if (use_sv && sv_has_lims) {
        lo = sv_lo;
        hi = sv_hi;
        have = 1;
    }
    if (have) orc_apply_outlier_thresh(norm, n, lo, hi, norm);
    free(tmp);
    sv_out[0] = shift; sv_out[1] = scale; sv_out[2] = lo; sv_out[3] = hi;
    return ORC_OK;
}

/* ts.remove_stall_cpts, tombo_stats.py:1576-1597 (same interval walk) */
static i64 orc_remove_stall_cpts(const i64 *stall, i64 n_stall, i64 *cpts, i64 n)
{
    if (n_stall == 0) return n;
    i64 cur = 0, out = 0;
    for (i64 i = 0; i < n; i++) {
        i64 c = cpts[i];
        while (c > stall[2 * cur + 1]) {
            if (cur + 1 >= n_stall) break;
            cur++;
        }
        if (!(stall[2 * cur] < c && c < stall[2 * cur + 1])) cpts[out++] = c;
    }
    return out;
}

/* ts.score_valid_bases, tombo_stats.py:2340-2362 with ts.get_read_seg_score :2327-2338.
 * returns -1 when no valid base exists ('Invalid path through read start') */
static double orc_score_valid_bases(const i64 *tb, i64 n_tb, const double *event_means,
                                    const double *ref_means, const double *ref_sds)
{
    double *v = (double *)malloc(sizeof(double) * (size_t)n_tb);
    i64 nv = 0;
    for (i64 i = 0; i + 1 < n_tb; i++) {
        if (tb[i] == tb[i + 1]) continue;
        double m = orc_np_sum(event_means + tb[i], tb[i + 1] - tb[i]) /
                   (double)(tb[i + 1] - tb[i]);
        v[nv++] = fabs((m - ref_means[i]) / ref_sds[i]);
    }
    double res = nv ? orc_np_sum(v, nv) / (double)nv : -1.0;
    free(v);
    return res;
}

/* ts.calc_kmer_fitted_shift_scale(method='theil_sen'), tombo_stats.py:401-425,446-450
 * (the caller has already applied the np.random.choice subsample).
 * out4 = shift, scale, shift_corr_factor, scale_corr_factor */
int orc_theil_sen(const double *ev, const double *model, i64 n, double prev_shift,
                  double prev_scale, double *out4)
{
    i64 ns = n * (n - 1) / 2;
    if (ns <= 0) return ORC_INTERNAL;
    double *slopes = (double *)malloc(sizeof(double) * (size_t)ns);
    orc_compute_slopes(ev, model, n, 1000.0, slopes);
    double slope = orc_median(slopes, ns);
    free(slopes);
    double *t = (double *)malloc(sizeof(double) * (size_t)n);
    for (i64 i = 0; i < n; i++) t[i] = model[i] - (slope * ev[i]);
    double inter = orc_median(t, n);
    free(t);
    if (slope == 0) return ORC_RESCALE_FAIL;
    double scale_corr = 1 / slope;
    double shift_corr = -inter / slope;
    out4[0] = prev_shift + (shift_corr * prev_scale);
    out4[1] = prev_scale * scale_corr;
    out4[2] = shift_corr;
    out4[3] = scale_corr;
    return ORC_OK;
}

/* ---------------------------------------------------------------- resquiggle.py ---------- */

/* shifted half z-score row used by the numpy twins (resquiggle.py:574-582, 712-720):
 * z_shift - min(max_half_z, |e - mu| / sd) */
static void orc_shifted_z_row(const double *ev, i64 n, double mu, double sd, const orc_params *p,
                              double *out)
{
    for (i64 i = 0; i < n; i++) {
        double a = fabs(ev[i] - mu) / sd;
        if (p->do_winsorize_z) a = p->max_half_z_score < a ? p->max_half_z_score : a;
        out[i] = p->z_shift - a;
    }
}

/* rq.find_seq_start_in_events, resquiggle.py:685-752.  check_score: seq_samp_type given. */
static int orc_find_seq_start_in_events(const double *event_means, i64 n_ev,
    const double *ref_means, const double *ref_sds, i64 n_ref, const orc_params *p,
    i64 num_bases, i64 num_events, int check_score, double thresh, i64 *start_loc,
    double *events_per_base)
{
    if (n_ev < num_events + num_bases) return ORC_READ_TOO_SHORT_START;
    if (n_ref < num_bases) return ORC_MAP_TOO_SHORT_START;
    double *z = (double *)malloc(sizeof(double) * (size_t)(num_bases * num_events));
    i64 *starts = (i64 *)malloc(sizeof(i64) * (size_t)num_bases);
    for (i64 r = 0; r < num_bases; r++) {
        orc_shifted_z_row(event_means + r, num_events, ref_means[r], ref_sds[r], p,
                          z + r * num_events);
        starts[r] = r;
    }
    double *fwd = (double *)malloc(sizeof(double) * (size_t)((num_bases + 1) * num_events));
    int8_t *tb = (int8_t *)malloc((size_t)((num_bases + 1) * num_events));
    orc_banded_forward_pass(z, num_bases, num_events, starts, p->skip_pen, p->stay_pen, fwd, tb);
    i64 top = orc_argmax(fwd + num_bases * num_events, num_events);
    i64 *stb = (i64 *)malloc(sizeof(i64) * (size_t)(num_bases + 1));
    int rc = orc_banded_traceback(tb, num_bases, num_events, starts, top, -1, stb);
    if (rc == ORC_OK && check_score) {
        double sc = orc_score_valid_bases(stb, num_bases + 1, event_means, ref_means, ref_sds);
        if (sc < 0) rc = ORC_INVALID_START_PATH;
        else if (sc > thresh) rc = ORC_POOR_START;
    }
    if (rc == ORC_OK) {
        *events_per_base = (double)(stb[num_bases] - stb[0]) / (double)(num_bases + 1);
        *start_loc = stb[0];
    }
    free(z); free(starts); free(fwd); free(tb); free(stb);
    return rc;
}

/* rq.find_static_base_assignment, resquiggle.py:547-600 */
static int orc_find_static_base_assignment(const double *event_means, i64 n_ev,
    const double *ref_means, const double *ref_sds, i64 seq_len, const orc_params *p,
    i64 *read_tb, orc_debug *dbg)
{
    i64 mask_len = imin(seq_len, n_ev) / 4;
    i64 bw = n_ev - mask_len;
    if (bw <= 0 || seq_len - 2 * mask_len < 0) return ORC_INTERNAL;
    i64 *starts = (i64 *)malloc(sizeof(i64) * (size_t)seq_len);
    double *ls = (double *)malloc(sizeof(double) * (size_t)(2 * mask_len + 1));
    i64 nz = seq_len - 2 * mask_len;
    for (i64 i = 0; i < nz; i++) starts[i] = 0;
    orc_linspace(0.0, (double)mask_len, 2 * mask_len, ls);
    for (i64 i = 0; i < 2 * mask_len; i++) starts[nz + i] = (i64)ls[i];
    free(ls);
    double *z = (double *)malloc(sizeof(double) * (size_t)(seq_len * bw));
    for (i64 r = 0; r < seq_len; r++)
        orc_shifted_z_row(event_means + starts[r], bw, ref_means[r], ref_sds[r], p, z + r * bw);
    double *fwd = (double *)malloc(sizeof(double) * (size_t)((seq_len + 1) * bw));
    int8_t *tb = (int8_t *)malloc((size_t)((seq_len + 1) * bw));
    orc_banded_forward_pass(z, seq_len, bw, starts, p->skip_pen, p->stay_pen, fwd, tb);
    i64 top = orc_argmax(fwd + seq_len * bw, bw);
    if (dbg) {
        if (dbg->band_event_starts) memcpy(dbg->band_event_starts, starts, sizeof(i64) * (size_t)seq_len);
        if (dbg->fwd_last_row) {
            memcpy(dbg->fwd_last_row, fwd + seq_len * bw, sizeof(double) * (size_t)bw);
            dbg->fwd_last_row_len = bw;
        }
    }
    int rc = orc_banded_traceback(tb, seq_len, bw, starts, top, -1, read_tb);
    free(z); free(fwd); free(tb); free(starts);
    return rc;
}

/* rq._get_masked_start_fwd_pass, resquiggle.py:607-683.  Fills rows 0..mask_seq_len of
 * fwd/tb (bandwidth-wide) and band_event_starts[0..mask_seq_len). */
static int orc_masked_start_fwd_pass(const double *event_means, i64 n_ev,
    const double *ref_means, const double *ref_sds, i64 mapped_start_offset,
    const orc_params *p, double events_per_base, double *fwd, int8_t *tb,
    i64 *band_event_starts, i64 seq_len, i64 *mask_seq_len_out)
{
    i64 bw = p->bandwidth;
    if (n_ev - mapped_start_offset < bw) return ORC_STARTS_TOO_FAR;
    int winsor = (int)p->do_winsorize_z;
    double mh = winsor ? p->max_half_z_score : 0.0;
    i64 half_bw = bw / 2;
    i64 start_pos = half_bw <= mapped_start_offset ? 0 : mapped_start_offset - half_bw;
    i64 tmp_seq_len = imax(imax(half_bw, MASK_BASES),
                           (i64)((double)(half_bw + 1) / events_per_base)) + 1;
    double *ls = (double *)malloc(sizeof(double) * (size_t)tmp_seq_len);
    orc_linspace((double)start_pos,
                 (double)start_pos + ((double)tmp_seq_len * events_per_base), tmp_seq_len, ls);
    i64 first = -1;
    for (i64 i = 0; i < tmp_seq_len; i++)
        if ((i64)ls[i] >= mapped_start_offset) { first = i + 2; break; }
    if (first < 0) { free(ls); return ORC_INTERNAL; } /* StopIteration upstream */
    i64 mask_seq_len = imax(MASK_BASES, first);
    /* numpy slicing clamps [:mask_seq_len] to the array length */
    if (mask_seq_len > tmp_seq_len) mask_seq_len = tmp_seq_len;
    if (mask_seq_len > seq_len) { free(ls); return ORC_INTERNAL; } /* broadcast error upstream */
    for (i64 i = 0; i < mask_seq_len; i++) band_event_starts[i] = (i64)ls[i];
    free(ls);
    double msp[MASK_BASES];
    orc_linspace((double)(mapped_start_offset + 1),
                 (double)(band_event_starts[MASK_BASES - 1] + bw), MASK_BASES, msp);
    double *z = (double *)malloc(sizeof(double) * (size_t)(mask_seq_len * bw));
    double fill = MASK_FILL_Z_SCORE - p->z_shift;
    int rc = ORC_OK;
    for (i64 sp = 0; sp < mask_seq_len && rc == ORC_OK; sp++) {
        i64 ev_pos = band_event_starts[sp];
        i64 sml = imax(mapped_start_offset - ev_pos, 0);
        i64 eml = sp >= MASK_BASES ? 0 : bw - ((i64)msp[sp] - ev_pos);
        if (ev_pos + bw - eml > n_ev) eml = ev_pos + bw - n_ev;
        i64 lo = ev_pos + sml, hi = ev_pos + bw - eml;
        /* python slice semantics of event_means[lo:hi] (lo >= 0 here) */
        i64 hi_c = hi > n_ev ? n_ev : hi;
        if (hi < 0) hi_c = imax(hi + n_ev, 0);
        i64 lo_c = lo > n_ev ? n_ev : lo;
        i64 nzs = hi_c > lo_c ? hi_c - lo_c : 0;
        i64 eml_n = eml > 0 ? eml : 0; /* [x] * negative == [] */
        if (sml + nzs + eml_n != bw) { rc = ORC_MASK_TOO_FEW; break; }
        double *row = z + sp * bw;
        for (i64 i = 0; i < sml; i++) row[i] = fill;
        orc_base_z_scores(event_means + lo_c, nzs, ref_means[sp], ref_sds[sp], winsor, mh,
                          row + sml);
        for (i64 i = 0; i < eml_n; i++) row[sml + nzs + i] = fill;
        for (i64 i = 0; i < bw; i++) row[i] = row[i] + p->z_shift; /* shifted_z_scores += z_shift */
    }
    if (rc == ORC_OK)
        orc_banded_forward_pass(z, mask_seq_len, bw, band_event_starts, p->skip_pen,
                                p->stay_pen, fwd, tb);
    free(z);
    *mask_seq_len_out = mask_seq_len;
    return rc;
}

/* rq._trim_traceback, resquiggle.py:754-764 */
static int orc_trim_traceback(i64 *tb, i64 n, i64 events_len)
{
    i64 i = 0;
    while (tb[i] < 0) { tb[i] = 0; i++; if (i >= n) return ORC_INTERNAL; }
    i64 j = 1;
    while (tb[n - j] > events_len) { tb[n - j] = events_len; j++; if (j > n) return ORC_INTERNAL; }
    return ORC_OK;
}

/* rq.find_adaptive_base_assignment (start_clip_bases=None), resquiggle.py:866-1050, with
 * get_short_read_results :885-893 and get_rel_raw_coords :858-864.
 * Output: segs[seq_len+1] (relative), read_start_rel_to_raw. */
static int orc_find_adaptive_base_assignment(const i64 *valid_cpts, const double *event_means,
    i64 n_ev, const double *ref_means, const double *ref_sds, i64 seq_len, const orc_params *p,
    const orc_opts *o, i64 *segs, i64 *read_start, orc_debug *dbg)
{
    i64 *read_tb = (i64 *)malloc(sizeof(i64) * (size_t)(seq_len + 1));
    int rc = ORC_OK;
    int use_static = 0;
    i64 mapped_start = 0;
    double epb = 0;
    i64 clip = 0;
    if (n_ev < p->start_bw + p->start_n_bases || seq_len < p->start_n_bases) {
        use_static = 1;
    } else {
        rc = orc_find_seq_start_in_events(event_means, n_ev, ref_means, ref_sds, seq_len, p,
                                          p->start_n_bases, p->start_bw,
                                          (int)o->check_start_score, o->sig_match_thresh,
                                          &mapped_start, &epb);
        if (dbg && rc == ORC_OK) {
            dbg->start_calls[0] = (double)mapped_start; dbg->start_calls[1] = epb;
            dbg->n_start_calls = 1;
        }
        if (rc != ORC_OK && rc != ORC_INTERNAL) { /* except th.TomboError */
            if (n_ev < p->start_save_bw + p->start_n_bases) {
                use_static = 1;
                rc = ORC_OK;
            } else {
                rc = orc_find_seq_start_in_events(event_means, n_ev, ref_means, ref_sds,
                                                  seq_len, p, p->start_n_bases,
                                                  p->start_save_bw, 0, 0.0, &mapped_start,
                                                  &epb);
                if (dbg && rc == ORC_OK) {
                    dbg->start_calls[2] = (double)mapped_start; dbg->start_calls[3] = epb;
                    dbg->n_start_calls = 2;
                }
            }
        }
        if (rc != ORC_OK) { free(read_tb); return rc; }
    }
    if (!use_static) {
        if (epb == 0) { free(read_tb); return ORC_OPEN_PORE; }
        i64 half_bw = p->bandwidth / 2;
        i64 offset;
        if (mapped_start < half_bw) { clip = 0; offset = mapped_start; }
        else { clip = mapped_start - half_bw; offset = half_bw; }
        if ((i64)((double)(half_bw + 1) / epb) >= seq_len ||
            n_ev - offset - clip < p->bandwidth) {
            use_static = 1;
        } else {
            /* run_fwd_pass, resquiggle.py:895-942 */
            i64 bw = p->bandwidth;
            double *fwd = (double *)malloc(sizeof(double) * (size_t)((seq_len + 1) * bw));
            int8_t *tb = (int8_t *)malloc((size_t)((seq_len + 1) * bw));
            i64 *bes = (i64 *)malloc(sizeof(i64) * (size_t)seq_len);
            i64 msl = 0;
            rc = orc_masked_start_fwd_pass(event_means + clip, n_ev - clip, ref_means, ref_sds,
                                           offset, p, epb, fwd, tb, bes, seq_len, &msl);
            if (rc == ORC_OK)
                rc = orc_adaptive_banded_forward_pass(
                    fwd, tb, seq_len, bw, bes, event_means + clip, n_ev - clip, ref_means,
                    ref_sds, p->z_shift, p->skip_pen, p->stay_pen, msl, MASK_FILL_Z_SCORE,
                    (int)p->do_winsorize_z, p->do_winsorize_z ? p->max_half_z_score : 0.0);
            if (rc == ORC_OK) {
                i64 top = orc_argmax(fwd + seq_len * bw, bw);
                if (dbg) {
                    dbg->mask_seq_len = msl;
                    if (dbg->band_event_starts)
                        memcpy(dbg->band_event_starts, bes, sizeof(i64) * (size_t)seq_len);
                    if (dbg->fwd_last_row) {
                        memcpy(dbg->fwd_last_row, fwd + seq_len * bw, sizeof(double) * (size_t)bw);
                        dbg->fwd_last_row_len = bw;
                    }
                }
                rc = orc_banded_traceback(tb, seq_len, bw, bes, top, p->band_bound_thresh,
                                          read_tb);
            }
            if (rc == ORC_OK) rc = orc_trim_traceback(read_tb, seq_len + 1, n_ev - clip);
            free(fwd); free(tb); free(bes);
            if (rc != ORC_OK) { free(read_tb); return rc; }
        }
    }
    if (use_static) {
        clip = 0;
        rc = orc_find_static_base_assignment(event_means, n_ev, ref_means, ref_sds, seq_len, p,
                                             read_tb, dbg);
        if (rc != ORC_OK) { free(read_tb); return rc; }
        for (i64 i = 0; i <= seq_len; i++)
            if (read_tb[i] < -(n_ev + 1) || read_tb[i] > n_ev) { free(read_tb); return ORC_INTERNAL; }
    }
    if (dbg) {
        dbg->used_static = use_static;
        if (dbg->read_tb) memcpy(dbg->read_tb, read_tb, sizeof(i64) * (size_t)(seq_len + 1));
    }
    /* get_rel_raw_coords: valid_cpts[clip:][read_tb] (numpy negative indices wrap) */
    i64 n_c = n_ev + 1 - clip;
    for (i64 i = 0; i <= seq_len; i++) {
        i64 t = read_tb[i];
        if (t < 0) t += n_c;
        segs[i] = valid_cpts[clip + t];
    }
    *read_start = segs[0];
    for (i64 i = seq_len; i >= 0; i--) segs[i] -= segs[0];
    free(read_tb);
    return ORC_OK;
}

/* raw-signal DP of one deletion window: c_reg_z_scores (pyx:34-97) with reg_start=0,
 * reg_end=n, max_base_shift=n; rq.raw_forward_pass (resquiggle.py:345-380) over
 * c_base_forward_pass (pyx:99-163); rq.raw_traceback (:382-400) over c_base_traceback
 * (pyx:165-182).  new_segs[n-1] = interior boundaries relative to the window signal. */
static int orc_raw_window_dp(const double *sig, i64 L, const double *means, const double *sds,
                             i64 n, i64 m, int winsor, double mh, i64 *new_segs)
{
    if (n < 2) return ORC_INTERNAL;
    i64 *bs = (i64 *)malloc(sizeof(i64) * (size_t)n), *be = (i64 *)malloc(sizeof(i64) * (size_t)n);
    i64 *off = (i64 *)malloc(sizeof(i64) * (size_t)(n + 1));
    /* c_reg_z_scores start/end clipping with r_b_starts[0] = 0, r_b_starts[n] = L */
    for (i64 i = 0; i < n; i++) bs[i] = i == 0 ? 0 : (0 < bs[i - 1] + m ? bs[i - 1] + m : 0);
    for (i64 i = n - 1; i >= 0; i--)
        be[i] = i == n - 1 ? L : (L > be[i + 1] - m ? be[i + 1] - m : L);
    off[0] = 0;
    int rc = ORC_OK;
    for (i64 i = 0; i < n; i++) {
        if (be[i] <= bs[i]) rc = ORC_INTERNAL;
        off[i + 1] = off[i] + (be[i] > bs[i] ? be[i] - bs[i] : 0);
    }
    if (rc != ORC_OK) { free(bs); free(be); free(off); return rc; }
    double *z = (double *)malloc(sizeof(double) * (size_t)off[n]);
    double *fw = (double *)malloc(sizeof(double) * (size_t)off[n]);
    i64 *ld = (i64 *)malloc(sizeof(i64) * (size_t)off[n]);
    for (i64 i = 0; i < n; i++)
        orc_base_z_scores(sig + bs[i], be[i] - bs[i], means[i], sds[i], winsor, mh, z + off[i]);
    /* first row: cumsum, last_diag = m */
    {
        double acc = 0;
        for (i64 k = 0; k < be[0] - bs[0]; k++) {
            acc = k == 0 ? z[k] : acc + z[k];
            fw[k] = acc;
            ld[k] = m;
        }
    }
    i64 max_len = 0;
    for (i64 i = 0; i < n; i++) max_len = imax(max_len, be[i] - bs[i]);
    double *cum = (double *)malloc(sizeof(double) * (size_t)max_len);
    for (i64 i = 1; i < n && rc == ORC_OK; i++) {
        const double *pz = z + off[i - 1], *pf = fw + off[i - 1];
        const i64 *pl = ld + off[i - 1];
        const double *bz = z + off[i];
        double *bf = fw + off[i];
        i64 *bl = ld + off[i];
        i64 ps = bs[i - 1], pe = be[i - 1], b_s = bs[i], b_e = be[i];
        i64 plen = pe - ps;
        { /* np.cumsum(prev_b_data) */
            double acc = 0;
            for (i64 k = 0; k < plen; k++) { acc = k == 0 ? pz[k] : acc + pz[k]; cum[k] = acc; }
        }
        if (b_s - ps - 1 < 0 || b_s - ps - 1 >= plen) { rc = ORC_INTERNAL; break; }
        bf[0] = bz[0] + pf[b_s - ps - 1];
        bl[0] = 1;
        for (i64 pos = b_s + 1; pos < pe + 1; pos++) {
            if (pos - b_s >= b_e - b_s) break; /* would index past the base upstream */
            i64 lag = 1;
            while (1) {
                i64 idx = pos - ps - lag;
                if (idx < 0) idx += plen; /* python wrap-around (boundscheck on) */
                if (idx < 0 || idx >= plen) { rc = ORC_INTERNAL; break; }
                if (pl[idx] + lag <= m) lag++;
                else break;
            }
            if (rc != ORC_OK) break;
            i64 di = pos - ps - lag;
            if (di < 0) di += plen;
            double diag = pf[di];
            if (lag > 1) diag += cum[pos - ps - 1] - cum[di];
            double stay = bf[pos - b_s - 1];
            double best;
            i64 dv;
            if (diag > stay) { best = diag; dv = 1; }
            else { best = stay; dv = bl[pos - b_s - 1] + 1; }
            bf[pos - b_s] = bz[pos - b_s] + best;
            bl[pos - b_s] = dv;
        }
        if (rc == ORC_OK && b_e > pe + 1) {
            double fv = bf[pe - b_s];
            i64 cl = bl[pe - b_s];
            for (i64 k = 0; k < b_e - pe - 1; k++) {
                fv += bz[k + pe - b_s + 1];
                cl += 1;
                bf[k + pe - b_s + 1] = fv;
                bl[k + pe - b_s + 1] = cl;
            }
        }
    }
    /* traceback */
    if (rc == ORC_OK) {
        i64 sig_start = be[n - 1] - 1;
        for (i64 b = n - 1; b >= 1 && rc == ORC_OK; b--) {
            const double *cf = fw + off[b], *nf = fw + off[b - 1];
            i64 cs = bs[b], ns = bs[b - 1], ne = be[b - 1];
            i64 cnt = 1, found = -1;
            for (i64 sp = sig_start; sp >= 0; sp--) {
                cnt += 1;
                if (cnt <= m || sp - 1 >= ne) continue;
                if (sp <= cs) { found = sp; break; }
                if (nf[sp - ns - 1] > cf[sp - cs - 1]) { found = sp; break; }
            }
            if (found < 0) { rc = ORC_INTERNAL; break; }
            new_segs[b - 1] = found;
            sig_start = found - 1;
        }
    }
    free(bs); free(be); free(off); free(z); free(fw); free(ld); free(cum);
    return rc;
}

/* ---- stand-alone forms of the raw-signal DP kernels (the Cython call signatures) -------- */

/* c_reg_z_scores, _c_dynamic_programming.pyx:34-97: admissible signal interval of every base
 * of [reg_start, reg_end) (absolute positions in r_sig).  The z-scores themselves are
 * orc_base_z_scores over r_sig[sig_starts[i]:sig_ends[i]]; the reference returns the bounds
 * relative to r_b_starts[reg_start]. */
void orc_reg_z_bounds(const i64 *r_b_starts, i64 reg_start, i64 reg_end, i64 max_base_shift,
                      i64 min_obs_per_base, i64 *sig_starts, i64 *sig_ends)
{
    const i64 reg_len = reg_end - reg_start;
    i64 prev = 0;
    for (i64 idx = 0; idx < reg_len; idx++) {
        const i64 base_i = reg_start + idx;
        i64 b = r_b_starts[imax(reg_start, base_i - max_base_shift)];
        if (idx > 0 && b < prev + min_obs_per_base) b = prev + min_obs_per_base;
        sig_starts[idx] = b;
        prev = b;
    }
    for (i64 idx = 0; idx < reg_len; idx++) {
        const i64 base_i = reg_start + (reg_len - idx - 1);
        i64 b = r_b_starts[imin(reg_end, base_i + max_base_shift + 1)];
        if (idx > 0 && b > prev - min_obs_per_base) b = prev - min_obs_per_base;
        sig_ends[reg_len - idx - 1] = b;
        prev = b;
    }
}

/* c_base_forward_pass, _c_dynamic_programming.pyx:99-163.  Negative indices wrap like the
 * (bounds-checked, wraparound) Cython buffer accesses; anything else out of range is the
 * IndexError the reference would raise (ORC_INTERNAL). */
int orc_base_forward_pass(const double *b_data, i64 b_start, i64 b_end, const double *prev_b_data,
                          i64 prev_b_start, i64 prev_b_end, const double *prev_b_fwd_data,
                          const i64 *prev_b_last_diag, i64 min_obs_per_base, double *b_fwd_data,
                          i64 *b_last_diag)
{
    const i64 b_len = b_end - b_start, plen = prev_b_end - prev_b_start;
    if (b_len <= 0 || plen <= 0) return ORC_INTERNAL;
    double *cum = (double *)malloc(sizeof(double) * (size_t)plen);
    { double acc = 0; for (i64 k = 0; k < plen; k++) { acc = k == 0 ? prev_b_data[k] : acc + prev_b_data[k]; cum[k] = acc; } }
    int rc = ORC_OK;
    i64 i0 = b_start - prev_b_start - 1;
    if (i0 < 0) i0 += plen;
    if (i0 < 0 || i0 >= plen) { free(cum); return ORC_INTERNAL; }
    b_fwd_data[0] = b_data[0] + prev_b_fwd_data[i0];
    b_last_diag[0] = 1;
    for (i64 pos = b_start + 1; pos < prev_b_end + 1 && rc == ORC_OK; pos++) {
        if (pos - b_start >= b_len) { rc = ORC_INTERNAL; break; }
        i64 lag = 1;
        while (1) {
            i64 idx = pos - prev_b_start - lag;
            if (idx < 0) idx += plen;
            if (idx < 0 || idx >= plen) { rc = ORC_INTERNAL; break; }
            if (prev_b_last_diag[idx] + lag <= min_obs_per_base) lag++;
            else break;
        }
        if (rc != ORC_OK) break;
        i64 di = pos - prev_b_start - lag;
        if (di < 0) di += plen;
        double diag = prev_b_fwd_data[di];
        if (lag > 1) diag += cum[pos - prev_b_start - 1] - cum[di];
        const double stay = b_fwd_data[pos - b_start - 1];
        double best;
        i64 dv;
        if (diag > stay) { best = diag; dv = 1; }
        else { best = stay; dv = b_last_diag[pos - b_start - 1] + 1; }
        b_fwd_data[pos - b_start] = b_data[pos - b_start] + best;
        b_last_diag[pos - b_start] = dv;
    }
    if (rc == ORC_OK && b_end > prev_b_end + 1) {
        i64 at = prev_b_end - b_start;
        if (at < 0) at += b_len;
        if (at < 0 || at >= b_len) rc = ORC_INTERNAL;
        else {
            double fv = b_fwd_data[at];
            i64 cl = b_last_diag[at];
            for (i64 k = 0; k < b_end - prev_b_end - 1; k++) {
                fv += b_data[k + prev_b_end - b_start + 1];
                cl += 1;
                b_fwd_data[k + prev_b_end - b_start + 1] = fv;
                b_last_diag[k + prev_b_end - b_start + 1] = cl;
            }
        }
    }
    free(cum);
    return rc;
}

/* c_base_traceback, _c_dynamic_programming.pyx:165-182; -1 where the reference falls off the
 * loop and returns None. */
i64 orc_base_traceback(const double *curr_b_data, i64 curr_start, const double *next_b_data,
                       i64 next_start, i64 next_end, i64 sig_start, i64 min_obs_per_base)
{
    i64 cnt = 1;
    for (i64 sp = sig_start; sp >= 0; sp--) {
        cnt += 1;
        if (cnt <= min_obs_per_base || sp - 1 >= next_end) continue;
        if (sp <= curr_start) return sp;
        if (next_b_data[sp - next_start - 1] > curr_b_data[sp - curr_start - 1]) return sp;
    }
    return -1;
}

typedef struct { i64 s, e; } win_t;

static i64 merge_windows(win_t *w, i64 n)
{
    i64 m = 0;
    for (i64 i = 0; i < n; i++) {
        if (m > 0 && w[i].s < w[m - 1].e) w[m - 1].e = w[i].e;
        else w[m++] = w[i];
    }
    return m;
}

static void trim_windows(win_t *w, i64 n, i64 n_segs)
{
    if (w[0].s < 0) w[0].s = 0;
    if (w[n - 1].e > n_segs - 1) w[n - 1].e = n_segs - 1;
}

static inline i64 seg_at(const i64 *segs, i64 n_segs, i64 i)
{
    return segs[i < 0 ? i + n_segs : i]; /* numpy negative index */
}

static int window_too_small(const i64 *segs, i64 n_segs, win_t w, i64 m, double extra_sig_factor)
{
    i64 n_events = w.e - w.s;
    if (w.e >= n_segs || w.s < -n_segs) return -1;
    i64 sig_len = seg_at(segs, n_segs, w.e) - seg_at(segs, n_segs, w.s);
    return (double)sig_len <= (double)((n_events + 1) * m) * extra_sig_factor;
}

/* rq.resolve_skipped_bases_with_raw, resquiggle.py:402-540, with its keyword arguments
 * del_fix_window / max_del_fix_window / extra_sig_factor (:405-407) */
int orc_resolve_skipped_bases_w(const i64 *dp_segs, i64 n_segs, const double *norm, i64 n_norm,
    const double *ref_means, const double *ref_sds, const orc_params *p, i64 max_raw_cpts,
    i64 del_fix_window, i64 max_del_fix_window, double extra_sig_factor, i64 *out_segs)
{
    i64 m = p->raw_min_obs_per_base;
    memcpy(out_segs, dp_segs, sizeof(i64) * (size_t)n_segs);
    win_t *w = (win_t *)malloc(sizeof(win_t) * (size_t)(n_segs + 1));
    i64 nw = 0;
    for (i64 d = 0; d + 1 < n_segs; d++) {
        if (dp_segs[d + 1] - dp_segs[d] != 0) continue;
        if (nw > 0 && d < w[nw - 1].e + del_fix_window) w[nw - 1].e = d + del_fix_window + 1;
        else { w[nw].s = d - del_fix_window; w[nw].e = d + del_fix_window + 1; nw++; }
    }
    if (nw == 0) { free(w); goto checks; }
    {
        int expanded = 0;
        nw = merge_windows(w, nw);
        trim_windows(w, nw, n_segs);
        for (i64 it = 0; it < max_del_fix_window - del_fix_window; it++) {
            expanded = 0;
            for (i64 i = 0; i < nw; i++) {
                int ts = window_too_small(dp_segs, n_segs, w[i], m, extra_sig_factor);
                if (ts < 0) { free(w); return ORC_INTERNAL; }
                if (ts) { expanded = 1; w[i].s -= 1; w[i].e += 1; }
            }
            if (!expanded) break;
            nw = merge_windows(w, nw);
            trim_windows(w, nw, n_segs);
        }
        if (expanded) {
            for (i64 i = 0; i < nw; i++) {
                int ts = window_too_small(dp_segs, n_segs, w[i], m, extra_sig_factor);
                if (ts < 0) { free(w); return ORC_INTERNAL; }
                if (ts) { free(w); return ORC_NOT_ENOUGH_DEL_SIGNAL; }
            }
        }
        if (max_raw_cpts >= 0) {
            i64 mx = 0;
            for (i64 i = 0; i < nw; i++) mx = imax(mx, w[i].e - w[i].s);
            if (mx > max_raw_cpts) { free(w); return ORC_TOO_MANY_DELS; }
        }
        i64 *ns = (i64 *)malloc(sizeof(i64) * (size_t)n_segs);
        for (i64 i = 0; i < nw; i++) {
            i64 s = w[i].s, e = w[i].e, n = e - s;
            if (s < 0 || e >= n_segs) { free(ns); free(w); return ORC_INTERNAL; }
            i64 sig_start = dp_segs[s], sig_end = dp_segs[e];
            if (sig_start < 0 || sig_end > n_norm) { free(ns); free(w); return ORC_INTERNAL; }
            int rc = orc_raw_window_dp(norm + sig_start, sig_end - sig_start, ref_means + s,
                                       ref_sds + s, n, m, (int)p->do_winsorize_z,
                                       p->max_half_z_score, ns);
            if (rc != ORC_OK) { free(ns); free(w); return rc; }
            for (i64 k = 0; k < n - 1; k++) out_segs[s + 1 + k] = ns[k] + sig_start;
        }
        free(ns);
        free(w);
    }
checks:
    for (i64 i = 0; i + 1 < n_segs; i++)
        if (out_segs[i + 1] - out_segs[i] < 1) return ORC_ZERO_LEN;
    if (out_segs[0] < 0) return ORC_NEG_START;
    if (out_segs[n_segs - 1] > n_norm) return ORC_PAST_END;
    return ORC_OK;
}

/* ... at the reference's defaults (_default_parameters.py:67,72,73) */
int orc_resolve_skipped_bases(const i64 *dp_segs, i64 n_segs, const double *norm, i64 n_norm,
    const double *ref_means, const double *ref_sds, const orc_params *p, i64 max_raw_cpts,
    i64 *out_segs)
{
    return orc_resolve_skipped_bases_w(dp_segs, n_segs, norm, n_norm, ref_means, ref_sds, p, max_raw_cpts,
                                       DEL_FIX_WINDOW, MAX_DEL_FIX_WINDOW, EXTRA_SIG_FACTOR, out_segs);
}

/* rq.segment_signal, resquiggle.py:1057-1120.  Returns n valid cpts via *n_cpts. */
static int orc_segment_signal(const double *raw, i64 n_raw, i64 num_events, const orc_params *p,
    const orc_opts *o, const i64 *stall, i64 n_stall, i64 *valid_cpts, i64 *n_cpts,
    double *norm, double *sv)
{
    int rc;
    if (p->use_t_test_seg) {
        rc = orc_valid_cpts_w_cap_t_test(raw, n_raw, p->min_obs_per_base, p->running_stat_width,
                                         num_events, valid_cpts);
        if (rc != ORC_OK) return rc;
        i64 n = num_events;
        if (stall) n = orc_remove_stall_cpts(stall, n_stall, valid_cpts, n);
        *n_cpts = n;
        if (o->has_scale_values)
            return orc_normalize_raw_signal(raw, n_raw, o, 1, o->sv_shift, o->sv_scale,
                                            (int)o->sv_has_lims, o->sv_lower, o->sv_upper, norm, sv);
        if (o->has_const_scale)
            return orc_normalize_raw_signal(raw, n_raw, o, 0, 0, 0, 0, 0, 0, norm, sv);
        if (o->use_rna_event_scale) {
            /* ts.get_scale_values_from_events, tombo_stats.py:217-233 */
            i64 ne = o->rna_scale_num_events;
            if ((double)n * o->rna_scale_max_frac_events < (double)ne)
                ne = (i64)((double)n * o->rna_scale_max_frac_events);
            if (ne > n) ne = n;
            if (ne < 2 || !o->has_outlier_thresh) return ORC_INTERNAL;
            double *em = (double *)malloc(sizeof(double) * (size_t)ne);
            orc_new_means(raw, valid_cpts, ne - 1, em);
            double med = orc_median(em, ne - 1);
            for (i64 i = 0; i < ne - 1; i++) em[i] = fabs(em[i] - med);
            double mad = orc_median(em, ne - 1);
            free(em);
            return orc_normalize_raw_signal(raw, n_raw, o, 1, med, mad, 1, -o->outlier_thresh,
                                            o->outlier_thresh, norm, sv);
        }
        /* scale_values=None: normalize_raw_signal(raw) with the default 'median' type and
         * no outlier threshold */
        {
            orc_opts o2 = *o;
            o2.has_outlier_thresh = 0;
            o2.has_const_scale = 0;
            return orc_normalize_raw_signal(raw, n_raw, &o2, 0, 0, 0, 0, 0, 0, norm, sv);
        }
    }
    if (o->has_scale_values)
        rc = orc_normalize_raw_signal(raw, n_raw, o, 1, o->sv_shift, o->sv_scale,
                                      (int)o->sv_has_lims, o->sv_lower, o->sv_upper, norm, sv);
    else
        rc = orc_normalize_raw_signal(raw, n_raw, o, 0, 0, 0, 0, 0, 0, norm, sv);
    if (rc != ORC_OK) return rc;
    rc = orc_valid_cpts_w_cap(norm, n_raw, p->min_obs_per_base, p->running_stat_width,
                              num_events, valid_cpts);
    if (rc != ORC_OK) return rc;
    i64 n = num_events;
    if (stall) n = orc_remove_stall_cpts(stall, n_stall, valid_cpts, n);
    *n_cpts = n;
    return ORC_OK;
}

/* rq.resquiggle_read, resquiggle.py:1122-1214 */
int orc_resquiggle_read(
    const double *raw, i64 n_raw, const uint8_t *seq_codes, i64 seq_len,
    const double *kmer_means, const double *kmer_sds,
    const orc_params *p, const orc_opts *o,
    const i64 *stall_ints, i64 n_stall, const i64 *samp_ind, i64 n_samp,
    i64 *segs, i64 *read_start_rel_to_raw, double *norm_signal, i64 *norm_len,
    double *scale_values, double *sig_match_score, i64 *norm_params_changed, orc_debug *dbg)
{
    if (raw == NULL) return ORC_NO_RAW;
    i64 K = o->kmer_width;
    i64 B = seq_len - K + 1; /* num_mapped_bases */
    /* ts.compute_num_events, tombo_stats.py:1558-1574 */
    i64 num_events = imax(n_raw / p->mean_obs_per_event,
                          (i64)((double)B * o->min_event_to_seq_ratio));
    if ((double)num_events / (double)p->bandwidth > (double)B) return ORC_TOO_MUCH_SIGNAL;
    if (B <= 0 || num_events <= 1) return ORC_INTERNAL;

    int rc;
    i64 *valid_cpts = (i64 *)malloc(sizeof(i64) * (size_t)num_events);
    double *norm = (double *)malloc(sizeof(double) * (size_t)n_raw);
    double sv[4];
    i64 n_cpts = 0;
    double *event_means = NULL, *ref_means = NULL, *ref_sds = NULL, *bm = NULL;
    i64 *dp_segs = NULL;
    rc = orc_segment_signal(raw, n_raw, num_events, p, o, stall_ints, n_stall, valid_cpts,
                            &n_cpts, norm, sv);
    if (rc != ORC_OK) goto done;
    if (dbg) {
        dbg->n_valid_cpts = n_cpts;
        if (dbg->valid_cpts) memcpy(dbg->valid_cpts, valid_cpts, sizeof(i64) * (size_t)n_cpts);
        if (dbg->seg_norm_signal) memcpy(dbg->seg_norm_signal, norm, sizeof(double) * (size_t)n_raw);
        memcpy(dbg->seg_scale_values, sv, sizeof(sv));
    }
    i64 n_ev = n_cpts - 1;
    if (n_ev < 1) { rc = ORC_INTERNAL; goto done; }
    event_means = (double *)malloc(sizeof(double) * (size_t)n_ev);
    orc_new_means(norm, valid_cpts, n_ev, event_means);
    if (dbg && dbg->event_means) memcpy(dbg->event_means, event_means, sizeof(double) * (size_t)n_ev);

    /* std_ref.get_exp_levels_from_seq, tombo_stats.py:834-862 */
    ref_means = (double *)malloc(sizeof(double) * (size_t)B);
    ref_sds = (double *)malloc(sizeof(double) * (size_t)B);
    for (i64 i = 0; i < B; i++) {
        i64 code = 0;
        for (i64 j = 0; j < K; j++) {
            if (seq_codes[i + j] > 3) { rc = ORC_INVALID_SEQ; goto done; }
            code = code * 4 + seq_codes[i + j];
        }
        ref_means[i] = kmer_means[code];
        ref_sds[i] = kmer_sds[code];
    }
    /* genome_seq[central_pos:-dnstrm_bases] has B bases when dnstrm_bases > 0 */
    if (K - o->central_pos - 1 <= 0) { rc = ORC_DISCORDANT; goto done; }

    dp_segs = (i64 *)malloc(sizeof(i64) * (size_t)(B + 1));
    i64 read_start = 0;
    rc = orc_find_adaptive_base_assignment(valid_cpts, event_means, n_ev, ref_means, ref_sds, B,
                                           p, o, dp_segs, &read_start, dbg);
    if (rc != ORC_OK) goto done;
    if (dbg) {
        dbg->dp_read_start = read_start;
        if (dbg->dp_segs) memcpy(dbg->dp_segs, dp_segs, sizeof(i64) * (size_t)(B + 1));
    }
    /* norm_signal[read_start : read_start + segs[-1]] */
    i64 nl = dp_segs[B];
    if (read_start < 0 || read_start + nl > n_raw || nl < 0) { rc = ORC_INTERNAL; goto done; }
    memmove(norm, norm + read_start, sizeof(double) * (size_t)nl);

    rc = orc_resolve_skipped_bases(dp_segs, B + 1, norm, nl, ref_means, ref_sds, p,
                                   o->max_raw_cpts, segs);
    if (rc != ORC_OK) goto done;

    bm = (double *)malloc(sizeof(double) * (size_t)B);
    int changed = 0;
    if (!o->skip_seq_scaling) {
        orc_new_means(norm, segs, B, bm);
        double ts4[4];
        if (B > 1000) { /* MAX_POINTS_FOR_THEIL_SEN, tombo_stats.py:411-416 */
            if (n_samp != 1000 || samp_ind == NULL) { rc = ORC_INTERNAL; goto done; }
            double ev[1000], md[1000];
            for (i64 i = 0; i < 1000; i++) {
                if (samp_ind[i] < 0 || samp_ind[i] >= B) { rc = ORC_INTERNAL; goto done; }
                ev[i] = bm[samp_ind[i]];
                md[i] = ref_means[samp_ind[i]];
            }
            rc = orc_theil_sen(ev, md, 1000, sv[0], sv[1], ts4);
        } else {
            rc = orc_theil_sen(bm, ref_means, B, sv[0], sv[1], ts4);
        }
        if (rc != ORC_OK) goto done;
        if (dbg) memcpy(dbg->theil_sen, ts4, sizeof(ts4));
        sv[0] = ts4[0];
        sv[1] = ts4[1];
        for (i64 i = 0; i < nl; i++) norm[i] = (norm[i] - ts4[2]) / ts4[3];
        changed = fabs(ts4[2]) > SHIFT_CHANGE_THRESH || fabs(ts4[3] - 1) > SCALE_CHANGE_THRESH;
    }
    /* ts.get_read_seg_score, tombo_stats.py:2327-2338 */
    orc_new_means(norm, segs, B, bm);
    for (i64 i = 0; i < B; i++) bm[i] = fabs((bm[i] - ref_means[i]) / ref_sds[i]);
    *sig_match_score = orc_np_sum(bm, B) / (double)B;
    *read_start_rel_to_raw = read_start;
    memcpy(norm_signal, norm, sizeof(double) * (size_t)nl);
    *norm_len = nl;
    memcpy(scale_values, sv, sizeof(sv));
    *norm_params_changed = changed;
    rc = ORC_OK;
done:
    free(valid_cpts); free(norm); free(event_means); free(ref_means); free(ref_sds);
    free(dp_segs); free(bm);
    return rc;
}
