/* oracle/tombo_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C, single thread, IEEE double in source order, built with
 * -O2 -ffp-contract=off) of the reference resquiggle hot path
 * (nanoporetech/tombo v1.5.1: tombo/resquiggle.py:345-1214, tombo/_c_dynamic_programming.pyx,
 * tombo/_c_helper.pyx, parts of tombo/tombo_stats.py).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library, and only as the checker.
 *
 * Parity status: PINNED -- checked bit-for-bit against tests/golden/ (npz files), which were generated
 * from the live reference in the build container (tests/golden/gen_golden.py).
 */
#ifndef TOMBO_ORACLE_H
#define TOMBO_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t i64;

/* status codes; tombo_amd/errors.py maps them to the reference's TomboError strings */
enum {
    ORC_OK = 0,
    ORC_TOO_MUCH_SIGNAL = 1,       /* resquiggle.py:1160 */
    ORC_FEWER_CPTS = 2,            /* _c_helper.pyx:118,200 */
    ORC_READ_TOO_SHORT_START = 3,  /* resquiggle.py:704 */
    ORC_MAP_TOO_SHORT_START = 4,   /* resquiggle.py:706 */
    ORC_POOR_START = 5,            /* resquiggle.py:745 */
    ORC_INVALID_START_PATH = 6,    /* tombo_stats.py:2356 */
    ORC_OPEN_PORE = 7,             /* resquiggle.py:1009 */
    ORC_STARTS_TOO_FAR = 8,        /* resquiggle.py:612 */
    ORC_MASK_TOO_FEW = 9,          /* resquiggle.py:672 */
    ORC_ADAPT_BEYOND = 10,         /* _c_dynamic_programming.pyx:354 */
    ORC_BEYOND_BANDWIDTH = 11,     /* _c_dynamic_programming.pyx:305 */
    ORC_DISCORDANT = 12,           /* resquiggle.py:976 */
    ORC_NOT_ENOUGH_DEL_SIGNAL = 13,/* resquiggle.py:490 */
    ORC_TOO_MANY_DELS = 14,        /* resquiggle.py:495 */
    ORC_INVALID_SEG = 15,          /* resquiggle.py:530 */
    ORC_ZERO_LEN = 16,             /* resquiggle.py:534 */
    ORC_NEG_START = 17,            /* resquiggle.py:536 */
    ORC_PAST_END = 18,             /* resquiggle.py:538 */
    ORC_RESCALE_FAIL = 19,         /* tombo_stats.py:421 */
    ORC_SEQ_SEG_MISMATCH = 20,     /* resquiggle.py:1201 */
    ORC_NO_RAW = 21,               /* resquiggle.py:1148 */
    ORC_INVALID_SEQ = 22,          /* tombo_stats.py:858 */
    ORC_INTERNAL = 100             /* the reference would raise a non-Tombo exception */
};

/* th.resquiggleParams (tombo_helper.py:173-198) */
typedef struct {
    double match_evalue, skip_pen, max_half_z_score, z_shift, stay_pen;
    i64 bandwidth, running_stat_width, min_obs_per_base, raw_min_obs_per_base,
        mean_obs_per_event, use_t_test_seg, band_bound_thresh, start_bw, start_save_bw,
        start_n_bases;
    i64 do_winsorize_z; /* max_half_z_score is not None */
} orc_params;

/* the remaining arguments of resquiggle_read (resquiggle.py:1122-1127) + map_res fields */
typedef struct {
    i64 has_outlier_thresh; double outlier_thresh;
    i64 has_const_scale;    double const_scale;
    i64 has_scale_values;   double sv_shift, sv_scale;
    i64 sv_has_lims;        double sv_lower, sv_upper;
    i64 skip_seq_scaling;
    i64 check_start_score;  double sig_match_thresh; /* seq_samp_type given */
    i64 max_raw_cpts;       /* < 0: None */
    double min_event_to_seq_ratio;
    i64 kmer_width, central_pos;
    i64 use_rna_event_scale; i64 rna_scale_num_events; double rna_scale_max_frac_events;
} orc_opts;

/* optional stage-wise outputs (any pointer may be NULL; capacities are the caller's job) */
typedef struct {
    i64 *valid_cpts; i64 n_valid_cpts;
    double *event_means;
    double *seg_norm_signal; double seg_scale_values[4];
    double start_calls[4]; i64 n_start_calls; /* (loc, events_per_base) x up to 2 */
    i64 *band_event_starts;
    double *fwd_last_row; i64 fwd_last_row_len;
    i64 *read_tb;
    i64 *dp_segs; i64 dp_read_start;
    double theil_sen[4];
    i64 used_static, mask_seq_len;
} orc_debug;

int orc_resquiggle_read(
    const double *raw, i64 n_raw, const uint8_t *seq_codes, i64 seq_len,
    const double *kmer_means, const double *kmer_sds,
    const orc_params *p, const orc_opts *o,
    const i64 *stall_ints, i64 n_stall,          /* [n_stall][2] or NULL */
    const i64 *samp_ind, i64 n_samp,            /* Theil-Sen subsample (np.random.choice) */
    i64 *segs, i64 *read_start_rel_to_raw,      /* out: [B+1], scalar */
    double *norm_signal, i64 *norm_len,         /* out: cap n_raw */
    double *scale_values,                       /* out: shift, scale, lower, upper */
    double *sig_match_score, i64 *norm_params_changed, orc_debug *dbg);

/* kernel-level restatements (same semantics as the Cython functions they cite) */
void orc_base_z_scores(const double *sig, i64 n, double mean, double sd, int winsor,
                       double max_half_z, double *out);
void orc_banded_forward_pass(const double *z, i64 n_bases, i64 bw, const i64 *event_starts,
                             double skip_pen, double stay_pen, double *fwd, int8_t *tb);
int orc_adaptive_banded_forward_pass(double *fwd, int8_t *tb, i64 n_bases, i64 bw,
    i64 *event_starts, const double *event_means, i64 n_events, const double *ref_means,
    const double *ref_sds, double z_shift, double skip_pen, double stay_pen, i64 start_seq_pos,
    double mask_fill_z, int winsor, double max_half_z);
int orc_banded_traceback(const int8_t *tb, i64 n_bases, i64 bw, const i64 *event_starts,
                         i64 band_pos, i64 band_boundary_thresh, i64 *seq_poss);
int orc_valid_cpts_w_cap(const double *sig, i64 n, i64 min_base_obs, i64 width, i64 num_cpts,
                         i64 *cpts);
int orc_valid_cpts_w_cap_t_test(const double *sig, i64 n, i64 min_base_obs, i64 width,
                                i64 num_cpts, i64 *cpts);
void orc_new_means(const double *sig, const i64 *segs, i64 n_segs, double *means);
void orc_new_mean_stds(const double *sig, const i64 *segs, i64 n_segs, double *means,
                       double *stds);
void orc_apply_outlier_thresh(const double *sig, i64 n, double lo, double hi, double *out);
void orc_compute_slopes(const double *ev, const double *model, i64 n, double max_slope,
                        double *slopes);
void orc_reg_z_bounds(const i64 *r_b_starts, i64 reg_start, i64 reg_end, i64 max_base_shift,
    i64 min_obs_per_base, i64 *sig_starts, i64 *sig_ends);
int orc_base_forward_pass(const double *b_data, i64 b_start, i64 b_end, const double *prev_b_data,
    i64 prev_b_start, i64 prev_b_end, const double *prev_b_fwd_data,
    const i64 *prev_b_last_diag, i64 min_obs_per_base, double *b_fwd_data, i64 *b_last_diag);
i64 orc_base_traceback(const double *curr_b_data, i64 curr_start, const double *next_b_data,
    i64 next_start, i64 next_end, i64 sig_start, i64 min_obs_per_base);
double orc_calc_llh_ratio(const double *means, const double *ref_means, const double *alt_means,
    const double *ref_vars, const double *alt_vars, i64 n);
double orc_calc_llh_ratio_const_var(const double *means, const double *ref_means,
    const double *alt_means, i64 n, double const_var);
double orc_calc_scaled_llh_ratio_const_var(const double *means, const double *ref_means,
    const double *alt_means, i64 n, double const_var, double scale_factor,
    double density_height_factor, double density_height_power);
double orc_median(const double *x, i64 n);
double orc_np_sum(const double *a, i64 n);
void orc_linspace(double start, double stop, i64 num, double *out);
int orc_normalize_raw_signal(const double *raw, i64 n, const orc_opts *o, int use_sv,
                             double sv_shift, double sv_scale, int sv_has_lims, double sv_lo,
                             double sv_hi, double *norm, double *sv_out);
int orc_resolve_skipped_bases_w(const i64 *dp_segs, i64 n_segs, const double *norm, i64 n_norm,
    const double *ref_means, const double *ref_sds, const orc_params *p, i64 max_raw_cpts,
    i64 del_fix_window, i64 max_del_fix_window, double extra_sig_factor, i64 *out_segs);
int orc_resolve_skipped_bases(const i64 *dp_segs, i64 n_segs, const double *norm, i64 n_norm,
    const double *ref_means, const double *ref_sds, const orc_params *p, i64 max_raw_cpts,
    i64 *out_segs);
int orc_theil_sen(const double *ev, const double *model, i64 n, double prev_shift,
                  double prev_scale, double *out4);

#ifdef __cplusplus
}
#endif
#endif
