/* include/tombo_amd.h -- C ABI of the MI355X-native resquiggle engine (libtombo_amd.so).
 *
 * Drop-in boundary for the hot path of nanoporetech/tombo v1.5.1: everything below
 * tombo.resquiggle.resquiggle_read() (tombo/resquiggle.py:1122-1214) -- i.e. the thirteen
 * Python-callable Cython kernels of tombo/_c_dynamic_programming.pyx and tombo/_c_helper.pyx
 * plus the numpy glue between them -- runs as HIP kernels for gfx950 behind these entry points.
 * Plain pointers and sizes only; caller owns every host buffer; no exceptions cross the ABI:
 * every per-read failure is a status code (TBA_* below == the reference's TomboError strings,
 * table in tombo_amd/errors.py).  All arithmetic on the path is IEEE float64 / int64 in the
 * reference's operation order (device code is built with -ffp-contract=off).
 *
 * Two levels:
 *   1. the batch engine (tba_engine_*, tba_batch_*): N reads packed as ragged SoA buffers, one
 *      fixed kernel sequence per batch on one HIP stream; this is what resquiggle_read /
 *      resquiggle_batch call and what bench.py times (inputs resident in HBM).  The same
 *      sequence can be run stage by stage with the caller's intermediates injected
 *      (tba_batch_run_stages / tba_batch_put: the reference's public per-stage functions,
 *      resquiggle.py:63-67), and tba_batch_base_stats adds the per-base statistics of the
 *      Events table (tombo_helper.py:2341-2362).
 *   2. per-kernel entry points (tba_c_*): one call == one call of the Cython function it
 *      cites, same argument meaning, host pointers in / host pointers out, executed by the same
 *      device code (batch of one).  These are what a maintainer binds in place of the Cython
 *      modules (INTEGRATION.md shows the ctypes stubs).
 */
#ifndef TOMBO_AMD_H
#define TOMBO_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (per read) ------------------------------------------------------------ */
enum {
    TBA_OK = 0,
    TBA_TOO_MUCH_SIGNAL = 1,        /* resquiggle.py:1160 */
    TBA_FEWER_CPTS = 2,             /* _c_helper.pyx:118,200 */
    TBA_READ_TOO_SHORT_START = 3,   /* resquiggle.py:704 */
    TBA_MAP_TOO_SHORT_START = 4,    /* resquiggle.py:706 */
    TBA_POOR_START = 5,             /* resquiggle.py:745 */
    TBA_INVALID_START_PATH = 6,     /* tombo_stats.py:2356 */
    TBA_OPEN_PORE = 7,              /* resquiggle.py:1009 */
    TBA_STARTS_TOO_FAR = 8,         /* resquiggle.py:612 */
    TBA_MASK_TOO_FEW = 9,           /* resquiggle.py:672 */
    TBA_ADAPT_BEYOND = 10,          /* _c_dynamic_programming.pyx:354 */
    TBA_BEYOND_BANDWIDTH = 11,      /* _c_dynamic_programming.pyx:305 */
    TBA_DISCORDANT = 12,            /* resquiggle.py:976 */
    TBA_NOT_ENOUGH_DEL_SIGNAL = 13, /* resquiggle.py:490 */
    TBA_TOO_MANY_DELS = 14,         /* resquiggle.py:495 */
    TBA_INVALID_SEG = 15,           /* resquiggle.py:530 */
    TBA_ZERO_LEN = 16,              /* resquiggle.py:534 */
    TBA_NEG_START = 17,             /* resquiggle.py:536 */
    TBA_PAST_END = 18,              /* resquiggle.py:538 */
    TBA_RESCALE_FAIL = 19,          /* tombo_stats.py:421 */
    TBA_SEQ_SEG_MISMATCH = 20,      /* resquiggle.py:1201 */
    TBA_NO_RAW = 21,                /* resquiggle.py:1148 */
    TBA_INVALID_SEQ = 22,           /* tombo_stats.py:858 */
    TBA_INTERNAL = 100,             /* the reference would raise a non-Tombo exception here */
    TBA_UNSUPPORTED = 101           /* valid in the reference, outside this engine's limits
                                       (band wider than TBA_MAX_BAND, scratch arena exhausted) */
};
/* call-level return values of the functions below (0 == success) */
enum { TBA_E_ARG = -1, TBA_E_HIP = -2, TBA_E_NOMEM = -3, TBA_E_STATE = -4 };

#define TBA_MAX_BAND 3072 /* widest DP band (cells per row) the wave-per-read kernel carries */
#define TBA_MAX_BATCH_READS 65535 /* reads per batch (a launch-grid dimension); longer lists are
                                     cut into batches by the host (tombo_amd/planner.py) */

/* th.resquiggleParams (tombo/tombo_helper.py:173-198) */
typedef struct {
    double match_evalue, skip_pen, max_half_z_score, z_shift, stay_pen;
    int64_t bandwidth, running_stat_width, min_obs_per_base, raw_min_obs_per_base,
        mean_obs_per_event, use_t_test_seg, band_bound_thresh, start_bw, start_save_bw,
        start_n_bases;
    int64_t do_winsorize_z; /* max_half_z_score is not None */
} tba_params;

/* remaining arguments of resquiggle_read (resquiggle.py:1122-1127) that apply to a whole batch */
typedef struct {
    int64_t has_outlier_thresh; double outlier_thresh;
    int64_t has_const_scale;    double const_scale;
    int64_t skip_seq_scaling;
    int64_t check_start_score;  double sig_match_thresh; /* seq_samp_type given */
    int64_t max_raw_cpts;                                 /* < 0: None */
    double  min_event_to_seq_ratio;
    int64_t use_rna_event_scale, rna_scale_num_events;    /* _default_parameters.py:78-80 */
    double  rna_scale_max_frac_events;
    int64_t skip_norm_out; /* engine option: do not materialise the final normalised signal (the
                              caller recomputes it from raw + scale_values, as Tombo does when it
                              reads a resquiggled FAST5 back: tombo_helper.py get_raw_read_slot /
                              tombo_stats.normalize_raw_signal); norm_signal downloads are refused */
    /* ---- the worker's per-read preparation (_resquiggle_worker.adjust_map_res,
     * resquiggle.py:1506-1530), on the device, part of the upload / of tba_batch_enqueue ---- */
    int64_t reverse_raw;   /* the samples arrive in acquisition order (direct RNA: 3'->5') and are
                              flipped in place once after the upload: raw_signal[::-1], :1516 */
    int64_t detect_stalls; /* ts.identify_stalls(raw_signal, MEAN_STALL_PARAMS) (tombo_stats.py:
                              269-368, the running-window-mean method; :1524-1528) over the
                              (flipped) raw samples; its intervals take the place of stall_ints,
                              which must then be NULL */
    int64_t stall_window_size, stall_n_windows, stall_mini_window_size,
        stall_min_consecutive_obs, stall_edge_buffer; /* th.stallParams, tombo_helper.py:200-214 */
    double  stall_threshold;
    int64_t device_subsample; /* draw the Theil-Sen subsample of reads with more than 1000 bases
                              (np.random.choice(B, 1000, replace=False), tombo_stats.py:411-416)
                              on the device instead of taking samp_ind: the first 1000 images of
                              a keyed pseudo-random permutation of [0, B) (counter based: a
                              function of subsample_seed and the read's index in the batch) */
    uint64_t subsample_seed;
    int64_t subsample_first_read; /* index, in the caller's job, of the batch's first read: read i of the batch
                              draws under the key of (subsample_seed, subsample_first_read + i), so the subsample
                              of a read does not depend on how its list was cut into batches */
    /* keyword arguments of resolve_skipped_bases_with_raw (resquiggle.py:405-407): bases a window
     * around a skipped base starts with / may grow to, and the signal a window must hold relative to
     * its bases.  All three zero (a zero-initialised struct): the reference's defaults 2 / 10 / 1.1
     * (_default_parameters.py:72,73,67). */
    int64_t del_fix_window, max_del_fix_window;
    double  extra_sig_factor;
} tba_opts;

typedef struct tba_engine tba_engine;

/* ---- engine ----------------------------------------------------------------------------- */
/* One engine per process per GPU (reads shard across GPUs by process; no collective).
 * device: HIP ordinal.  Fails with TBA_E_HIP when no gfx950 device is usable: there is no CPU
 * fallback in this library. */
int  tba_engine_create(int device, tba_engine **out);
void tba_engine_destroy(tba_engine *e);
const char *tba_last_error(void);
int  tba_device_count(void);
/* free / total bytes of the engine's device (hipMemGetInfo): what a batch planner budgets against */
int  tba_device_mem(tba_engine *e, int64_t *free_bytes, int64_t *total_bytes);

/* device bytes held by this engine's (grow-only) batch buffers */
int  tba_engine_held_bytes(tba_engine *e, int64_t *bytes);
/* Scheduling hint, no effect on results: n_engines = how many engines are fed concurrently on this
 * engine's device (the slots of a streaming pipeline; 1 = this engine has the device to itself, the
 * default).  With n_engines > 1 a batch whose work is mostly outside the banded DP (RNA) runs the
 * register-capped build of the DP kernel so that the other engines' kernels fit beside it. */
int  tba_engine_set_sharing(tba_engine *e, int n_engines);
/* (ABI 9: tba_engine_set_dp_workgroup_batch is gone with the workgroup-per-read form of the main forward pass it
 * switched on -- measured slower in round 4, profiles/r04_dp_workgroup_form.txt; TBA_GET_DP_WORKGROUP reads zeros.) */
/* Event detection (c_valid_cpts_w_cap, _c_helper.pyx:89-120) and the main traceback
 * (c_banded_traceback, _c_dynamic_programming.pyx:281-310) each have a LATENCY and a THROUGHPUT
 * form with identical results; which one a batch takes depends on its read count alone:
 *   - DNA event detection: batches of at most small_batch_reads reads scan a workgroup per read
 *     and keep the score array (k_cumsum_scores_long + k_peaks); larger ones run the score-free
 *     pipeline k_detect + k_pick (with the last pass of the normalisation in its loader);
 *   - traceback: batches of at most tb_wave_below reads give every read a wavefront
 *     (k_main_tb_par<64>), larger ones 16 lanes (k_main_tb_par<16>).
 * Both thresholds default to 1 024 reads; a negative argument leaves a threshold as it is, 0 sends
 * every batch through the throughput form.  Applied from the next tba_batch_enqueue /
 * tba_batch_run_stages.  The parity tests run through both forms by this entry; the environment
 * variables TBA_SMALL_BATCH_READS / TBA_TB_WAVE_BELOW set the same at engine creation.
 * TBA_GET_ED_FORM / TBA_GET_TB_FORM report which kernels actually produced each read's result. */
int  tba_engine_set_dispatch(tba_engine *e, int64_t small_batch_reads, int64_t tb_wave_below);
int  tba_engine_get_dispatch(tba_engine *e, int64_t *small_batch_reads, int64_t *tb_wave_below);
/* The side stream.  A full run (tba_batch_enqueue / tba_batch_run) puts what does not depend on the
 * signal's normalisation -- the stall detector (ts.identify_stalls, RNA) and the expected levels of the
 * sequences (get_exp_levels_from_seq) -- on a second stream of the engine, beside normalisation and
 * event detection, and joins it where the reference's order needs the results; identical results (an
 * invalid base still loses to an earlier segmentation error of the same read).  mode -1 (default):
 * used while at most two engines are alive on the device in this process (TBA_SIDE_STREAM_MAX_ENGINES)
 * and tba_engine_set_sharing says no more -- more streams than hardware queues serialise behind each
 * other; 0: never; 1: always.  tba_engine_last_side_stream: whether the last full run used it. */
int  tba_engine_set_side_stream(tba_engine *e, int mode);
int  tba_engine_last_side_stream(tba_engine *e);
/* TBA_ED_FORM_* of the last tba_c_valid_cpts_w_cap / tba_c_valid_cpts_w_cap_t_test call on this engine
 * (those entries follow the engine's dispatch like a batch of one read) */
int  tba_c_last_ed_form(tba_engine *e);

/* canonical k-mer level table, lexicographic k-mer order (TomboModel, tombo_stats.py:580-919;
 * lookup replaces get_exp_levels_from_seq :834-862) */
int tba_set_model(tba_engine *e, const double *kmer_means, const double *kmer_sds,
                  int64_t kmer_width, int64_t central_pos);

/* ---- host memory for streaming -----------------------------------------------------------
 * Page-locked host buffers: uploads from / downloads into them are true DMA transfers that
 * overlap with kernels of other engines (one engine == one batch slot == one HIP stream; a
 * streaming caller keeps 2-3 engines per GPU busy: upload N+1 || compute N || download N-1,
 * the analogue of the reference's reader -> worker -> writer processes,
 * resquiggle.py:1859-1950).  Pageable buffers work everywhere too, but make the "async" calls
 * block while HIP stages them. */
int tba_pinned_alloc(int64_t bytes, void **out);
int tba_pinned_free(void *p);

/* raw sample types accepted at the boundary: float64 (the reference's in-memory type), float32,
 * or the int16 DAC values as the FAST5 file stores them (`Signal`, resquiggle.py:1397).  Widening
 * to float64 happens on the device and is exact: results are bit-identical for the same values. */
enum { TBA_RAW_F64 = 0, TBA_RAW_F32 = 1, TBA_RAW_I16 = 2 };

/* ---- batch pipeline == resquiggle_read() for n_reads reads -------------------------------
 * raw_off[n+1], seq_off[n+1]: CSR offsets into raw (float64 pA / DAC values) and seq (uint8
 * codes 0..3 = ACGT, genome_seq including the k-mer flanks).
 * Optional per-read inputs (NULL to omit):
 *   sv_in[n][4] + sv_flags[n]: map_res.scale_values (shift, scale, lower, upper); flag bit0 =
 *       scale values given, bit1 = limits given (second and later run_rsqgl_iters passes,
 *       resquiggle.py:1498-1503);
 *   samp_ind[n][1000]: the np.random.choice subsample of calc_kmer_fitted_shift_scale
 *       (tombo_stats.py:411-416); required for reads with more than 1000 bases unless
 *       skip_seq_scaling;
 *   stall_off[n+1], stall_ints[][2]: map_res.stall_ints (RNA).
 */
int tba_batch_upload(tba_engine *e, const tba_params *p, const tba_opts *o, int64_t n_reads,
                     const double *raw, const int64_t *raw_off,
                     const uint8_t *seq, const int64_t *seq_off,
                     const double *sv_in, const int32_t *sv_flags,
                     const int64_t *samp_ind,
                     const int64_t *stall_ints, const int64_t *stall_off);
/* Same with a typed raw buffer (TBA_RAW_*), enqueue only: every copy is issued on the engine's
 * stream and the call returns; the caller's buffers (raw, seq, sv_in, samp_ind, stall_ints) must
 * stay valid and unchanged until tba_batch_sync / tba_batch_query reports the engine idle.
 * tba_batch_upload == this with TBA_RAW_F64 followed by tba_batch_sync.
 * raw and seq may also be device memory of the engine's GPU (a batch made by tba_synth_generate):
 * the copy kind is taken from the pointers. */
int tba_batch_upload_async(tba_engine *e, const tba_params *p, const tba_opts *o, int64_t n_reads,
                           const void *raw, int raw_dtype, const int64_t *raw_off,
                           const uint8_t *seq, const int64_t *seq_off,
                           const double *sv_in, const int32_t *sv_flags,
                           const int64_t *samp_ind,
                           const int64_t *stall_ints, const int64_t *stall_off);
/* device bytes a batch of these reads would occupy (what tba_batch_upload* allocates): lets a
 * host planner cut a read list into batches that fit a memory budget before uploading */
int tba_batch_footprint(const tba_params *p, const tba_opts *o, int64_t kmer_width, int raw_dtype,
                        int64_t n_reads, const int64_t *n_raw, const int64_t *seq_len,
                        double *bytes);
/* runs the whole kernel sequence on the uploaded batch; returns after the stream is idle */
int tba_batch_run(tba_engine *e);
/* same, but only enqueues (for timing with events / overlapping); pair with tba_batch_sync */
int tba_batch_enqueue(tba_engine *e);
int tba_batch_sync(tba_engine *e);
/* the kernel sequence enqueued NEXT on e starts after the one last enqueued on `other` has
 * finished (copies are not ordered): a streaming caller keeps the batches' kernels back to back
 * instead of interleaved while uploads and downloads of other slots overlap them */
int tba_batch_wait_for(tba_engine *e, tba_engine *other);
/* 0: the engine's stream is idle; 1: work still in flight (never blocks) */
int tba_batch_query(tba_engine *e);
/* Outputs (any pointer may be NULL):
 *   status[n]; segs (CSR by seg_off[i] = seq_off[i] - i*(K-1) + i, B_i+1 entries per read);
 *   read_start_rel_to_raw[n]; norm_signal (same CSR as raw; first norm_len[i] entries valid);
 *   scale_values[n][4]; sig_match_score[n]; norm_params_changed[n] */
int tba_batch_download(tba_engine *e, int32_t *status, int64_t *segs,
                       int64_t *read_start_rel_to_raw, double *norm_signal, int64_t *norm_len,
                       double *scale_values, double *sig_match_score,
                       int32_t *norm_params_changed);
/* Compact per-read record + enqueue-only download: results[n] (64 bytes per read: what
 * resquiggle_read returns besides the arrays), the base boundaries as int32 (segs32) and / or
 * int64 (segs64), CSR by seg_off like tba_batch_download, and the normalised signal (refused
 * under tba_opts.skip_norm_out).  The records and int32 boundaries are packed by a kernel on the
 * engine's stream, the copies follow on the same stream; the outputs are valid after
 * tba_batch_sync.  lower_lim / upper_lim are NaN where scale_values has None. */
typedef struct {
    int32_t status, norm_params_changed;
    int64_t read_start_rel_to_raw, norm_len;
    double shift, scale, lower_lim, upper_lim, sig_match_score;
} tba_read_result;
int tba_batch_download_async(tba_engine *e, tba_read_result *results, int32_t *segs32,
                             int64_t *segs64, double *norm_signal);
/* stage-wise intermediates of the last run, for parity tests (what: TBA_GET_*; CSR layouts in
 * tombo_amd/_native.py) */
enum {
    TBA_GET_VALID_CPTS = 1,   /* int64, CSR by ev_off (capacity num_events per read) */
    TBA_GET_N_CPTS = 2,       /* int64[n] */
    TBA_GET_EVENT_MEANS = 3,  /* float64, CSR by ev_off */
    TBA_GET_SEG_NORM = 4,     /* float64, CSR by raw_off: normalised signal before trimming */
    TBA_GET_SEG_SV = 5,       /* float64[n][4] */
    TBA_GET_START = 6,        /* float64[n][4]: (loc, events_per_base) of call 0 and call 1 */
    TBA_GET_BAND_STARTS = 7,  /* int64, CSR by ref_off (B per read) */
    TBA_GET_READ_TB = 8,      /* int64, CSR by seg_off */
    TBA_GET_DP_SEGS = 9,      /* int64, CSR by seg_off */
    TBA_GET_THEIL_SEN = 10,   /* float64[n][4] */
    TBA_GET_PATH = 11,        /* int32[n][4]: path (1 adaptive, 2 static), n_static, W, n_start_calls */
    TBA_GET_LAST_ROW = 12,    /* float64[n][TBA_MAX_BAND]: last forward-pass row */
    TBA_GET_DP_READ_START = 13, /* int64[n] */
    TBA_GET_KERNEL_MS = 14,   /* float32[32]: per-stage GPU time of the last run (events) */
    TBA_GET_REF_MEANS = 15,   /* float64, CSR by ref_off: expected levels of every base */
    TBA_GET_REF_SDS = 16,
    TBA_GET_SEGS = 17,        /* int64, CSR by seg_off: boundaries after skipped-base resolution */
    TBA_GET_STATUS = 18,      /* int32[n] */
    TBA_GET_START_FAIL = 19,  /* int32[n]: status that failed the first start-discovery try (0: none) */
    TBA_GET_STALL_INTS = 20,  /* int64[][2]: the stall intervals in force (given, or detected under
                                 tba_opts.detect_stalls), read i at STALL_OFF[i], N_STALL[i] of them */
    TBA_GET_N_STALL = 21,     /* int64[n] */
    TBA_GET_STALL_OFF = 22,   /* int64[n] */
    TBA_GET_SAMP_IND = 23,    /* int64[n][1000]: the Theil-Sen subsamples used (as uploaded, or as
                                 drawn under tba_opts.device_subsample; reads of <= 1000 bases: unused) */
    TBA_GET_TB_PARALLEL = 24, /* int32[n]: 1 where the chunk-parallel traceback walked the read (performance
                               * diagnostics: 0 on an adaptive read means the lane-per-read walk had to) */
    TBA_GET_ED_FUSED = 25,    /* int32[n]: 1 where the score-free event detection (k_detect / k_pick) finished the
                               * read, 0 where the kernels that keep the score array had to (diagnostics) */
    TBA_GET_ED_TAKEN_POS = 26, /* int32, two slots per sample (CSR by 2 * raw_off): positions of the taken list k_detect
                                * left, valid right after stage TBA_STAGE_SEGMENT only (later stages reuse the buffer) */
    TBA_GET_ED_N_TAKEN = 27,  /* int64[n]: its length per read */
    TBA_GET_DP_WORKGROUP = 28, /* int32[n]: zeros since ABI 9 (was: 1 where the removed workgroup-per-read form ran the main forward pass) */
    TBA_GET_ED_FORM = 29,     /* int32[n]: TBA_ED_FORM_*: the kernels that produced the read's change points
                               * (0: none did -- the read had failed before) */
    TBA_GET_TB_FORM = 30,     /* int32[n]: TBA_TB_FORM_*: the kernel that walked the read's main traceback */
    TBA_GET_TB_VERIFY_FAIL = 31, /* int32[n]: rows of the read where the verifier of the chunk-parallel traceback
                               * (k_tb_par_verify) found something else than its own walk; such a read was walked again
                               * by the lane-per-read kernel, so its result is the serial walk's either way.  Zero for
                               * every read is the expected state: a binding should treat anything else as a fault of
                               * the machine or the build worth reporting (ABI 9) */
    TBA_GET_DEBUG_COUNTERS = 99 /* int64[n][8]: ReadState.dbg, only filled by -DTBA_PHASE_DEBUG /
                                   -DTBA_SWEEP_STATS profiling builds (zeros otherwise) */
};
enum {
    TBA_ED_FORM_NONE = 0,
    TBA_ED_FORM_WG_SCAN_PEAKS = 1,  /* latency form: workgroup-per-read scan (k_cumsum_scores_long) + k_peaks */
    TBA_ED_FORM_DETECT_PICK = 2,    /* throughput form: k_detect + k_pick, no score array */
    TBA_ED_FORM_SCORES_PEAKS = 3,   /* pipelined k_cumsum_scores (or k_cumsum + k_scores_dna) + k_peaks: reads
                                     * k_detect / k_pick flagged, parameter sets outside their limits */
    TBA_ED_FORM_DETECT_TT_PICK = 4, /* RNA: k_detect_tt + k_pick */
    TBA_ED_FORM_TTEST_PEAKS = 5     /* RNA: k_scores_ttest + k_peaks */
};
enum {
    TBA_TB_FORM_NONE = 0,
    TBA_TB_FORM_LANE = 1,           /* k_main_tb: a lane per read (static bands, broken chains) */
    TBA_TB_FORM_LONG = 2,           /* k_main_tb_long */
    TBA_TB_FORM_PAR16 = 16,         /* k_main_tb_par<16>: throughput form */
    TBA_TB_FORM_PAR64 = 64          /* k_main_tb_par<64>: latency form, and the long reads of any batch */
};
int tba_batch_get(tba_engine *e, int what, void *out, int64_t out_bytes);
/* ---- stepwise execution (the reference's public per-stage API, resquiggle.py:63-67) ---------
 * The pipeline in stages; tba_batch_run_stages runs first..last on the uploaded batch (starting
 * at TBA_STAGE_SEGMENT resets the per-read state).  tba_batch_put injects the inputs of a later
 * stage so that it can run without the earlier ones:
 *   segment_signal                 = stage SEGMENT   (get VALID_CPTS, SEG_NORM, SEG_SV)
 *   find_seq_start_in_events       = put EVENT_MEANS/VALID_CPTS/REF_*, stage START (get START)
 *   find_adaptive_base_assignment  = put VALID_CPTS + EVENT_MEANS, stages REF_LEVELS..ASSIGN
 *   find_static_base_assignment    = same with put START_STATE = 4 (get READ_TB)
 *   resolve_skipped_bases_with_raw = put NORM, REF_*, DP_SEGS, stage SKIP (get SEGS) */
enum {
    TBA_STAGE_SEGMENT = 0,      /* normalisation + event detection (+ stall removal) */
    TBA_STAGE_EVENT_MEANS = 1,  /* compute_base_means over the change points */
    TBA_STAGE_REF_LEVELS = 2,   /* get_exp_levels_from_seq */
    TBA_STAGE_START = 3,        /* find_seq_start_in_events (+ retry) */
    TBA_STAGE_ASSIGN = 4,       /* masked start / static fallback, adaptive DP, traceback */
    TBA_STAGE_SKIP = 5,         /* resolve_skipped_bases_with_raw */
    TBA_STAGE_RESCALE = 6       /* Theil-Sen rescale + final score */
};
enum {
    TBA_PUT_VALID_CPTS = 1,   /* int64, CSR by ev_off; per_read[i] = count */
    TBA_PUT_EVENT_MEANS = 2,  /* float64, CSR by ev_off */
    TBA_PUT_NORM = 3,         /* float64, CSR by raw_off: normalised signal */
    TBA_PUT_REF_MEANS = 4,    /* float64, CSR by ref_off */
    TBA_PUT_REF_SDS = 5,
    TBA_PUT_DP_SEGS = 6,      /* int64, CSR by seg_off; per_read[2i], [2i+1] = read_start, length */
    TBA_PUT_START_STATE = 7   /* per_read[i] = 4: force the static whole-read assignment */
};
int tba_batch_run_stages(tba_engine *e, int first_stage, int last_stage);
int tba_batch_put(tba_engine *e, int what, const void *data, int64_t bytes,
                  const int64_t *per_read);
/* per-read num_events for the NEXT tba_batch_upload (segment_signal's argument); NULL clears */
int tba_set_num_events(tba_engine *e, const int64_t *num_events, int64_t n_reads);

/* Per-base event statistics of the finished batch (the compute part of the Events table that
 * write_new_fast5_group stores, tombo_helper.py:2341-2362): c_new_mean_stds (_c_helper.pyx:38-57)
 * over every successful read's final signal and boundaries, on the device.  means / stds: CSR by
 * the reads' base counts (same layout as the ref_means of tba_batch_get), B_tot doubles each;
 * entries of failed reads are left untouched. */
int tba_batch_base_stats(tba_engine *e, double *means, double *stds, int64_t n_values);

/* bytes of algorithmic traffic / cell updates of the last uploaded batch */
int tba_batch_stats(tba_engine *e, double *algorithmic_bytes, double *dp_cells);

/* ---- per-kernel entry points (one call == one Cython call; host buffers) -----------------
 * These are the parity surface, not the throughput path: every call allocates and frees its
 * device temporaries (hipMalloc / hipFree) and synchronises the stream. */
/* c_adaptive_banded_forward_pass, _c_dynamic_programming.pyx:314-412: fwd_pass[(n_bases+1)*bw],
 * fwd_pass_tb[(n_bases+1)*bw] (int64 moves) and event_starts[n_bases] are updated in place
 * from row start_seq_pos. */
int tba_c_adaptive_banded_forward_pass(tba_engine *e, double *fwd_pass, int64_t *fwd_pass_tb,
    int64_t n_bases, int64_t bandwidth, int64_t *event_starts, const double *event_means,
    int64_t n_events, const double *r_ref_means, const double *r_ref_sds, double z_shift,
    double skip_pen, double stay_pen, int64_t start_seq_pos, double mask_fill_z_score,
    int do_winsorize_z, double max_half_z_score);
/* ... with return_z_scores=True (pyx:324,339,387-388,409-410): z_scores[(n_bases - start_seq_pos) * bw]
 * receives the shifted z-scores of every row the pass computed (NULL: as above) */
int tba_c_adaptive_banded_forward_pass_z(tba_engine *e, double *fwd_pass, int64_t *fwd_pass_tb,
    int64_t n_bases, int64_t bandwidth, int64_t *event_starts, const double *event_means,
    int64_t n_events, const double *r_ref_means, const double *r_ref_sds, double z_shift,
    double skip_pen, double stay_pen, int64_t start_seq_pos, double mask_fill_z_score,
    int do_winsorize_z, double max_half_z_score, double *z_scores);
/* c_banded_forward_pass, pyx:240-279 */
int tba_c_banded_forward_pass(tba_engine *e, const double *shifted_z_scores, int64_t n_bases,
    int64_t bandwidth, const int64_t *event_starts, double skip_pen, double stay_pen,
    double *fwd_pass, int64_t *fwd_pass_tb);
/* c_banded_traceback, pyx:281-310 */
int tba_c_banded_traceback(tba_engine *e, const int64_t *fwd_pass_tb, int64_t n_bases,
    int64_t bandwidth, const int64_t *event_starts, int64_t band_pos,
    int64_t band_boundary_thresh, int64_t *seq_poss);
/* c_base_z_scores, pyx:17-32 */
int tba_c_base_z_scores(tba_engine *e, const double *b_sig, int64_t n, double ref_mean,
    double ref_sd, int do_winsorize_z, double max_half_z_score, double *out);
/* c_new_means, _c_helper.pyx:59-71 */
int tba_c_new_means(tba_engine *e, const double *norm_signal, int64_t n_sig,
    const int64_t *new_segs, int64_t n_segs, double *means);
/* c_apply_outlier_thresh, _c_helper.pyx:73-87 */
int tba_c_apply_outlier_thresh(tba_engine *e, const double *sig, int64_t n, double lower_lim,
    double upper_lim, double *out);
/* c_valid_cpts_w_cap, _c_helper.pyx:89-120 (+ the sort of tombo_helper.py:76-82);
 * returns a TBA_* status (TBA_FEWER_CPTS ...) */
int tba_c_valid_cpts_w_cap(tba_engine *e, const double *sig, int64_t n, int64_t min_base_obs,
    int64_t running_stat_width, int64_t num_cpts, int64_t *cpts);
/* c_valid_cpts_w_cap_t_test, _c_helper.pyx:144-202 (+ sort) */
int tba_c_valid_cpts_w_cap_t_test(tba_engine *e, const double *sig, int64_t n,
    int64_t min_base_obs, int64_t running_stat_width, int64_t num_cpts, int64_t *cpts);

/* c_new_mean_stds, _c_helper.pyx:38-57: segment means and population standard deviations */
int tba_c_new_mean_stds(tba_engine *e, const double *norm_signal, int64_t n_sig,
    const int64_t *new_segs, int64_t n_segs, double *means, double *stds);
/* c_compute_slopes, _c_helper.pyx:362-377: all i<j slopes in itertools.combinations order;
 * slopes[n*(n-1)/2] */
int tba_c_compute_slopes(tba_engine *e, const double *r_event_means,
    const double *r_model_means, int64_t n, double max_slope, double *slopes);
/* c_reg_z_scores, _c_dynamic_programming.pyx:34-97: per base of [reg_start, reg_end) the
 * admissible signal interval and the negative half z-scores over it.  r_b_starts has
 * n_b_starts entries (indices up to reg_end are read).  bounds[2*i..] = (start, end) relative
 * to r_b_starts[reg_start]; z_off[reg_len+1] = offsets of each base's scores inside z (capacity
 * z_cap doubles; TBA_E_ARG if too small -- reg_len * (r_b_starts[reg_end] -
 * r_b_starts[reg_start]) always suffices). */
int tba_c_reg_z_scores(tba_engine *e, const double *r_sig, int64_t n_sig,
    const double *r_ref_means, const double *r_ref_sds, int64_t n_bases,
    const int64_t *r_b_starts, int64_t n_b_starts, int64_t reg_start, int64_t reg_end,
    int64_t max_base_shift, int64_t min_obs_per_base, int do_winsorize_z,
    double max_half_z_score, int64_t *bounds, int64_t *z_off, double *z, int64_t z_cap);
/* c_base_forward_pass, _c_dynamic_programming.pyx:99-163: b_data has b_end - b_start entries,
 * the prev_* arrays prev_b_end - prev_b_start; outputs b_fwd_data / b_last_diag of b_end -
 * b_start entries.  TBA_INTERNAL where the reference raises IndexError. */
int tba_c_base_forward_pass(tba_engine *e, const double *b_data, int64_t b_start, int64_t b_end,
    const double *prev_b_data, int64_t prev_b_start, int64_t prev_b_end,
    const double *prev_b_fwd_data, const int64_t *prev_b_last_diag, int64_t min_obs_per_base,
    double *b_fwd_data, int64_t *b_last_diag);
/* c_base_traceback, _c_dynamic_programming.pyx:165-182: *sig_pos = the new base boundary, or
 * -1 where the reference returns None */
int tba_c_base_traceback(tba_engine *e, const double *curr_b_data, int64_t curr_len,
    int64_t curr_start, const double *next_b_data, int64_t next_len, int64_t next_start,
    int64_t next_end, int64_t sig_start, int64_t min_obs_per_base, int64_t *sig_pos);

/* ---- row N4: per-position log-likelihood ratios of the model-comparison statistics ---------
 * c_calc_llh_ratio (kind 0), c_calc_llh_ratio_const_var (1), c_calc_scaled_llh_ratio_const_var (2)
 * (_c_helper.pyx:277-358) for n_windows windows of `width` values starting at starts[i] of the
 * per-base arrays (n_values entries each), the slicing of tombo_stats.py:4042-4074.  kind 1 / 2
 * take the constant variance from ref_vars[starts[i]] and ignore alt_vars (may be NULL);
 * kind 2: par = {scale_factor, density_height_factor, density_height_power}.  One call with
 * one window == one call of the Cython function.  Transcendental functions are the device
 * library's: results match the reference to ~1e-13 relative, kind 1 bit for bit. */
int tba_llh_ratio_windows(tba_engine *e, int kind, const double *means, const double *ref_means,
    const double *alt_means, const double *ref_vars, const double *alt_vars, int64_t n_values,
    int64_t width, const int64_t *starts, int64_t n_windows, const double *par, double *out);

/* compute_sample_compare_read_stats / compute_de_novo_read_stats (tombo_stats.py:3675-3873)
 * after their file access: per-base p-values 2 * Phi(-|mean - ref_mean| / ref_sd), combined by
 * Fisher's method over windows of 2 * fm_offset + 1 positions when fm_offset > 0
 * (calc_window_fishers_method :2252-2271, SMALLEST_PVAL floor; the first / last fm_offset
 * positions of a read are NaN), NaN wherever an input is NaN (control positions without
 * coverage).  floor_out != 0: the de novo form (result floored at smallest_pval).  n_reads reads
 * as CSR slices off[n_reads + 1] of the three arrays and of pvals.  erfc / log / exp are the
 * device library's: ~1e-14 relative to scipy. */
int tba_read_pvals(tba_engine *e, const double *means, const double *ref_means,
    const double *ref_sds, const int64_t *off, int64_t n_reads, int64_t fm_offset, int floor_out,
    double smallest_pval, double *pvals);
/* The de novo statistic of every read of the finished resident batch, nothing uploaded: per-base
 * means (c_new_means over the final signal and boundaries, as tba_batch_base_stats) against the
 * batch's own expected levels, which are the canonical model's levels of the read sequence --
 * compute_de_novo_read_stats (tombo_stats.py:3771-3873) for a read tested over its whole length.
 * pvals: CSR by the reads' base counts (layout of tba_batch_base_stats), n_values >= B_tot;
 * per read only [central_pos, B - (K - central_pos - 1)) is tested (k-mers inside the read),
 * the rest and failed reads are NaN. */
int tba_batch_de_novo_stats(tba_engine *e, int64_t fm_offset, double smallest_pval,
                            double *pvals, int64_t n_values);

/* ---- the worker's per-read preparation (row P10) ---------------------------------------------
 * ts.identify_stalls(all_raw_signal, stall_params) (tombo_stats.py:269-368), running-window-mean
 * method, for one read in host memory (raw_dtype: TBA_RAW_*): ints[2 i], ints[2 i + 1] = the
 * i-th (widened, merged) stall interval, *n_ints of them (cap = capacity of ints in intervals;
 * TBA_E_ARG with *n_ints set when it is too small).  Inside a batch the same kernels run under
 * tba_opts.detect_stalls. */
int tba_identify_stalls(tba_engine *e, const void *raw, int raw_dtype, int64_t n,
    int64_t window_size, int64_t n_windows, int64_t mini_window_size, double threshold,
    int64_t min_consecutive_obs, int64_t edge_buffer, int64_t *ints, int64_t cap, int64_t *n_ints);

/* Host-side, no GPU involved: pack n_reads per-read sample arrays (raw_ptrs[i], raw_off[i+1] -
 * raw_off[i] samples of type raw_dtype; reverse != 0: copied back to front) and sequences
 * (seq_ptrs[i]: ASCII A/C/G/T, seq_off[i+1] - seq_off[i] letters, stored as codes 0..3, anything
 * else as 255 -> TBA_INVALID_SEQ) into the CSR buffers tba_batch_upload_async takes, with
 * n_threads threads.  This is the reader of the reference's worker pool (resquiggle.py:1385-1486)
 * for callers that hold one array per read. */
int tba_pack_reads(int64_t n_reads, const void *const *raw_ptrs, int raw_dtype, int reverse,
    const int64_t *raw_off, void *raw_out, const char *const *seq_ptrs, const int64_t *seq_off,
    uint8_t *seq_out, int n_threads);

/* ... and back: count[i] elements of elem_bytes bytes from src + src_off[i] (elements) into
 * dst_ptrs[i], n_threads threads (per-read result arrays out of one flat download) */
int tba_unpack_reads(int64_t n_reads, const void *src, int64_t elem_bytes, const int64_t *src_off,
    const int64_t *count, void *const *dst_ptrs, int n_threads);

/* ---- synthetic reads made on the device (bench / test support) ---------------------------------
 * No counterpart in the reference: its benchmark input is a directory of FAST5 files.  The
 * multi-GPU job of BASELINE.json (a million distinct 10 kb reads through a host work queue) cannot
 * be synthesised on the host cores inside a benchmark's set-up, so every batch of that job is
 * drawn on the device from a counter-based generator keyed by (seed, first_read + read, element):
 * the reads of tombo_amd/synth.py (uniform ACGT, level = the model's k-mer mean, dwell =
 * max(min_dwell, Geometric(1 / mean_dwell)), level + noise per sample, n_lead / n_trail samples
 * of open-pore-like signal around them), reproducible on any rank and bit for bit by the numpy
 * restatement tombo_amd.synth.device_reads_reference (csrc/k_synth.h states the draws). */
typedef struct tba_synth tba_synth;
typedef struct {
    int64_t mean_dwell, min_dwell, n_lead, n_trail;
    double scale, offset, noise_sd;   /* pA = (level + noise * noise_sd) * scale + offset */
    double dac_per_pa, dac_offset;    /* TBA_RAW_I16: rint(pA * dac_per_pa + dac_offset) */
    int32_t reverse, pad;             /* reverse: samples in acquisition order of a 3'->5' (RNA) run */
} tba_synth_params;
int tba_synth_create(int device, const double *kmer_means, int64_t kmer_width, tba_synth **out);
void tba_synth_destroy(tba_synth *g);
/* n_reads reads of n_bases[i] bases (sequence: n_bases[i] + kmer_width - 1 codes), raw_dtype
 * TBA_RAW_I16 or TBA_RAW_F64.  Returns when the batch is in device memory: raw_off / seq_off
 * (n_reads + 1, host) get its CSR offsets, *d_raw / *d_seq the device arrays (owned by the
 * generator, valid until its next call) -- tba_batch_upload_async takes them in place of host
 * arrays. */
int tba_synth_generate(tba_synth *g, const tba_synth_params *p, uint64_t seed, int64_t first_read,
    int64_t n_reads, const int64_t *n_bases, int raw_dtype, int64_t *raw_off, int64_t *seq_off,
    const void **d_raw, const uint8_t **d_seq);
/* the last generated batch to host memory (either may be NULL) */
int tba_synth_download(tba_synth *g, void *raw, uint8_t *seq);
/* host only: the generator's dwell thresholds (n <= 256) and noise constant under p */
int tba_synth_dwell_thresholds(const tba_synth_params *p, uint32_t *thr, int64_t n, double *noise_norm);

/* out[0..2] = sizeof(tba_params), sizeof(tba_opts), sizeof(tba_read_result) of this build, out[3]
 * (n >= 4) = TBA_ABI_VERSION: lets a binding without a C compiler (ctypes) check its struct mirrors
 * and refuse a stale build of the library */
#define TBA_ABI_VERSION 9
int tba_abi_sizes(int64_t *out, int64_t n);

/* self-test: out[t] = index t of the subsample tba_opts.device_subsample draws for read
 * `read_index` of a batch under `seed`, for a read of n bases (t < count <= n) */
int tba_selftest_subsample(tba_engine *e, int64_t n, uint64_t seed, int64_t read_index,
                           int64_t count, int64_t *out);
/* self-test: out[i] = the row-constant division used inside the DP kernel (reciprocal + two
 * residual corrections) for a[i] / b[i]; must equal the IEEE quotient bit for bit */
int tba_selftest_division(tba_engine *e, const double *a, const double *b, int64_t n,
                          double *out);
/* self-test: out[i] = the approximate quotient a[i] * r, r = v_rcp_f64(b[i]) + one Newton step,
 * that the Theil-Sen kernel classifies slope pairs with (calc_kmer_fitted_shift_scale,
 * tombo_stats.py:401-425); its relative error must stay far inside the 1e-5 guard band */
int tba_selftest_approx_quotient(tba_engine *e, const double *a, const double *b, int64_t n,
                                 double *out);

#ifdef __cplusplus
}
#endif
#endif
