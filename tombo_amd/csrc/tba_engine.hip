// tba_engine.hip -- batch engine + C ABI (include/tombo_amd.h) of the gfx950 resquiggle path.
// One engine == one GPU == one HIP stream; a batch is a fixed sequence of kernels over ragged
// SoA buffers that stay resident in HBM between upload and download.
#include <atomic>
#include "tba_common.h"
#include "k_select.h"
#include "k_segment.h"
#include "k_detect.h"
#include "k_prep_raw.h"
#include "k_dp.h"
#include "k_tb_par.h"
#include "k_dp_multi.h"
#include "k_long.h"
#include "k_dp_wg.h"
#include "k_tail.h"
#include "k_cabi.h"
#include "k_synth.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_last_error;
static int set_err(int code, const std::string &msg) { g_last_error = msg; return code; }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return set_err(TBA_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return set_err(TBA_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *)p; }
};
// page-locked host buffer (engine-owned staging of the small per-batch records, so that the
// "async" upload never falls back to HIP's blocking pageable path)
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return set_err(TBA_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *)p; }
};

enum { N_STAGE = 16 };
static const char *STAGE_NAMES[N_STAGE] = {
    "normalize", "cumsum", "scores", "peaks", "event_means", "ref_levels", "start_dp",
    "start_tb", "prep", "main_dp", "main_tb", "skip_resolve", "theil_sen", "rescale_score",
    "stalls", "total"};

#define WIDE_BLOCKS 64 // workgroups of k_dp_wide (each owns two scratch rows)
#define TB_LANES 16    // reads per wavefront of the latency-bound lane-per-read kernels
#ifndef TBA_SMALL_BATCH
#define TBA_SMALL_BATCH 1024 // up to this many reads the scan of event detection runs a workgroup per read
                             // (k_detect costs ~7 us per 128 samples whatever the batch: 10 kb reads, event detection
                             // 384 reads 5.1 -> 1.4 ms, 1 024 reads 5.1 -> 3.2, 2 048 reads 5.4 -> 6.2: profiles/r04_small_batch_scan.txt)
#endif

// k_peaks is compiled per exclusion radius (min_obs_per_base - 1): 2 and 5 are the defaults of
// the DNA / RNA parameter sets, anything else takes the generic kernel
static void launch_peaks(i64 min_obs_per_base, unsigned n_blocks, hipStream_t s, ReadState *rs,
                         const DevParams *dp, const double *score, unsigned char *state,
                         double *dense, i64 *valid_cpts, int ttest, int only_flagged = 0,
                         int form = TBA_ED_FORM_SCORES_PEAKS)
{
    if (min_obs_per_base - 1 == 2) k_peaks<2><<<n_blocks, SEL_NT, 0, s>>>(rs, dp, score, state, dense, valid_cpts, ttest, only_flagged, form);
    else if (min_obs_per_base - 1 == 5) k_peaks<5><<<n_blocks, SEL_NT, 0, s>>>(rs, dp, score, state, dense, valid_cpts, ttest, only_flagged, form);
    else k_peaks<0><<<n_blocks, SEL_NT, 0, s>>>(rs, dp, score, state, dense, valid_cpts, ttest, only_flagged, form);
}
struct tba_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    // Side stream of a full pipeline run: what does not depend on event detection -- the worker's
    // stall detection over the raw samples (RNA) and the expected levels of the sequence -- runs
    // beside the normalisation / event detection kernels of the main stream and is joined before its
    // first consumer (k_remove_stalls; start discovery).  Both groups wait on memory most of their time
    // (SQ_WAIT_ANY 60-80 % of their wave cycles): together they fill what each leaves idle.
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_stalls = nullptr, ev_levels = nullptr, ev_st0 = nullptr, ev_st1 = nullptr, ev_skip0 = nullptr, ev_skip1 = nullptr;
    hipEvent_t ev[N_STAGE + 1] = {};
    float stage_ms[32] = {};
    bool have_model = false, have_batch = false, ran = false;
    bool finished = false; // the last stage (rescale + score) has run on the uploaded batch
    DevParams hp;
    i64 n_reads = 0, S_tot = 0, seq_tot = 0, B_tot = 0, E_tot = 0, max_raw = 0, max_B = 0;
    i64 start_moves_stride = 0, moves_arena = 0, skip_arena = 0, wide_w = 0, n_stall_cap = 0;
    bool any_stall = false, have_samp = false, have_sv = false;
    std::vector<i64> ne_override; // per-read num_events for the next upload (stepwise API)
    double algo_bytes = 0, dp_cells = 0;
    int n_sharing = 1;            // engines fed concurrently on this device (tba_engine_set_sharing)
    int side_mode = -1;           // tba_engine_set_side_stream: -1 by the engines alive, 0 never, 1 always
    bool last_side = false;       // the last full run used the side stream
    // latency / throughput forms of event detection and traceback (tba_engine_set_dispatch)
    i64 small_batch = TBA_SMALL_BATCH, tb_wave_below = TBP_WAVE_BELOW;
    int last_c_ed_form = 0;       // tba_c_last_ed_form
    int raw_dtype = TBA_RAW_F64;
    PinBuf h_rs, h_dp;            // ReadState[n] / DevParams as uploaded (pinned)
    DevBuf d_res, d_segs32;       // packed results of tba_batch_download_async
    DevBuf d_skipq;               // window queues of k_skip_dp_wave
    DevBuf d_rs, d_dp, d_kmeans, d_ksds, d_raw, d_norm, d_norm_out, d_csum, d_score, d_state,
        d_cpts, d_evm, d_seq, d_refm, d_refs, d_bst, d_lo, d_hi, d_readtb, d_dpsegs, d_segs,
        d_win, d_absz, d_sv_in, d_samp, d_stall, d_lastrow, d_startvals, d_smoves,
        d_moves, d_dscr, d_wide, d_stat, d_order, d_long,
        d_stall_csum, d_stall_bits;  // the stall detector's own scratch (it runs beside event detection)
    PinBuf h_order;               // read indices by decreasing length (k_dp_multi's grouping)
    PinBuf h_long;                // indices of the long reads (k_long.h)
    i64 n_long = 0;
    void release_all()
    {
        DevBuf *all[] = {&d_rs, &d_dp, &d_kmeans, &d_ksds, &d_raw, &d_norm, &d_norm_out, &d_csum,
                         &d_score, &d_state, &d_cpts, &d_evm, &d_seq, &d_refm, &d_refs, &d_bst,
                         &d_lo, &d_hi, &d_readtb, &d_dpsegs, &d_segs, &d_win, &d_absz,
                         &d_sv_in, &d_samp, &d_stall, &d_lastrow, &d_startvals, &d_smoves,
                         &d_moves, &d_dscr, &d_wide, &d_stat, &d_res, &d_segs32, &d_skipq, &d_order, &d_long,
                         &d_stall_csum, &d_stall_bits};
        for (DevBuf *b : all) b->release();
        h_order.release();
        h_long.release();
        h_rs.release();
        h_dp.release();
    }
};

extern "C" const char *tba_last_error(void) { return g_last_error.c_str(); }

extern "C" int tba_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// engines alive per device in this process: the side stream (enqueue_stages) is only used while the
// process's streams fit the device's hardware queues (four by default) -- beyond that streams share
// a queue, and a side stream queued behind ANOTHER engine's half-second forward pass holds its own
// engine's main stream at the join (measured: eight resident long-tail batches 37.8 k -> 31.4 k reads/s)
#define TBA_MAX_DEVICES 64
static std::atomic<int> g_live_engines[TBA_MAX_DEVICES];
static int side_stream_max_engines()
{
    static const int v = [] { const char *x = getenv("TBA_SIDE_STREAM_MAX_ENGINES"); return x ? atoi(x) : 2; }();
    return v;
}

extern "C" int tba_engine_create(int device, tba_engine **out)
{
    if (!out) return set_err(TBA_E_ARG, "out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return set_err(TBA_E_HIP, "no HIP device visible: the resquiggle engine has no CPU fallback");
    if (device < 0 || device >= n) return set_err(TBA_E_ARG, "bad device ordinal");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return set_err(TBA_E_HIP, std::string("device is ") + prop.gcnArchName +
                                      ", this library carries gfx950 code only");
    tba_engine *e = new tba_engine();
    e->device = device;
    if (const char *v = getenv("TBA_SMALL_BATCH_READS")) e->small_batch = std::max<i64>(atoll(v), 0);
    if (const char *v = getenv("TBA_TB_WAVE_BELOW")) e->tb_wave_below = std::max<i64>(atoll(v), 0);
    HIP_TRY(hipStreamCreate(&e->stream));
    if (device < TBA_MAX_DEVICES) g_live_engines[device]++;
    for (hipEvent_t *x : {&e->ev_fork, &e->ev_stalls, &e->ev_levels, &e->ev_st0, &e->ev_st1, &e->ev_skip0, &e->ev_skip1}) HIP_TRY(hipEventCreate(x));
    for (int i = 0; i <= N_STAGE; i++) HIP_TRY(hipEventCreate(&e->ev[i]));
    *out = e;
    return 0;
}

extern "C" void tba_engine_destroy(tba_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    if (e->stream2) (void)hipStreamSynchronize(e->stream2);
    e->release_all();
    for (int i = 0; i <= N_STAGE; i++) if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
    for (hipEvent_t x : {e->ev_fork, e->ev_stalls, e->ev_levels, e->ev_st0, e->ev_st1, e->ev_skip0, e->ev_skip1}) if (x) (void)hipEventDestroy(x);
    if (e->stream2) (void)hipStreamDestroy(e->stream2);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->device < TBA_MAX_DEVICES) g_live_engines[e->device]--;
    delete e;
}

extern "C" int tba_set_model(tba_engine *e, const double *kmer_means, const double *kmer_sds,
                             int64_t kmer_width, int64_t central_pos)
{
    if (!e || !kmer_means || !kmer_sds || kmer_width < 1 || kmer_width > 12)
        return set_err(TBA_E_ARG, "bad model arguments");
    HIP_TRY(hipSetDevice(e->device));
    size_t n = (size_t)1 << (2 * kmer_width);
    if (e->d_kmeans.ensure(n * 8) || e->d_ksds.ensure(n * 8)) return TBA_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(e->d_kmeans.p, kmer_means, n * 8, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_ksds.p, kmer_sds, n * 8, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->hp.kmer_width = kmer_width;
    e->hp.central_pos = central_pos;
    e->have_model = true;
    // a batch uploaded under another model has stale per-read geometry (B depends on K)
    e->have_batch = e->ran = e->finished = false;
    return 0;
}

extern "C" int tba_device_mem(tba_engine *e, int64_t *free_bytes, int64_t *total_bytes)
{
    if (!e) return set_err(TBA_E_ARG, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return 0;
}

extern "C" int tba_pinned_alloc(int64_t bytes, void **out)
{
    if (!out || bytes < 0) return set_err(TBA_E_ARG, "bad arguments");
    void *p = nullptr;
    hipError_t rc = hipHostMalloc(&p, (size_t)(bytes > 0 ? bytes : 1), hipHostMallocDefault);
    if (rc != hipSuccess) return set_err(TBA_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(rc));
    *out = p;
    return 0;
}
extern "C" int tba_pinned_free(void *p)
{
    if (p && hipHostFree(p) != hipSuccess) return set_err(TBA_E_HIP, "hipHostFree failed");
    return 0;
}

// launch a kernel template instantiated for the batch's raw sample type
#define RAW_DISPATCH(dt_, call_)                                                               \
    do {                                                                                       \
        if ((dt_) == TBA_RAW_I16) { typedef int16_t RT; call_; }                               \
        else if ((dt_) == TBA_RAW_F32) { typedef float RT; call_; }                            \
        else { typedef double RT; call_; }                                                     \
    } while (0)

// ---- batch sizing -----------------------------------------------------------------------------
// Everything the device buffers of a batch depend on, from the per-read lengths alone (shared by
// tba_batch_upload_async and tba_batch_footprint).
static size_t raw_elem_bytes(int dt) { return dt == TBA_RAW_I16 ? 2 : dt == TBA_RAW_F32 ? 4 : 8; }

struct BatchSizes {
    i64 S_tot = 0, seq_tot = 0, B_tot = 0, E_tot = 0, max_raw = 0, max_B = 0, max_nev = 0;
    i64 moves_need = 0, start_moves_stride = 0, moves_arena = 0, skip_arena = 0, wide_w = 0;
    i64 n_stall = 0;
    double algo_bytes = 0, cells = 0;
};

// per-read geometry; rs may be NULL (footprint only).  Returns 0 or TBA_E_ARG.
static int plan_batch(const tba_params *p, const tba_opts *o, i64 K, i64 n, const i64 *raw_off,
                      const i64 *n_raw_arr, const i64 *seq_off, const i64 *seq_len_arr,
                      const std::vector<i64> &ne_override, const int32_t *sv_flags,
                      const i64 *stall_off, ReadState *rs, BatchSizes &z)
{
    const int cpl_main = cpl_class(p->bandwidth);
    if (cpl_class(p->start_bw) == 0 || cpl_class(p->start_save_bw) == 0 || cpl_main == 0)
        return set_err(TBA_E_ARG, "bandwidth / start bandwidth above TBA_MAX_BAND");
    i64 ref_acc = 0, ev_acc = 0, raw_acc = 0, seq_acc = 0, stall_acc = 0;
    for (i64 i = 0; i < n; i++) {
        const i64 n_raw = raw_off ? raw_off[i + 1] - raw_off[i] : n_raw_arr[i];
        const i64 seq_len = seq_off ? seq_off[i + 1] - seq_off[i] : seq_len_arr[i];
        if (n_raw < 0 || seq_len < 0) return set_err(TBA_E_ARG, "negative read length");
        ReadState tmp;
        ReadState &r = rs ? rs[i] : tmp;
        memset(&r, 0, sizeof(r));
        r.raw_off = raw_acc; r.n_raw = n_raw; raw_acc += n_raw;
        r.seq_off = seq_acc; r.seq_len = seq_len; seq_acc += seq_len;
        const i64 B = seq_len - K + 1;
        r.ref_off = ref_acc;
        r.seg_off = ref_acc + i;
        r.B = B > 0 ? B : 0;
        ref_acc += r.B;
        r.ev_off = ev_acc;
        r.status = TBA_OK;
        r.sv_flags = sv_flags ? sv_flags[i] : 0;
        if (stall_off) { r.stall_off = stall_off[i]; r.n_stall = stall_off[i + 1] - stall_off[i]; }
        else if (o->detect_stalls) { // capacity of k_stall_runs: a run is longer than min_consecutive_obs
            r.stall_off = stall_acc;
            stall_acc += (n_raw > 0 ? n_raw : 0) / (o->stall_min_consecutive_obs + 1) + 2;
        }
        if (B <= 0 || n_raw <= 0) {
            r.status = n_raw <= 0 ? TBA_NO_RAW : TBA_INTERNAL;
            continue;
        }
        // ts.compute_num_events (tombo_stats.py:1558-1574) and the guard of resquiggle.py:1159
        i64 num_events = std::max(n_raw / p->mean_obs_per_event,
                                  (i64)((double)B * o->min_event_to_seq_ratio));
        const bool forced = (i64)ne_override.size() == n && ne_override[(size_t)i] > 0;
        if (forced) num_events = ne_override[(size_t)i]; // caller-chosen (segment_signal)
        r.num_events = num_events; // event space is reserved for every read with B > 0
        ev_acc += num_events;
        if (!forced && (double)num_events / (double)p->bandwidth > (double)B) { r.status = TBA_TOO_MUCH_SIGNAL; continue; }
        if (num_events <= 1 || n_raw < 4 * p->running_stat_width + 2) { r.status = TBA_INTERNAL; continue; }
        z.max_raw = std::max(z.max_raw, n_raw);
        z.max_B = std::max(z.max_B, B);
        r.is_long = n_raw > TBA_LONG_RAW || B > TBA_LONG_BASES;
        const i64 n_ev = num_events - 1;
        const bool short_read = n_ev < p->start_bw + p->start_n_bases || B < p->start_n_bases;
        // packed move rows: the adaptive band, or the whole-read static band of a short read
        // (n_ev - mask_len cells, any width: k_dp_wide beyond the widest class)
        i64 row_bytes = mv_row_bytes(p->bandwidth);
        if (short_read) row_bytes = std::max(row_bytes, mv_row_bytes(n_ev));
        z.moves_need += (B + 1) * (row_bytes + MV_STRIP_BYTES); // (+ the centre strip of the adaptive rows, k_dp.h)
        z.max_nev = std::max(z.max_nev, n_ev);
        // algorithmic traffic: raw in + seq + norm out + segs + band starts +
        // 2-bit moves + scalars
        z.algo_bytes += 8.0 * n_raw + (double)seq_len + 8.0 * n_raw + 8.0 * (B + 1) + 8.0 * B +
                        (double)((B * p->bandwidth + 3) / 4) + 64.0;
        z.cells += (double)B * p->bandwidth + (short_read ? 0.0 : (double)p->start_n_bases * p->start_bw);
    }
    z.S_tot = raw_acc; z.seq_tot = seq_acc; z.B_tot = ref_acc; z.E_tot = ev_acc;
    z.wide_w = z.max_nev > TBA_MAX_BAND ? ((z.max_nev + 63) / 64) * 64 : 0;
    const i64 start_w = std::max(p->start_bw, p->start_save_bw);
    z.start_moves_stride = (p->start_n_bases + 1) * (i64)mv_class_rowb(cpl_class(start_w));
    z.moves_arena = z.moves_need + z.moves_need / 8 + (64ll << 20);
    // raw-DP scratch arena (8-byte units): windows are a few bases x tens of samples; reads that
    // do not fit the arena get TBA_UNSUPPORTED
    z.skip_arena = n * 32768 + (32ll << 20);
    z.n_stall = stall_off ? stall_off[n] : stall_acc;
    return 0;
}

// the device buffers of a batch: (buffer, bytes) through `f`
template <class F>
static void for_each_batch_buffer(tba_engine *e, const tba_params *p, const tba_opts *o, i64 n,
                                  const BatchSizes &z, int raw_dtype, F f)
{
    const size_t S = (size_t)std::max<i64>(z.S_tot, 1), Bt = (size_t)std::max<i64>(z.B_tot, 1),
                 Et = (size_t)std::max<i64>(z.E_tot, 1), N = (size_t)n;
    tba_engine *q = e; // q == NULL: sizes only
#define BUF(name_, bytes_) f(q ? &q->name_ : (DevBuf *)nullptr, (size_t)(bytes_))
    BUF(d_rs, N * sizeof(ReadState));
    BUF(d_dp, sizeof(DevParams));
    // (+ 64 bytes: the 16-byte accesses of a pass over a signal may touch the element past an odd end)
    BUF(d_raw, S * raw_elem_bytes(raw_dtype) + 64);
    BUF(d_norm, S * 8 + 64);
    if (!o->skip_norm_out) BUF(d_norm_out, S * 8 + 64);
    BUF(d_csum, (S + N) * 8 + 64);
    BUF(d_score, S * 8 + 64);
    BUF(d_state, std::max(S, S / 8 + 8 * N + 64)); // (also the stall detector's bit words)
    BUF(d_cpts, Et * 8);
    BUF(d_evm, Et * 8);
    BUF(d_seq, (size_t)std::max<i64>(z.seq_tot, 1));
    BUF(d_refm, Bt * 8);
    BUF(d_refs, Bt * 8);
    BUF(d_bst, Bt * 8);
    BUF(d_lo, Bt * 4);
    BUF(d_hi, Bt * 4);
#ifdef TBA_TB_B2
    BUF(d_readtb, (Bt + N) * 8 * 3);
#else
    BUF(d_readtb, (Bt + N) * 8);
#endif
    BUF(d_dpsegs, (Bt + N) * 8);
    BUF(d_segs, (Bt + N) * 8);
    BUF(d_win, (Bt + N) * 24);
    BUF(d_absz, Bt * 8);
    BUF(d_sv_in, N * 32);
    BUF(d_samp, N * MAX_TS_POINTS * 8);
    BUF(d_lastrow, N * TBA_MAX_BAND * 8);
    BUF(d_startvals, N * (size_t)p->start_n_bases * 8);
    BUF(d_smoves, N * (size_t)z.start_moves_stride);
    BUF(d_moves, (size_t)z.moves_arena);
    BUF(d_dscr, (size_t)z.skip_arena * 8);
    if (z.wide_w) BUF(d_wide, (size_t)WIDE_BLOCKS * 2 * (size_t)z.wide_w * 8);
    if (z.n_stall > 0) BUF(d_stall, (size_t)z.n_stall * 16);
    if (o->detect_stalls) { // (its scratch is its own: the detector runs beside event detection, which owns csum / state)
        BUF(d_stall_bits, S / 8 + 8 * N + 64);
        if (!(raw_dtype == TBA_RAW_I16 && o->stall_window_size <= SI_MAXW)) BUF(d_stall_csum, (S + N) * 8 + 64);
    }
    BUF(d_skipq, 64 + 3 * (N * 32 + 4096) * 8);
    BUF(d_order, N * 4);
    BUF(d_long, N * 4);
    BUF(d_res, N * sizeof(tba_read_result));
    BUF(d_segs32, (Bt + N) * 4);
#undef BUF
}

extern "C" int tba_batch_footprint(const tba_params *p, const tba_opts *o, int64_t kmer_width,
                                   int raw_dtype, int64_t n_reads, const int64_t *n_raw,
                                   const int64_t *seq_len, double *bytes)
{
    if (!p || !o || !n_raw || !seq_len || !bytes || n_reads <= 0 || kmer_width < 1)
        return set_err(TBA_E_ARG, "bad arguments");
    if (raw_dtype < TBA_RAW_F64 || raw_dtype > TBA_RAW_I16) return set_err(TBA_E_ARG, "unknown raw dtype");
    BatchSizes z;
    std::vector<i64> none;
    if (int rc = plan_batch(p, o, kmer_width, n_reads, nullptr, n_raw, nullptr, seq_len, none, nullptr,
                            nullptr, nullptr, z))
        return rc;
    double tot = 0;
    for_each_batch_buffer(nullptr, p, o, n_reads, z, raw_dtype,
                          [&](DevBuf *, size_t b) { tot += (double)(b + b / 8 + 256); });
    *bytes = tot;
    return 0;
}

extern "C" int tba_batch_upload_async(tba_engine *e, const tba_params *p, const tba_opts *o,
                                      int64_t n_reads, const void *raw, int raw_dtype,
                                      const int64_t *raw_off, const uint8_t *seq,
                                      const int64_t *seq_off, const double *sv_in,
                                      const int32_t *sv_flags, const int64_t *samp_ind,
                                      const int64_t *stall_ints, const int64_t *stall_off)
{
    // the forced event counts apply to this upload only, whatever its outcome
    std::vector<i64> ne_override;
    if (e) ne_override.swap(e->ne_override);
    if (!e || !p || !o || n_reads <= 0 || !raw || !raw_off || !seq || !seq_off)
        return set_err(TBA_E_ARG, "bad batch arguments");
    if (raw_dtype < TBA_RAW_F64 || raw_dtype > TBA_RAW_I16) return set_err(TBA_E_ARG, "unknown raw dtype");
    // several kernels index the read with the y dimension of the grid
    if (n_reads > TBA_MAX_BATCH_READS) return set_err(TBA_E_ARG, "more than TBA_MAX_BATCH_READS reads in one batch");
    if (!e->have_model) return set_err(TBA_E_STATE, "tba_set_model has not been called");
    if (o->del_fix_window < 0 || o->max_del_fix_window < 0 || !(o->extra_sig_factor >= 0.0))
        return set_err(TBA_E_ARG, "bad skipped-base window parameters");
    if (p->bandwidth < 2 || p->running_stat_width < 1 || p->min_obs_per_base < 1 ||
        p->raw_min_obs_per_base < 1 || p->mean_obs_per_event < 1 || p->start_n_bases < 1)
        return set_err(TBA_E_ARG, "bad resquiggle parameters");
    // CSR offsets: start at 0, never decrease (a foreign caller's mistake must not become an
    // out-of-bounds device access)
    if (raw_off[0] != 0 || seq_off[0] != 0) return set_err(TBA_E_ARG, "offset arrays must start at 0");
    for (i64 i = 0; i < n_reads; i++)
        if (raw_off[i + 1] < raw_off[i] || seq_off[i + 1] < seq_off[i])
            return set_err(TBA_E_ARG, "offset arrays must be non-decreasing");
    const bool stalls = stall_ints && stall_off;
    if (o->detect_stalls) {
        if (stalls) return set_err(TBA_E_ARG, "stall_ints given together with tba_opts.detect_stalls");
        // th.stallParams of the running-window-mean method (tombo_stats.py:317-323)
        if (o->stall_n_windows < 2 || o->stall_n_windows > 16 || o->stall_mini_window_size < 1 ||
            o->stall_window_size != o->stall_n_windows * o->stall_mini_window_size ||
            o->stall_min_consecutive_obs < 0 || !(o->stall_threshold == o->stall_threshold))
            return set_err(TBA_E_ARG, "bad stall detection parameters");
    }
    if (stalls) {
        if (stall_off[0] != 0) return set_err(TBA_E_ARG, "offset arrays must start at 0");
        for (i64 i = 0; i < n_reads; i++) {
            if (stall_off[i + 1] < stall_off[i]) return set_err(TBA_E_ARG, "offset arrays must be non-decreasing");
            // interval ends ascend (k_remove_stalls binary-searches them; the values themselves are
            // only compared, never used as indices: identify_stalls widens past the signal ends)
            for (i64 k = stall_off[i] + 1; k < stall_off[i + 1]; k++)
                if (stall_ints[2 * k + 1] < stall_ints[2 * k - 1])
                    return set_err(TBA_E_ARG, "stall interval ends must ascend");
        }
    }
    HIP_TRY(hipSetDevice(e->device));
    // the previous batch of this engine must be done with the buffers (and with h_rs)
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->have_batch = false;
    e->ran = false;
    e->finished = false;
    e->hp.p = *p;
    e->hp.o = *o;
    e->hp.fill_masked = (MASK_FILL_Z_SCORE - p->z_shift) + p->z_shift;
    if (o->del_fix_window == 0 && o->max_del_fix_window == 0 && o->extra_sig_factor == 0.0) {
        // a zero-initialised tba_opts: the reference's defaults
        e->hp.o.del_fix_window = DEL_FIX_WINDOW; e->hp.o.max_del_fix_window = MAX_DEL_FIX_WINDOW;
        e->hp.o.extra_sig_factor = EXTRA_SIG_FACTOR;
    }
    const i64 n = n_reads;
    if (e->h_rs.ensure((size_t)n * sizeof(ReadState)) || e->h_dp.ensure(sizeof(DevParams))) return TBA_E_NOMEM;
    BatchSizes z;
    if (int rc = plan_batch(p, o, e->hp.kmer_width, n, raw_off, nullptr, seq_off, nullptr, ne_override,
                            sv_flags, stalls ? stall_off : nullptr, e->h_rs.as<ReadState>(), z))
        return rc;
    e->n_reads = n;
    e->S_tot = z.S_tot; e->seq_tot = z.seq_tot; e->B_tot = z.B_tot; e->E_tot = z.E_tot;
    e->max_raw = z.max_raw; e->max_B = z.max_B; e->wide_w = z.wide_w;
    e->algo_bytes = z.algo_bytes; e->dp_cells = z.cells;
    e->any_stall = (stalls && stall_off[n] > 0) || o->detect_stalls;
    e->n_stall_cap = z.n_stall;
    e->have_samp = samp_ind != nullptr;
    e->have_sv = sv_in != nullptr && sv_flags != nullptr;
    e->start_moves_stride = z.start_moves_stride;
    e->moves_arena = z.moves_arena;
    e->skip_arena = z.skip_arena;
    e->raw_dtype = raw_dtype;
    int rc = 0;
    for_each_batch_buffer(e, p, o, n, z, raw_dtype, [&](DevBuf *b, size_t bytes) { rc |= b->ensure(bytes); });
    if (rc) return TBA_E_NOMEM;

    hipStream_t s = e->stream;
    const size_t N = (size_t)n;
    { // reads by decreasing length (stable): the groups of a k_dp_multi wavefront finish together
        if (e->h_order.ensure(N * 4)) return TBA_E_NOMEM;
        i32 *ord = e->h_order.as<i32>();
        const ReadState *hrs = e->h_rs.as<ReadState>();
        for (i64 i = 0; i < n; i++) ord[i] = (i32)i;
        std::stable_sort(ord, ord + n, [hrs](i32 a, i32 b) { return hrs[a].B > hrs[b].B; });
        HIP_TRY(hipMemcpyAsync(e->d_order.p, ord, N * 4, hipMemcpyHostToDevice, s));
        // the long reads, longest first (is_long was set by plan_batch)
        if (e->h_long.ensure(N * 4)) return TBA_E_NOMEM;
        i32 *lg = e->h_long.as<i32>();
        e->n_long = 0;
        for (i64 i = 0; i < n; i++) if (hrs[ord[i]].is_long) lg[e->n_long++] = ord[i];
        if (e->n_long > 0) HIP_TRY(hipMemcpyAsync(e->d_long.p, lg, (size_t)e->n_long * 4, hipMemcpyHostToDevice, s));
    }
    e->hp.dp_wg_mode = 0;
    memcpy(e->h_dp.p, &e->hp, sizeof(DevParams));
    HIP_TRY(hipMemcpyAsync(e->d_rs.p, e->h_rs.p, N * sizeof(ReadState), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(e->d_dp.p, e->h_dp.p, sizeof(DevParams), hipMemcpyHostToDevice, s));
    // (raw / seq may be device memory -- a batch made by tba_synth_generate: the kind is taken from the pointer)
    HIP_TRY(hipMemcpyAsync(e->d_raw.p, raw, (size_t)e->S_tot * raw_elem_bytes(raw_dtype), hipMemcpyDefault, s));
    HIP_TRY(hipMemcpyAsync(e->d_seq.p, seq, (size_t)e->seq_tot, hipMemcpyDefault, s));
    if (e->have_sv) HIP_TRY(hipMemcpyAsync(e->d_sv_in.p, sv_in, N * 32, hipMemcpyHostToDevice, s));
    if (e->have_samp)
        HIP_TRY(hipMemcpyAsync(e->d_samp.p, samp_ind, N * MAX_TS_POINTS * 8, hipMemcpyHostToDevice, s));
    if (stalls && stall_off[n] > 0)
        HIP_TRY(hipMemcpyAsync(e->d_stall.p, stall_ints, (size_t)stall_off[n] * 16, hipMemcpyHostToDevice, s));
    if (o->reverse_raw) { // once per upload, in stream order behind the copy
        const unsigned gr = (unsigned)std::min<i64>(std::max<i64>((z.max_raw / 2 + 1023) / 1024, 1), 64);
        RAW_DISPATCH(raw_dtype, (k_reverse_raw<RT><<<dim3(gr, (unsigned)n), 256, 0, s>>>(e->d_rs.as<ReadState>(), e->d_raw.as<RT>())));
        HIP_TRY(hipGetLastError());
    }
    e->have_batch = true;
    return 0;
}

extern "C" int tba_batch_upload(tba_engine *e, const tba_params *p, const tba_opts *o,
                                int64_t n_reads, const double *raw, const int64_t *raw_off,
                                const uint8_t *seq, const int64_t *seq_off, const double *sv_in,
                                const int32_t *sv_flags, const int64_t *samp_ind,
                                const int64_t *stall_ints, const int64_t *stall_off)
{
    int rc = tba_batch_upload_async(e, p, o, n_reads, raw, TBA_RAW_F64, raw_off, seq, seq_off, sv_in,
                                    sv_flags, samp_ind, stall_ints, stall_off);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int tba_set_num_events(tba_engine *e, const int64_t *num_events, int64_t n_reads)
{
    if (!e || n_reads < 0) return set_err(TBA_E_ARG, "bad arguments");
    if (!num_events) { e->ne_override.clear(); return 0; }
    e->ne_override.assign(num_events, num_events + n_reads);
    return 0;
}

// k_dp<8> or its 112-register build (k_dp.h): the latter when other engines run their kernels beside
// this one (streaming slots) and the batch has more than 0.04 samples per DP cell, i.e. event
// detection and normalisation, not the DP, are most of the work (RNA 3 kb: 0.087, DNA 10 kb: 0.018)
static bool dp_lowreg(const tba_engine *e)
{
    return e->n_sharing > 1 && e->dp_cells > 0 && (double)e->S_tot > 0.04 * e->dp_cells;
}
template <int CPL>
static void launch_dp_t(tba_engine *e, int mode)
{
#define DP_LAUNCH_ARGS e->d_rs.as<ReadState>(), e->d_dp.as<DevParams>(), mode, e->d_evm.as<double>(), \
        e->d_refm.as<double>(), e->d_refs.as<double>(), e->d_bst.as<i64>(), e->d_lo.as<i32>(), \
        e->d_hi.as<i32>(), \
        mode == DP_MAIN ? e->d_moves.as<unsigned char>() : e->d_smoves.as<unsigned char>(), \
        e->start_moves_stride, e->d_lastrow.as<double>(), nullptr
    if (CPL == 8 && mode == DP_MAIN && dp_lowreg(e))
        k_dp8_lowreg<<<dim3((unsigned)e->n_reads), dim3(64), 0, e->stream>>>(DP_LAUNCH_ARGS);
    else
        k_dp<CPL, false><<<dim3((unsigned)e->n_reads), dim3(64), 0, e->stream>>>(DP_LAUNCH_ARGS);
#undef DP_LAUNCH_ARGS
}
template <int CPL, int RPW>
static void launch_dp_multi_t(tba_engine *e)
{
    k_dp_multi<CPL, RPW><<<dim3((unsigned)((e->n_reads + RPW - 1) / RPW)), dim3(64), 0, e->stream>>>(
        e->d_rs.as<ReadState>(), e->n_reads, e->d_order.as<i32>(), e->d_dp.as<DevParams>(),
        e->d_evm.as<double>(), e->d_refm.as<double>(), e->d_refs.as<double>(), e->d_bst.as<i64>(),
        e->d_lo.as<i32>(), e->d_hi.as<i32>(), e->d_moves.as<unsigned char>(), e->d_lastrow.as<double>());
}
static void launch_dp_multi(tba_engine *e)
{
    const DpMultiClass c = dp_multi_class(e->hp.p.bandwidth);
    if (c.cpl == 8 && c.rpw == 4) launch_dp_multi_t<8, 4>(e);
    else if (c.cpl == 4 && c.rpw == 2) launch_dp_multi_t<4, 2>(e);
    else if (c.cpl == 8 && c.rpw == 2) launch_dp_multi_t<8, 2>(e);
    else if (c.cpl == 10 && c.rpw == 2) launch_dp_multi_t<10, 2>(e);
}
static void launch_dp(tba_engine *e, int cpl, int mode)
{
    switch (cpl) {
    case 4: launch_dp_t<4>(e, mode); break;
    case 5: launch_dp_t<5>(e, mode); break;
    case 8: launch_dp_t<8>(e, mode); break;
    case 12: launch_dp_t<12>(e, mode); break;
    case 16: launch_dp_t<16>(e, mode); break;
    case 24: launch_dp_t<24>(e, mode); break;
    case 32: launch_dp_t<32>(e, mode); break;
    case 48: launch_dp_t<48>(e, mode); break;
    default: break;
    }
}

static int enqueue_stages(tba_engine *e, int first, int last)
{
    if (!e || !e->have_batch) return set_err(TBA_E_STATE, "no batch uploaded");
    if (first < 0 || last > TBA_STAGE_RESCALE || first > last) return set_err(TBA_E_ARG, "bad stage range");
    if (first > 0 && !e->ran) return set_err(TBA_E_STATE, "stages before `first` have not been run or injected");
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const i64 n = e->n_reads;
    const tba_params &P = e->hp.p;
    ReadState *rs = e->d_rs.as<ReadState>();
    const DevParams *dp = e->d_dp.as<DevParams>();
    // starting from the top discards the state of a previous run of the same batch
    if (first == 0)
        HIP_TRY(hipMemcpyAsync(e->d_rs.p, e->h_rs.p, (size_t)n * sizeof(ReadState), hipMemcpyHostToDevice, s));
    const unsigned nb = (unsigned)n;
    const unsigned tpr = (unsigned)((n + 63) / 64); // blocks for thread-per-read kernels
    // workgroups per read of the (blocks, reads) kernels: 256 items per workgroup when the batch is
    // small (parallelism), up to 4096 when the reads alone fill the machine -- short-lived
    // workgroups cost more in launches than they win in balance (RNA, 10 k reads: 113 -> 107 ms)
    auto gx = [n](i64 items) {
        const i64 fine = (items + 255) / 256, coarse = (items + 4095) / 4096;
        const i64 want = (16384 + n - 1) / n; // enough workgroups in all for ~8 per CU-slot
        const i64 g = std::max<i64>(coarse, std::min<i64>(fine, want));
        return (unsigned)std::min<i64>(std::max<i64>(g, 1), 128);
    };
    const unsigned gS = gx(e->max_raw), gB = gx(e->max_B), gE = gx(e->max_raw / std::max<i64>(P.mean_obs_per_event, 1) + 1);
    int st = 0;
#define MARK() HIP_TRY(hipEventRecord(e->ev[st++], s))
#define ON(stage_) ((stage_) >= first && (stage_) <= last)
    const bool rna = P.use_t_test_seg != 0;
    const int rdt = e->raw_dtype;
    HIP_TRY(hipEventRecord(e->ev[15], s));                 // start of the sequence (the `total` bracket)
    // A full run forks the side stream here (-DTBA_NO_SIDE_STREAM / TBA_NO_SIDE_STREAM=1: everything on
    // the main stream, in this order); a partial run (stepwise API) stays on one stream.
#ifdef TBA_NO_SIDE_STREAM
    const bool side = false;
    e->last_side = false;
#else
    static const bool side_off = getenv("TBA_NO_SIDE_STREAM") != nullptr;
    const bool side = !side_off && first == TBA_STAGE_SEGMENT && last == TBA_STAGE_RESCALE && e->side_mode != 0 &&
                      (e->side_mode == 1 || (e->device < TBA_MAX_DEVICES &&
                       std::max(e->n_sharing, g_live_engines[e->device].load()) <= side_stream_max_engines()));
    e->last_side = side;
    if (side && !e->stream2) HIP_TRY(hipStreamCreate(&e->stream2)); // (created on first use: a stream takes a queue slot)
#endif
    hipStream_t s2 = side ? e->stream2 : s;
    if (side) {
        HIP_TRY(hipEventRecord(e->ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(s2, e->ev_fork, 0));
    }
    // caller-side preparation: ts.identify_stalls over the raw samples (its own scratch: it runs beside
    // event detection)
    HIP_TRY(hipEventRecord(e->ev_st0, s2));
    if (ON(TBA_STAGE_SEGMENT) && e->hp.o.detect_stalls) {
        double *csum = e->d_stall_csum.as<double>();
        u64 *bits = e->d_stall_bits.as<u64>();
        const unsigned gq = gx(e->max_raw / 8 + 1); // chunks of SI_T positions
        if (rdt == TBA_RAW_I16 && e->hp.o.stall_window_size <= SI_MAXW) { // exact integer sums: no cumulative sum in memory
            if (e->hp.o.stall_n_windows == 7) k_stall_metric_i16<7><<<dim3(gq, nb), 256, 0, s2>>>(rs, dp, e->d_raw.as<int16_t>(), bits);
            else k_stall_metric_i16<0><<<dim3(gq, nb), 256, 0, s2>>>(rs, dp, e->d_raw.as<int16_t>(), bits);
        } else {
        if (cs_reads_for(n) == 20) RAW_DISPATCH(rdt, (k_cumsum_scores<20, RT, 1><<<(unsigned)((n + 19) / 20), 256, 0, s2>>>(rs, n, dp, e->d_raw.as<RT>(), csum)));
        else RAW_DISPATCH(rdt, (k_cumsum_scores<32, RT, 1><<<(unsigned)((n + 31) / 32), 256, 0, s2>>>(rs, n, dp, e->d_raw.as<RT>(), csum)));
        if (e->n_long > 0) RAW_DISPATCH(rdt, (k_cumsum_scores_long<RT, 1><<<(unsigned)e->n_long, 256, 0, s2>>>(rs, e->d_long.as<i32>(), dp, e->d_raw.as<RT>(), csum)));
        if (e->hp.o.stall_n_windows == 7) k_stall_metric<7><<<dim3(gq, nb), 256, 0, s2>>>(rs, dp, csum, bits);
        else k_stall_metric<0><<<dim3(gq, nb), 256, 0, s2>>>(rs, dp, csum, bits);
        }
        k_stall_runs<<<dim3(gx(e->max_raw / 64 + 1), nb), 256, 0, s2>>>(rs, dp, bits, e->d_stall.as<i64>());
        k_stall_merge<<<tpr, 64, 0, s2>>>(rs, n, dp, e->d_stall.as<i64>());
    }
    HIP_TRY(hipEventRecord(e->ev_st1, s2));
    if (side) {
        HIP_TRY(hipEventRecord(e->ev_stalls, s2));
        // the expected levels need the sequence and the model only
        k_ref_levels<<<dim3(gB, nb), 256, 0, s2>>>(rs, dp, e->d_seq.as<uint8_t>(), e->d_kmeans.as<double>(), e->d_ksds.as<double>(), e->d_refm.as<double>(), e->d_refs.as<double>(), 1);
        HIP_TRY(hipEventRecord(e->ev_levels, s2));
    }
    MARK(); // 0 normalize
    const bool fused_scores = 2 * P.running_stat_width <= 64; // cumsum + scores in one kernel
    // DNA defaults: the scores never reach memory (k_detect.h); what that form leaves (flagged reads)
    // goes through the kernels below as before
#ifdef TBA_NO_FUSED_DETECT
    const bool fused_detect = false;
#else
    const bool fused_detect = !rna && 2 * P.running_stat_width <= DT_W2MAX && P.min_obs_per_base == 3;
#endif
    // A handful of reads cannot hide the scan's serial chain behind each other: k_detect /
    // k_cumsum_scores pay a pipeline step (barrier, memory round trip, greedy: ~7 us) per 128 samples
    // whatever the batch, 5 ms for a 10 kb read; a workgroup per read (k_long.h: the step is 1 856
    // dependent adds long) does the same in 0.5 ms.  (resquiggle_read, a batch of one: 20.4 -> 16 ms.)
    const bool wg_scan = !rna && fused_scores && n <= e->small_batch && (size_t)n * 4 <= e->d_order.cap;
#ifdef TBA_NO_FUSED_DETECT
    const bool fused_tt = false;
#else
    const bool fused_tt = rna && P.min_obs_per_base == 6 && P.running_stat_width <= TT_MAXW; // RNA defaults: radius 5
#endif
    const int only_flagged = (fused_detect && !wg_scan) || fused_tt ? 1 : 0;
    // (with k_detect on the way its loader writes the normalised signal: k_normalize only finds the
    // scale values then, and normalises the long reads, which k_detect leaves to k_long.h)
    if (ON(TBA_STAGE_SEGMENT) && !rna)
        RAW_DISPATCH(rdt, (k_normalize<RT><<<nb, SEL_NT, 0, s>>>(rs, dp, e->d_raw.as<RT>(), e->d_norm.as<double>(), e->d_sv_in.as<double>(), 0, fused_detect && !wg_scan ? 2 : 1)));
    MARK(); // 1 cumsum
    if (ON(TBA_STAGE_SEGMENT) && !rna && wg_scan) {
        k_cumsum_scores_long<double, 0><<<nb, 256, 0, s>>>(rs, e->d_order.as<i32>(), dp, e->d_norm.as<double>(), e->d_score.as<double>());
    } else if (ON(TBA_STAGE_SEGMENT) && !rna) {
        if (fused_detect) {
            RAW_DISPATCH(rdt, (k_detect<2, RT><<<(unsigned)((n + DT_READS - 1) / DT_READS), 256, 0, s>>>(rs, n, dp, e->d_raw.as<RT>(), e->d_norm.as<double>(), e->d_csum.as<double>(), e->d_score.as<double>(), e->S_tot)));
            k_pick<<<nb, SEL_NT, 0, s>>>(rs, dp, e->d_csum.as<double>(), e->d_score.as<double>(), e->d_cpts.as<i64>(), 0);
        }
        if (fused_scores) {
            if (cs_reads_for(n) == 20) k_cumsum_scores<20><<<(unsigned)((n + 19) / 20), 256, 0, s>>>(rs, n, dp, e->d_norm.as<double>(), e->d_score.as<double>(), only_flagged);
            else k_cumsum_scores<32><<<(unsigned)((n + 31) / 32), 256, 0, s>>>(rs, n, dp, e->d_norm.as<double>(), e->d_score.as<double>(), only_flagged);
            if (e->n_long > 0) k_cumsum_scores_long<double, 0><<<(unsigned)e->n_long, 256, 0, s>>>(rs, e->d_long.as<i32>(), dp, e->d_norm.as<double>(), e->d_score.as<double>());
        }
        else k_cumsum<<<tpr, 64, 0, s>>>(rs, n, e->d_norm.as<double>(), e->d_csum.as<double>());
    }
    MARK(); // 2 scores
    if (ON(TBA_STAGE_SEGMENT)) {
        if (!rna) { if (!fused_scores) k_scores_dna<<<dim3(gS, nb), 256, 0, s>>>(rs, dp, e->d_csum.as<double>(), e->d_score.as<double>()); }
        else {
            if (fused_tt) {
                if (P.running_stat_width == 12) RAW_DISPATCH(rdt, (k_detect_tt<5, 12, RT><<<nb, SEL_NT, 0, s>>>(rs, dp, e->d_raw.as<RT>(), e->d_csum.as<double>(), e->d_score.as<double>())));
                else RAW_DISPATCH(rdt, (k_detect_tt<5, 0, RT><<<nb, SEL_NT, 0, s>>>(rs, dp, e->d_raw.as<RT>(), e->d_csum.as<double>(), e->d_score.as<double>())));
                k_pick<<<nb, SEL_NT, 0, s>>>(rs, dp, e->d_csum.as<double>(), e->d_score.as<double>(), e->d_cpts.as<i64>(), 1);
            }
            RAW_DISPATCH(rdt, (k_scores_ttest<RT><<<dim3(gS, nb), 256, 0, s>>>(rs, dp, e->d_raw.as<RT>(), e->d_score.as<double>(), only_flagged)));
        }
    }
    MARK(); // 3 peaks
    if (ON(TBA_STAGE_SEGMENT)) {
        launch_peaks(P.min_obs_per_base, nb, s, rs, dp, e->d_score.as<double>(), e->d_state.as<unsigned char>(), e->d_csum.as<double>(), e->d_cpts.as<i64>(), rna ? 1 : 0, only_flagged,
                     rna ? TBA_ED_FORM_TTEST_PEAKS : wg_scan ? TBA_ED_FORM_WG_SCAN_PEAKS : TBA_ED_FORM_SCORES_PEAKS);
        if (side) HIP_TRY(hipStreamWaitEvent(s, e->ev_stalls, 0)); // the stall intervals (and nothing else of the side stream)
        if (e->any_stall) k_remove_stalls<<<nb, SEL_NT, 0, s>>>(rs, n, e->d_stall.as<i64>(), e->d_cpts.as<i64>(), e->d_csum.as<double>());
        if (rna) { // RNA normalises after event detection (segment_signal, resquiggle.py:1073-1098)
            RAW_DISPATCH(rdt, (k_event_means<RT, 1280><<<dim3(gE, nb), 256, 0, s>>>(rs, dp, e->d_raw.as<RT>(), e->d_cpts.as<i64>(), e->d_evm.as<double>(), 1)));
            k_rna_event_scale<<<nb, SEL_NT, 0, s>>>(rs, dp, e->d_evm.as<double>());
            RAW_DISPATCH(rdt, (k_normalize<RT><<<nb, SEL_NT, 0, s>>>(rs, dp, e->d_raw.as<RT>(), e->d_norm.as<double>(), e->d_sv_in.as<double>(), 1, 1)));
        }
    }
    MARK(); // 4 event means
    if (ON(TBA_STAGE_EVENT_MEANS)) {
        // long events (RNA: mean_obs_per_event 15): the wide staging slice
        if (P.mean_obs_per_event >= 10) k_event_means<double, 1280><<<dim3(gE, nb), 256, 0, s>>>(rs, dp, e->d_norm.as<double>(), e->d_cpts.as<i64>(), e->d_evm.as<double>(), 0);
        else k_event_means<double><<<dim3(gE, nb), 256, 0, s>>>(rs, dp, e->d_norm.as<double>(), e->d_cpts.as<i64>(), e->d_evm.as<double>(), 0);
    }
    MARK(); // 5 ref levels
    if (side) {                                                    // (computed on the side stream)
        HIP_TRY(hipStreamWaitEvent(s, e->ev_levels, 0));
        k_seq_status<<<tpr, 64, 0, s>>>(rs, n);
    } else if (ON(TBA_STAGE_REF_LEVELS))
        k_ref_levels<<<dim3(gB, nb), 256, 0, s>>>(rs, dp, e->d_seq.as<uint8_t>(), e->d_kmeans.as<double>(), e->d_ksds.as<double>(), e->d_refm.as<double>(), e->d_refs.as<double>());
    MARK(); // 6 start dp (+7 start tb): find_seq_start_in_events, first try then retry
    if (ON(TBA_STAGE_START)) {
        k_path0<<<tpr, 64, 0, s>>>(rs, n, dp);
        launch_dp(e, cpl_class(P.start_bw), DP_START_TRY);
        k_start_tb<<<(unsigned)((n + TB_LANES - 1) / TB_LANES), TB_LANES, 0, s>>>(rs, n, dp, DP_START_TRY, e->d_evm.as<double>(), e->d_refm.as<double>(), e->d_refs.as<double>(), e->d_smoves.as<unsigned char>(), e->start_moves_stride, e->d_readtb.as<i64>(), e->d_startvals.as<double>());
    }
    MARK(); // 7
    if (ON(TBA_STAGE_START)) {
        // the retry of the few reads whose first try failed: one workgroup per read (k_dp_wg.h)
        const int wcpl = P.start_n_bases <= WG_MAX_ROWS ? dp_wg_cpl(P.start_save_bw) : 0;
#define WG_ARGS rs, dp, e->d_evm.as<double>(), e->d_refm.as<double>(), e->d_refs.as<double>(), e->d_smoves.as<unsigned char>(), e->start_moves_stride, e->d_lastrow.as<double>()
        if (wcpl == 4) k_dp_wg<4><<<nb, 256, 0, s>>>(WG_ARGS);
        else if (wcpl == 8) k_dp_wg<8><<<nb, 256, 0, s>>>(WG_ARGS);
        else if (wcpl == 12) k_dp_wg<12><<<nb, 256, 0, s>>>(WG_ARGS);
        else launch_dp(e, cpl_class(P.start_save_bw), DP_START_RETRY);
#undef WG_ARGS
        k_start_tb<<<(unsigned)((n + TB_LANES - 1) / TB_LANES), TB_LANES, 0, s>>>(rs, n, dp, DP_START_RETRY, e->d_evm.as<double>(), e->d_refm.as<double>(), e->d_refs.as<double>(), e->d_smoves.as<unsigned char>(), e->start_moves_stride, e->d_readtb.as<i64>(), e->d_startvals.as<double>());
    }
    MARK(); // 8 prep
    if (ON(TBA_STAGE_ASSIGN)) {
        k_prep<<<tpr, 64, 0, s>>>(rs, n, dp, e->d_bst.as<i64>(), e->d_lo.as<i32>(), e->d_hi.as<i32>());
        k_scan_arena<0><<<1, 256, 0, s>>>(rs, n, e->moves_arena);
    }
    MARK(); // 9 main dp
    if (ON(TBA_STAGE_ASSIGN)) {
        const int cls[] = {4, 5, 8, 12, 16, 24, 32, 48};
        for (int c : cls) launch_dp(e, c, DP_MAIN);
        launch_dp_multi(e); // narrow adaptive bands: several reads per wavefront
        if (e->wide_w) // a static band wider than every class is possible in this batch
            k_dp_wide<<<WIDE_BLOCKS, 64, 0, s>>>(rs, n, dp, e->d_evm.as<double>(), e->d_refm.as<double>(), e->d_refs.as<double>(), e->d_bst.as<i64>(), e->d_moves.as<unsigned char>(), e->d_wide.as<double>(), e->wide_w);
    }
    MARK(); // 10 main tb
    if (ON(TBA_STAGE_ASSIGN)) {
#ifndef TBA_NO_TB_PAR
        // rows of a read over several lanes (k_tb_par.h): 16 lanes per read when the reads fill the
        // machine, a wavefront per read for small batches and for the long reads; what it leaves
        // (static bands, failed verification) is walked by the lane-per-read kernels below
#ifdef TBA_TB_POISON
        // (experiment build: read_tb holds the previous run's finished paths -- take them away, so that nothing
        // stale can compare equal; profiles/r06_traceback_rootcause.txt)
        HIP_TRY(hipMemsetAsync(e->d_readtb.p, 0xFF, (size_t)(e->B_tot + e->n_reads) * 8, s));
#endif
#ifdef TBA_TB_B2
        {   // (experiment build: d_readtb is three arrays long -- read_tb, phase B's stores, phase B's entry states)
            const i64 off_words = (i64)(e->B_tot + e->n_reads);
            HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(TBA_TB_B2_OFF), &off_words, sizeof(off_words)));
            HIP_TRY(hipMemsetAsync(e->d_readtb.as<i64>() + off_words, 0xFF, (size_t)off_words * 16, s));
        }
#endif
        if (n > e->tb_wave_below) k_main_tb_par<16><<<(unsigned)((n + 3) / 4), 64, 0, s>>>(rs, n, nullptr, dp, e->d_moves.as<unsigned char>(), e->d_bst.as<i64>(), e->d_readtb.as<i64>());
        else k_main_tb_par<64><<<nb, 64, 0, s>>>(rs, n, nullptr, dp, e->d_moves.as<unsigned char>(), e->d_bst.as<i64>(), e->d_readtb.as<i64>());
        if (e->n_long > 0 && n > e->tb_wave_below) k_main_tb_par<64><<<(unsigned)e->n_long, 64, 0, s>>>(rs, e->n_long, e->d_long.as<i32>(), dp, e->d_moves.as<unsigned char>(), e->d_bst.as<i64>(), e->d_readtb.as<i64>());
#ifndef TBA_NO_TB_VERIFY
        // behind the kernel boundary: the first block of rows under every chunk top walked again, compare
        // only; what disagrees is the serial kernels' (counted: TBA_GET_TB_VERIFY_FAIL), the rest is trimmed
        if (n > e->tb_wave_below) k_tb_par_verify<16><<<(unsigned)((n + 3) / 4), 64, 0, s>>>(rs, n, nullptr, dp, e->d_moves.as<unsigned char>(), e->d_bst.as<i64>(), e->d_readtb.as<i64>());
        else k_tb_par_verify<64><<<nb, 64, 0, s>>>(rs, n, nullptr, dp, e->d_moves.as<unsigned char>(), e->d_bst.as<i64>(), e->d_readtb.as<i64>());
        if (e->n_long > 0 && n > e->tb_wave_below) k_tb_par_verify<64><<<(unsigned)e->n_long, 64, 0, s>>>(rs, e->n_long, e->d_long.as<i32>(), dp, e->d_moves.as<unsigned char>(), e->d_bst.as<i64>(), e->d_readtb.as<i64>());
#endif
#endif
        k_main_tb<<<(unsigned)((n + TB_LANES - 1) / TB_LANES), TB_LANES, 0, s>>>(rs, n, dp, e->d_moves.as<unsigned char>(), e->d_bst.as<i64>(), e->d_readtb.as<i64>());
        if (e->n_long > 0) k_main_tb_long<<<(unsigned)e->n_long, 64, 0, s>>>(rs, e->d_long.as<i32>(), dp, e->d_moves.as<unsigned char>(), e->d_bst.as<i64>(), e->d_readtb.as<i64>());
        k_tb_gather<<<dim3(gB, nb), 256, 0, s>>>(rs, e->d_cpts.as<i64>(), e->d_readtb.as<i64>(), e->d_dpsegs.as<i64>());
    }
    MARK(); // 11 skip resolve
    if (ON(TBA_STAGE_SKIP)) {
        // window queues of the wave-per-window kernels: counters + three (read, window) lists
        const i64 qcap = n * 32 + 4096;
        i64 *skipq = e->d_skipq.as<i64>();
        i32 *lists = (i32 *)(skipq + 8);
        HIP_TRY(hipMemsetAsync(skipq, 0, 64, s));
        k_skip_plan<<<nb, 64, 0, s>>>(rs, n, dp, e->d_dpsegs.as<i64>(), e->d_segs.as<i64>(), e->d_win.as<i64>(), skipq, lists, qcap);
        k_scan_arena<1><<<1, 256, 0, s>>>(rs, n, e->skip_arena);
#define SKIP_WAVE_ARGS(c_) rs, dp, e->d_norm.as<double>(), e->d_refm.as<double>(), e->d_refs.as<double>(), e->d_dpsegs.as<i64>(), e->d_segs.as<i64>(), e->d_win.as<i64>(), skipq, lists + 2 * qcap * (c_), qcap
        if (P.raw_min_obs_per_base > 1) { // (k_skip_plan queues nothing otherwise)
            // The three classes own disjoint windows and each is a queue drained by lone wavefronts whose time is
            // one lane's stay recurrence: a kernel is as long as its slowest chain of windows, not as its work.
            // With the side stream the middle class runs beside the big one instead of behind it: 5.0 -> 4.1 ms for
            // the three on cfg4 (tools/skip_timeline.sh; before the kernels' LDS diet 7.5 -> 5.7; all three at once
            // on three streams: 3.7-4.1, no better for the stage).  TBA_SKIP_FORK=0: one after the other, as before
            // round 6.
            static const bool fork_off = getenv("TBA_SKIP_FORK") != nullptr && getenv("TBA_SKIP_FORK")[0] == '0';
            const bool fork = side && !fork_off;
            hipStream_t sw = fork ? s2 : s;
            if (fork) {
                HIP_TRY(hipEventRecord(e->ev_skip0, s));
                HIP_TRY(hipStreamWaitEvent(s2, e->ev_skip0, 0));
            }
            k_skip_dp_wave<SKIP_LEN_B, SKIP_BITS_B, 2><<<512, 64, 0, s>>>(SKIP_WAVE_ARGS(2));
            k_skip_dp_wave<SKIP_LEN_M, SKIP_BITS_M, 1><<<1024, 64, 0, sw>>>(SKIP_WAVE_ARGS(1));
            k_skip_dp_wave<SKIP_LEN_S, SKIP_BITS_S, 0><<<2048, 64, 0, s>>>(SKIP_WAVE_ARGS(0));
            if (fork) {
                HIP_TRY(hipEventRecord(e->ev_skip1, s2));
                HIP_TRY(hipStreamWaitEvent(s, e->ev_skip1, 0));
            }
        }
#undef SKIP_WAVE_ARGS
        // (raw_min_obs_per_base == 1, DNA: the small windows out of LDS -- k_tail.h)
        (e->hp.p.raw_min_obs_per_base == 1 ? k_skip_dp<true> : k_skip_dp<false>)<<<nb, 64, 0, s>>>(rs, dp, e->d_norm.as<double>(), e->d_refm.as<double>(), e->d_refs.as<double>(), e->d_dpsegs.as<i64>(), e->d_segs.as<i64>(), e->d_win.as<i64>(), e->d_dscr.as<double>());
    }
    MARK(); // 12 theil-sen
    if (ON(TBA_STAGE_RESCALE)) {
        k_theil_sen<<<nb, SEL_NT, 0, s>>>(rs, dp, e->d_norm.as<double>(), e->d_segs.as<i64>(), e->d_refm.as<double>(), e->have_samp || e->hp.o.device_subsample ? e->d_samp.as<i64>() : nullptr, e->d_csum.as<double>(), e->d_score.as<double>());
    }
    MARK(); // 13 rescale + score
    if (ON(TBA_STAGE_RESCALE)) {
        if (e->hp.o.skip_norm_out)
            k_rescale_absz<false><<<dim3(gB, nb), 256, 0, s>>>(rs, dp, e->d_norm.as<double>(), nullptr, e->d_segs.as<i64>(), e->d_refm.as<double>(), e->d_refs.as<double>(), e->d_absz.as<double>());
        else
            k_rescale_absz<true><<<dim3(gB, nb), 256, 0, s>>>(rs, dp, e->d_norm.as<double>(), e->d_norm_out.as<double>(), e->d_segs.as<i64>(), e->d_refm.as<double>(), e->d_refs.as<double>(), e->d_absz.as<double>());
        k_final_score<<<nb, 64, 0, s>>>(rs, n, e->d_absz.as<double>());
    }
    MARK(); // 14 end
#undef MARK
#undef ON
    HIP_TRY(hipGetLastError());
    e->ran = true;
    e->finished = last == TBA_STAGE_RESCALE;
    return 0;
}

extern "C" int tba_batch_enqueue(tba_engine *e) { return enqueue_stages(e, TBA_STAGE_SEGMENT, TBA_STAGE_RESCALE); }

// The kernels this engine enqueues next start only after `other`'s last enqueued kernel sequence has
// finished (its transfers are not waited for): a streaming caller keeps the kernel sequences of its
// slots back to back instead of interleaved, while their copies still overlap.
extern "C" int tba_batch_wait_for(tba_engine *e, tba_engine *other)
{
    if (!e || !other) return set_err(TBA_E_ARG, "engine is NULL");
    if (!other->ran || e == other) return 0; // nothing enqueued there yet
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamWaitEvent(e->stream, other->ev[14], 0));
    return 0;
}

extern "C" int tba_batch_run_stages(tba_engine *e, int first_stage, int last_stage)
{
    int rc = enqueue_stages(e, first_stage, last_stage);
    if (rc) return rc;
    return tba_batch_sync(e);
}

// inject stage inputs of the uploaded batch (stepwise API): see include/tombo_amd.h
extern "C" int tba_batch_put(tba_engine *e, int what, const void *data, int64_t bytes,
                             const int64_t *per_read)
{
    if (!e || !e->have_batch || !data) return set_err(TBA_E_STATE, "no batch uploaded");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const size_t N = (size_t)e->n_reads;
    auto put = [&](DevBuf &b, size_t cap_bytes) -> int {
        if ((size_t)bytes > cap_bytes) return set_err(TBA_E_ARG, "input larger than the batch buffer");
        HIP_TRY(hipMemcpy(b.p, data, (size_t)bytes, hipMemcpyHostToDevice));
        return 0;
    };
    std::vector<ReadState> rs(N);
    // the first injection of a fresh batch starts from the uploaded state
    if (!e->ran) HIP_TRY(hipMemcpy(e->d_rs.p, e->h_rs.p, N * sizeof(ReadState), hipMemcpyHostToDevice));
    e->ran = true;
    HIP_TRY(hipMemcpy(rs.data(), e->d_rs.p, N * sizeof(ReadState), hipMemcpyDeviceToHost));
    int rc = 0;
    switch (what) {
    case TBA_PUT_VALID_CPTS: // per_read[i] = number of change points of read i
        if (!per_read) return set_err(TBA_E_ARG, "per_read counts required");
        // the kernels index the signal with these values: strictly increasing inside [0, n_raw]
        for (size_t i = 0; i < N; i++) {
            if (per_read[i] > rs[i].num_events || per_read[i] < 2) return set_err(TBA_E_ARG, "change point count outside the reserved space");
            if ((size_t)(rs[i].ev_off + per_read[i]) * 8 > (size_t)bytes) return set_err(TBA_E_ARG, "change point array shorter than the counts");
            const i64 *c = (const i64 *)data + rs[i].ev_off;
            for (i64 k = 0; k < per_read[i]; k++)
                if (c[k] < 0 || c[k] > rs[i].n_raw || (k > 0 && c[k] <= c[k - 1]))
                    return set_err(TBA_E_ARG, "change points must be strictly increasing inside [0, n_raw]");
        }
        rc = put(e->d_cpts, (size_t)e->E_tot * 8);
        for (size_t i = 0; i < N && !rc; i++) { rs[i].n_cpts = per_read[i]; rs[i].n_ev = per_read[i] - 1; }
        break;
    case TBA_PUT_EVENT_MEANS: rc = put(e->d_evm, (size_t)e->E_tot * 8); break;
    case TBA_PUT_NORM: rc = put(e->d_norm, (size_t)e->S_tot * 8); break;
    case TBA_PUT_REF_MEANS: rc = put(e->d_refm, (size_t)e->B_tot * 8); break;
    case TBA_PUT_REF_SDS: rc = put(e->d_refs, (size_t)e->B_tot * 8); break;
    case TBA_PUT_DP_SEGS: // per_read[2i] = read_start_rel_to_raw, per_read[2i+1] = trimmed signal length
        if (!per_read) return set_err(TBA_E_ARG, "per_read (read_start, norm_len) required");
        if ((size_t)bytes < (size_t)(e->B_tot + e->n_reads) * 8) return set_err(TBA_E_ARG, "segment array shorter than the batch");
        for (size_t i = 0; i < N; i++) { // boundaries index the signal: non-decreasing inside [0, norm_len]
            const i64 rstart = per_read[2 * i], nl = per_read[2 * i + 1];
            if (rstart < 0 || nl < 0 || rstart + nl > rs[i].n_raw) return set_err(TBA_E_ARG, "segments outside the signal");
            const i64 *sg = (const i64 *)data + rs[i].seg_off;
            for (i64 k = 0; k <= rs[i].B; k++)
                if (sg[k] < 0 || sg[k] > nl || (k > 0 && sg[k] < sg[k - 1]))
                    return set_err(TBA_E_ARG, "segment boundaries must be non-decreasing inside [0, norm_len]");
        }
        rc = put(e->d_dpsegs, (size_t)(e->B_tot + e->n_reads) * 8);
        for (size_t i = 0; i < N && !rc; i++) {
            rs[i].read_start = rs[i].dp_read_start = per_read[2 * i];
            rs[i].norm_len = per_read[2 * i + 1];
        }
        break;
    case TBA_PUT_START_STATE: // per_read[i]: 4 = force the static whole-read path
        if (!per_read) return set_err(TBA_E_ARG, "per_read states required");
        for (size_t i = 0; i < N; i++) rs[i].start_state = (i32)per_read[i];
        break;
    default: return set_err(TBA_E_ARG, "unknown TBA_PUT_* selector");
    }
    if (rc) return rc;
    HIP_TRY(hipMemcpy(e->d_rs.p, rs.data(), N * sizeof(ReadState), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int tba_batch_sync(tba_engine *e)
{
    if (!e) return set_err(TBA_E_ARG, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->ran) {
        memset(e->stage_ms, 0, sizeof(e->stage_ms));
        for (int i = 0; i < 14; i++) (void)hipEventElapsedTime(&e->stage_ms[i], e->ev[i], e->ev[i + 1]);
        (void)hipEventElapsedTime(&e->stage_ms[14], e->ev_st0, e->ev_st1); // stall detection (side stream: overlaps the stages above)
        (void)hipEventElapsedTime(&e->stage_ms[15], e->ev[15], e->ev[14]);
    }
    return 0;
}

extern "C" int tba_batch_run(tba_engine *e)
{
    int rc = tba_batch_enqueue(e);
    if (rc) return rc;
    return tba_batch_sync(e);
}

extern "C" int tba_batch_download(tba_engine *e, int32_t *status, int64_t *segs,
                                  int64_t *read_start_rel_to_raw, double *norm_signal,
                                  int64_t *norm_len, double *scale_values,
                                  double *sig_match_score, int32_t *norm_params_changed)
{
    if (!e || !e->ran) return set_err(TBA_E_STATE, "no batch has been run");
    if (norm_signal && e->hp.o.skip_norm_out)
        return set_err(TBA_E_STATE, "the batch was run with skip_norm_out: there is no normalised signal to download");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const size_t N = (size_t)e->n_reads;
    std::vector<ReadState> rs(N);
    HIP_TRY(hipMemcpy(rs.data(), e->d_rs.p, N * sizeof(ReadState), hipMemcpyDeviceToHost));
    if (segs) HIP_TRY(hipMemcpy(segs, e->d_segs.p, (size_t)(e->B_tot + e->n_reads) * 8, hipMemcpyDeviceToHost));
    if (norm_signal) HIP_TRY(hipMemcpy(norm_signal, e->d_norm_out.p, (size_t)e->S_tot * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) {
        const ReadState &r = rs[i];
        if (status) status[i] = r.status;
        if (read_start_rel_to_raw) read_start_rel_to_raw[i] = r.read_start;
        if (norm_len) norm_len[i] = r.status == TBA_OK ? r.norm_len : 0;
        if (scale_values) {
            scale_values[4 * i + 0] = r.shift; scale_values[4 * i + 1] = r.scale;
            scale_values[4 * i + 2] = r.has_lims ? r.lower : NAN;
            scale_values[4 * i + 3] = r.has_lims ? r.upper : NAN;
        }
        if (sig_match_score) sig_match_score[i] = r.score;
        if (norm_params_changed) norm_params_changed[i] = r.changed;
    }
    return 0;
}

// per-read records + int32 boundaries, packed on the device so that the download is a few
// contiguous copies (grid: (blocks, reads))
__global__ __launch_bounds__(256) void k_pack_results(const ReadState *rs, const i64 *segs,
    tba_read_result *res, i32 *segs32)
{
    const ReadState &r = rs[blockIdx.y];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        tba_read_result o;
        o.status = r.status; o.norm_params_changed = r.changed;
        o.read_start_rel_to_raw = r.read_start;
        o.norm_len = r.status == TBA_OK ? r.norm_len : 0;
        o.shift = r.shift; o.scale = r.scale;
        o.lower_lim = r.has_lims ? r.lower : NAN;
        o.upper_lim = r.has_lims ? r.upper : NAN;
        o.sig_match_score = r.score;
        res[blockIdx.y] = o;
    }
    if (segs32 == nullptr) return;
    const i64 *sg = segs + r.seg_off;
    i32 *o32 = segs32 + r.seg_off;
    const bool ok = r.status == TBA_OK;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i <= r.B; i += (i64)gridDim.x * 256)
        o32[i] = ok ? (i32)sg[i] : 0;
}

extern "C" int tba_batch_download_async(tba_engine *e, tba_read_result *results, int32_t *segs32,
                                        int64_t *segs64, double *norm_signal)
{
    if (!e || !e->ran) return set_err(TBA_E_STATE, "no batch has been run");
    if (norm_signal && e->hp.o.skip_norm_out)
        return set_err(TBA_E_STATE, "the batch was run with skip_norm_out: there is no normalised signal to download");
    if (segs32 && e->max_raw > 0x7fffffffll) return set_err(TBA_E_ARG, "signal too long for int32 boundaries");
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const size_t N = (size_t)e->n_reads;
    const unsigned gB = (unsigned)std::min<i64>(std::max<i64>((e->max_B + 1 + 255) / 256, 1), 64);
    if (results || segs32) {
        k_pack_results<<<dim3(segs32 ? gB : 1, (unsigned)N), 256, 0, s>>>(e->d_rs.as<ReadState>(),
            e->d_segs.as<i64>(), e->d_res.as<tba_read_result>(), segs32 ? e->d_segs32.as<i32>() : nullptr);
        HIP_TRY(hipGetLastError());
    }
    if (results) HIP_TRY(hipMemcpyAsync(results, e->d_res.p, N * sizeof(tba_read_result), hipMemcpyDeviceToHost, s));
    if (segs32) HIP_TRY(hipMemcpyAsync(segs32, e->d_segs32.p, (size_t)(e->B_tot + e->n_reads) * 4, hipMemcpyDeviceToHost, s));
    if (segs64) HIP_TRY(hipMemcpyAsync(segs64, e->d_segs.p, (size_t)(e->B_tot + e->n_reads) * 8, hipMemcpyDeviceToHost, s));
    if (norm_signal) HIP_TRY(hipMemcpyAsync(norm_signal, e->d_norm_out.p, (size_t)e->S_tot * 8, hipMemcpyDeviceToHost, s));
    return 0;
}

extern "C" int tba_batch_query(tba_engine *e)
{
    if (!e) return set_err(TBA_E_ARG, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    const hipError_t rc = hipStreamQuery(e->stream);
    if (rc == hipSuccess) return 0;
    if (rc == hipErrorNotReady) return 1;
    return set_err(TBA_E_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(rc));
}

extern "C" int tba_batch_get(tba_engine *e, int what, void *out, int64_t out_bytes)
{
    if (!e || !e->ran || !out) return set_err(TBA_E_STATE, "no batch has been run");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const size_t N = (size_t)e->n_reads;
    auto copy = [&](const DevBuf &b, size_t bytes) -> int {
        if ((size_t)out_bytes < bytes) return set_err(TBA_E_ARG, "output buffer too small");
        HIP_TRY(hipMemcpy(out, b.p, bytes, hipMemcpyDeviceToHost));
        return 0;
    };
    std::vector<ReadState> rs;
    auto fetch_rs = [&]() -> int {
        rs.resize(N);
        HIP_TRY(hipMemcpy(rs.data(), e->d_rs.p, N * sizeof(ReadState), hipMemcpyDeviceToHost));
        return 0;
    };
#ifdef TBA_TB_B2
    if (what == 97) { // phase B's second and third array (experiment build)
        const size_t words = (size_t)(e->B_tot + e->n_reads);
        HIP_TRY(hipMemcpy(out, e->d_readtb.as<i64>() + words, std::min((size_t)out_bytes, words * 16), hipMemcpyDeviceToHost));
        return 0;
    }
#endif
    switch (what) {
    case TBA_GET_VALID_CPTS: return copy(e->d_cpts, (size_t)e->E_tot * 8);
    case TBA_GET_EVENT_MEANS: return copy(e->d_evm, (size_t)e->E_tot * 8);
    case TBA_GET_SEG_NORM: return copy(e->d_norm, (size_t)e->S_tot * 8);
    case TBA_GET_ED_TAKEN_POS: return copy(e->d_score, (size_t)e->S_tot * 8);
    case TBA_GET_BAND_STARTS: return copy(e->d_bst, (size_t)e->B_tot * 8);
    case TBA_GET_READ_TB: return copy(e->d_readtb, (size_t)(e->B_tot + e->n_reads) * 8);
    case TBA_GET_DP_SEGS: return copy(e->d_dpsegs, (size_t)(e->B_tot + e->n_reads) * 8);
    case TBA_GET_LAST_ROW: return copy(e->d_lastrow, N * TBA_MAX_BAND * 8);
    case TBA_GET_REF_MEANS: return copy(e->d_refm, (size_t)e->B_tot * 8);
    case TBA_GET_REF_SDS: return copy(e->d_refs, (size_t)e->B_tot * 8);
    case TBA_GET_SEGS: return copy(e->d_segs, (size_t)(e->B_tot + e->n_reads) * 8);
    case TBA_GET_SAMP_IND: return copy(e->d_samp, N * MAX_TS_POINTS * 8);
    case TBA_GET_STALL_INTS:
        if (!e->any_stall) return set_err(TBA_E_STATE, "the batch has no stall intervals");
        // (the caller sizes `out` by the intervals in use: max(STALL_OFF + N_STALL))
        return copy(e->d_stall, std::min((size_t)e->n_stall_cap * 16, (size_t)out_bytes));
    case TBA_GET_KERNEL_MS:
        if ((size_t)out_bytes < sizeof(e->stage_ms)) return set_err(TBA_E_ARG, "output buffer too small");
        memcpy(out, e->stage_ms, sizeof(e->stage_ms));
        return 0;
    default: break;
    }
    if (int rc = fetch_rs()) return rc;
    if (what == TBA_GET_START_FAIL) {
        if ((size_t)out_bytes < N * 4) return set_err(TBA_E_ARG, "output buffer too small");
        for (size_t i = 0; i < N; i++) ((i32 *)out)[i] = rs[i].pad0;
        return 0;
    }
    if (what == TBA_GET_STATUS) {
        if ((size_t)out_bytes < N * 4) return set_err(TBA_E_ARG, "output buffer too small");
        for (size_t i = 0; i < N; i++) ((i32 *)out)[i] = rs[i].status;
        return 0;
    }
    if (what == TBA_GET_ED_FUSED) { // (by the form the kernels recorded, not by the absence of a flag)
        if ((size_t)out_bytes < N * 4) return set_err(TBA_E_ARG, "output buffer too small");
        for (size_t i = 0; i < N; i++)
            ((i32 *)out)[i] = rs[i].ed_form == TBA_ED_FORM_DETECT_PICK || rs[i].ed_form == TBA_ED_FORM_DETECT_TT_PICK;
        return 0;
    }
    if (what == TBA_GET_TB_PARALLEL) {
        if ((size_t)out_bytes < N * 4) return set_err(TBA_E_ARG, "output buffer too small");
        for (size_t i = 0; i < N; i++) ((i32 *)out)[i] = rs[i].tb_done;
        return 0;
    }
    if (what == TBA_GET_ED_FORM || what == TBA_GET_TB_FORM || what == TBA_GET_TB_VERIFY_FAIL) {
        if ((size_t)out_bytes < N * 4) return set_err(TBA_E_ARG, "output buffer too small");
        for (size_t i = 0; i < N; i++)
            ((i32 *)out)[i] = what == TBA_GET_ED_FORM ? rs[i].ed_form : what == TBA_GET_TB_FORM ? rs[i].tb_form : rs[i].tb_verify_fail;
        return 0;
    }
    if (what == TBA_GET_DP_WORKGROUP) {
        if ((size_t)out_bytes < N * 4) return set_err(TBA_E_ARG, "output buffer too small");
        for (size_t i = 0; i < N; i++) ((i32 *)out)[i] = rs[i].dp_wg;
        return 0;
    }
    if (what == TBA_GET_ED_N_TAKEN) {
        if ((size_t)out_bytes < N * 8) return set_err(TBA_E_ARG, "output buffer too small");
        for (size_t i = 0; i < N; i++) ((i64 *)out)[i] = rs[i].n_taken;
        return 0;
    }
    if (what == TBA_GET_N_CPTS || what == TBA_GET_DP_READ_START || what == TBA_GET_N_STALL ||
        what == TBA_GET_STALL_OFF) {
        if ((size_t)out_bytes < N * 8) return set_err(TBA_E_ARG, "output buffer too small");
        for (size_t i = 0; i < N; i++)
            ((i64 *)out)[i] = what == TBA_GET_N_CPTS ? rs[i].n_cpts : what == TBA_GET_N_STALL ? rs[i].n_stall
                              : what == TBA_GET_STALL_OFF ? rs[i].stall_off : rs[i].dp_read_start;
        return 0;
    }
    if (what == TBA_GET_SEG_SV || what == TBA_GET_START || what == TBA_GET_THEIL_SEN) {
        if ((size_t)out_bytes < N * 32) return set_err(TBA_E_ARG, "output buffer too small");
        double *o = (double *)out;
        for (size_t i = 0; i < N; i++)
            for (int k = 0; k < 4; k++)
                o[4 * i + k] = what == TBA_GET_START ? rs[i].start_res[k]
                               : what == TBA_GET_THEIL_SEN ? rs[i].ts[k]
                               : (k == 0 ? rs[i].shift : k == 1 ? rs[i].scale : k == 2 ? rs[i].lower : rs[i].upper);
        return 0;
    }
    if (what == TBA_GET_DEBUG_COUNTERS) { // phase cycle / sweep counters of a profiling build
        if ((size_t)out_bytes < N * 64) return set_err(TBA_E_ARG, "output buffer too small");
        for (size_t i = 0; i < N; i++) memcpy((char *)out + 64 * i, rs[i].dbg, 64);
        return 0;
    }
    if (what == TBA_GET_PATH) {
        if ((size_t)out_bytes < N * 16) return set_err(TBA_E_ARG, "output buffer too small");
        i32 *o = (i32 *)out;
        for (size_t i = 0; i < N; i++) {
            o[4 * i + 0] = rs[i].path; o[4 * i + 1] = (i32)rs[i].n_static;
            o[4 * i + 2] = (i32)rs[i].W; o[4 * i + 3] = rs[i].n_start_calls;
        }
        return 0;
    }
    return set_err(TBA_E_ARG, "unknown TBA_GET_* selector");
}

extern "C" int tba_batch_base_stats(tba_engine *e, double *means, double *stds, int64_t n_values)
{
    if (!e || !e->have_batch || !e->finished) return set_err(TBA_E_STATE, "no finished batch");
    if (!means || !stds || n_values < e->B_tot) return set_err(TBA_E_ARG, "output buffers too small");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->B_tot == 0) return 0;
    if (e->d_stat.ensure((size_t)e->B_tot * 16)) return set_err(TBA_E_NOMEM, "hipMalloc failed");
    double *d_m = e->d_stat.as<double>(), *d_s = d_m + e->B_tot;
    const unsigned gB = (unsigned)std::min<i64>(std::max<i64>((e->max_B + 255) / 256, 1), 128);
    k_base_stats<<<dim3(gB, (unsigned)e->n_reads), 256, 0, e->stream>>>(e->d_rs.as<ReadState>(),
        e->d_dp.as<DevParams>(), e->hp.o.skip_norm_out ? nullptr : e->d_norm_out.as<double>(),
        e->d_norm.as<double>(), e->d_segs.as<i64>(), d_m, d_s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(means, d_m, (size_t)e->B_tot * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(stds, d_s, (size_t)e->B_tot * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int tba_batch_stats(tba_engine *e, double *algorithmic_bytes, double *dp_cells)
{
    if (!e || !e->have_batch) return set_err(TBA_E_STATE, "no batch uploaded");
    if (algorithmic_bytes) *algorithmic_bytes = e->algo_bytes;
    if (dp_cells) *dp_cells = e->dp_cells;
    return 0;
}

extern "C" const char *tba_stage_name(int i) { return i >= 0 && i < N_STAGE ? STAGE_NAMES[i] : ""; }

// ---------------------------------------------------------------------------------------------
// per-kernel entry points (tba_c_*): host buffers in, host buffers out, batch of one
namespace {
struct Tmp { // scoped device allocation
    void *p = nullptr;
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess ? 0 : 1; }
    ~Tmp() { if (p) (void)hipFree(p); }
    template <class T> T *as() { return (T *)p; }
};
#define C_TRY(expr) HIP_TRY(expr)
static unsigned grid_for(i64 n) { return (unsigned)std::min<i64>(std::max<i64>((n + 255) / 256, 1), 4096); }

template <int CPL>
static void launch_direct_t(tba_engine *e, DpJob *job)
{
    k_dp<CPL, true><<<dim3(1), dim3(64), 0, e->stream>>>(
        e->d_rs.as<ReadState>(), e->d_dp.as<DevParams>(), DP_DIRECT, nullptr, nullptr, nullptr,
        nullptr, nullptr, nullptr, nullptr, 0, nullptr, job);
}
static void launch_direct(tba_engine *e, int cpl, DpJob *job)
{
    switch (cpl) {
    case 4: launch_direct_t<4>(e, job); break;
    case 5: launch_direct_t<5>(e, job); break;
    case 8: launch_direct_t<8>(e, job); break;
    case 12: launch_direct_t<12>(e, job); break;
    case 16: launch_direct_t<16>(e, job); break;
    case 24: launch_direct_t<24>(e, job); break;
    case 32: launch_direct_t<32>(e, job); break;
    case 48: launch_direct_t<48>(e, job); break;
    default: break;
    }
}

// shared by the two forward-pass entry points
static int run_direct_dp(tba_engine *e, DpJob hj, int cpl, i64 n_rows, i64 W, i64 row0,
                         double *fwd_host, int64_t *tb_host, int64_t *starts_host, i64 starts_from)
{
    const i64 stride = (i64)cpl * 64;            // forward rows
    const i64 mstride = mv_class_rowb(cpl);      // packed 2-bit move rows
    Tmp d_fwd, d_mv, d_job;
    if (d_fwd.alloc((size_t)(n_rows + 1) * stride * 8) || d_mv.alloc((size_t)(n_rows + 1) * mstride) ||
        d_job.alloc(sizeof(DpJob)))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    // the kernel needs *some* ReadState / DevParams to bind its references to
    if (e->d_rs.ensure(sizeof(ReadState)) || e->d_dp.ensure(sizeof(DevParams))) return TBA_E_NOMEM;
    hj.fwd_out = d_fwd.as<double>();
    hj.mv = d_mv.as<unsigned char>();
    hj.status = TBA_OK;
    C_TRY(hipMemcpyAsync(d_job.p, &hj, sizeof(DpJob), hipMemcpyHostToDevice, e->stream));
    launch_direct(e, cpl, d_job.as<DpJob>());
    C_TRY(hipGetLastError());
    C_TRY(hipMemcpyAsync(&hj, d_job.p, sizeof(DpJob), hipMemcpyDeviceToHost, e->stream));
    std::vector<unsigned char> mv((size_t)(n_rows + 1) * mstride);
    C_TRY(hipMemcpyAsync(mv.data(), d_mv.p, mv.size(), hipMemcpyDeviceToHost, e->stream));
    C_TRY(hipStreamSynchronize(e->stream));
    if (hj.status != TBA_OK) return hj.status;
    // rows row0+1 .. n_rows are new (row 0 too when starting from scratch); device rows are
    // padded to the moves stride
    {
        std::vector<double> fw((size_t)(n_rows + 1) * stride);
        C_TRY(hipMemcpy(fw.data(), d_fwd.p, fw.size() * 8, hipMemcpyDeviceToHost));
        for (i64 r = row0 == 0 ? 0 : row0 + 1; r <= n_rows; r++)
            memcpy(fwd_host + r * W, fw.data() + r * stride, (size_t)W * 8);
    }
    for (i64 r = row0 + 1; r <= n_rows; r++)
        for (i64 b = 0; b < W; b++)
            tb_host[r * W + b] = (mv[(size_t)(r * mstride + (b >> 2))] >> (2 * (b & 3))) & 3;
    if (starts_host)
        C_TRY(hipMemcpy(starts_host + starts_from, hj.starts + starts_from,
                        (size_t)(n_rows - starts_from) * 8, hipMemcpyDeviceToHost));
    return TBA_OK;
}
} // namespace

extern "C" int tba_c_adaptive_banded_forward_pass_z(tba_engine *e, double *fwd_pass,
    int64_t *fwd_pass_tb, int64_t n_bases, int64_t bandwidth, int64_t *event_starts,
    const double *event_means, int64_t n_events, const double *r_ref_means,
    const double *r_ref_sds, double z_shift, double skip_pen, double stay_pen,
    int64_t start_seq_pos, double mask_fill_z_score, int do_winsorize_z, double max_half_z_score,
    double *z_scores)
{
    if (!e || !fwd_pass || !fwd_pass_tb || !event_starts || !event_means || !r_ref_means ||
        !r_ref_sds || n_bases < 1 || bandwidth < 2 || start_seq_pos < 1 || start_seq_pos > n_bases)
        return set_err(TBA_E_ARG, "bad arguments");
    const int cpl = cpl_class(bandwidth);
    if (!cpl) return TBA_UNSUPPORTED;
    if (n_events < 1) return set_err(TBA_E_ARG, "no events");
    for (i64 i = 0; i < start_seq_pos; i++) // the given (static) rows' band starts index the events
        if (event_starts[i] < 0 || event_starts[i] >= n_events || (i > 0 && event_starts[i] < event_starts[i - 1]))
            return set_err(TBA_E_ARG, "event_starts must be non-decreasing inside [0, n_events)");
    HIP_TRY(hipSetDevice(e->device));
    Tmp d_ev, d_mu, d_sd, d_st, d_init, d_z;
    const size_t z_bytes = (size_t)(n_bases - start_seq_pos) * (size_t)bandwidth * 8;
    if (z_scores && z_bytes && d_z.alloc(z_bytes)) return set_err(TBA_E_NOMEM, "hipMalloc failed");
    if (d_ev.alloc((size_t)n_events * 8) || d_mu.alloc((size_t)n_bases * 8) ||
        d_sd.alloc((size_t)n_bases * 8) || d_st.alloc((size_t)n_bases * 8) ||
        d_init.alloc((size_t)bandwidth * 8))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_ev.p, event_means, (size_t)n_events * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_mu.p, r_ref_means, (size_t)n_bases * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_sd.p, r_ref_sds, (size_t)n_bases * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_st.p, event_starts, (size_t)n_bases * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_init.p, fwd_pass + start_seq_pos * bandwidth, (size_t)bandwidth * 8, hipMemcpyHostToDevice));
    DpJob j;
    memset(&j, 0, sizeof(j));
    j.W = bandwidth; j.n_rows = n_bases; j.row0 = start_seq_pos; j.n_static = start_seq_pos;
    j.n_ev = n_events; j.ev = d_ev.as<double>(); j.mu = d_mu.as<double>(); j.sd = d_sd.as<double>();
    j.zmat = nullptr; j.starts = d_st.as<i64>(); j.init_row = d_init.as<double>();
    j.z_out = z_scores && z_bytes ? d_z.as<double>() : nullptr;
    j.z_shift = z_shift; j.skip_pen = skip_pen; j.stay_pen = stay_pen; j.max_half_z = max_half_z_score;
    j.fill = mask_fill_z_score; j.winsor = do_winsorize_z ? 1 : 0;
    const int rc = run_direct_dp(e, j, cpl, n_bases, bandwidth, start_seq_pos, fwd_pass, fwd_pass_tb,
                                 event_starts, start_seq_pos);
    if (rc == TBA_OK && j.z_out) C_TRY(hipMemcpy(z_scores, d_z.p, z_bytes, hipMemcpyDeviceToHost));
    return rc;
}

extern "C" int tba_c_adaptive_banded_forward_pass(tba_engine *e, double *fwd_pass,
    int64_t *fwd_pass_tb, int64_t n_bases, int64_t bandwidth, int64_t *event_starts,
    const double *event_means, int64_t n_events, const double *r_ref_means,
    const double *r_ref_sds, double z_shift, double skip_pen, double stay_pen,
    int64_t start_seq_pos, double mask_fill_z_score, int do_winsorize_z, double max_half_z_score)
{
    return tba_c_adaptive_banded_forward_pass_z(e, fwd_pass, fwd_pass_tb, n_bases, bandwidth, event_starts,
        event_means, n_events, r_ref_means, r_ref_sds, z_shift, skip_pen, stay_pen, start_seq_pos,
        mask_fill_z_score, do_winsorize_z, max_half_z_score, nullptr);
}


extern "C" int tba_c_banded_forward_pass(tba_engine *e, const double *shifted_z_scores,
    int64_t n_bases, int64_t bandwidth, const int64_t *event_starts, double skip_pen,
    double stay_pen, double *fwd_pass, int64_t *fwd_pass_tb)
{
    if (!e || !shifted_z_scores || !event_starts || !fwd_pass || !fwd_pass_tb || n_bases < 1 ||
        bandwidth < 2)
        return set_err(TBA_E_ARG, "bad arguments");
    const int cpl = cpl_class(bandwidth);
    if (!cpl) return TBA_UNSUPPORTED;
    for (i64 i = 0; i < n_bases; i++)
        if (event_starts[i] < 0 || (i > 0 && event_starts[i] < event_starts[i - 1]))
            return set_err(TBA_E_ARG, "event_starts must be non-negative and non-decreasing");
    HIP_TRY(hipSetDevice(e->device));
    Tmp d_z, d_st;
    if (d_z.alloc((size_t)n_bases * bandwidth * 8) || d_st.alloc((size_t)n_bases * 8))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_z.p, shifted_z_scores, (size_t)n_bases * bandwidth * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_st.p, event_starts, (size_t)n_bases * 8, hipMemcpyHostToDevice));
    DpJob j;
    memset(&j, 0, sizeof(j));
    j.W = bandwidth; j.n_rows = n_bases; j.row0 = 0; j.n_static = n_bases; j.n_ev = 0;
    j.zmat = d_z.as<double>(); j.starts = d_st.as<i64>(); j.init_row = nullptr;
    j.skip_pen = skip_pen; j.stay_pen = stay_pen;
    int rc = run_direct_dp(e, j, cpl, n_bases, bandwidth, 0, fwd_pass, fwd_pass_tb, nullptr, 0);
    if (rc == TBA_OK) // row 0 of the move matrix is never read; the reference leaves it empty
        for (i64 b = 0; b < bandwidth; b++) fwd_pass_tb[b] = 0;
    return rc;
}

extern "C" int tba_c_banded_traceback(tba_engine *e, const int64_t *fwd_pass_tb, int64_t n_bases,
    int64_t bandwidth, const int64_t *event_starts, int64_t band_pos,
    int64_t band_boundary_thresh, int64_t *seq_poss)
{
    if (!e || !fwd_pass_tb || !event_starts || !seq_poss || n_bases < 1 || bandwidth < 1 ||
        band_pos < 0 || band_pos >= bandwidth)
        return set_err(TBA_E_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(e->device));
    const size_t cells = (size_t)(n_bases + 1) * bandwidth;
    std::vector<unsigned char> mv(cells);
    for (size_t i = 0; i < cells; i++) mv[i] = (unsigned char)fwd_pass_tb[i];
    Tmp d_mv, d_st, d_out, d_status;
    if (d_mv.alloc(cells) || d_st.alloc((size_t)n_bases * 8) || d_out.alloc((size_t)(n_bases + 1) * 8) ||
        d_status.alloc(4))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_mv.p, mv.data(), cells, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_st.p, event_starts, (size_t)n_bases * 8, hipMemcpyHostToDevice));
    k_c_traceback<<<1, 64, 0, e->stream>>>(d_mv.as<unsigned char>(), bandwidth, n_bases, bandwidth,
                                           d_st.as<i64>(), band_pos, band_boundary_thresh,
                                           d_out.as<i64>(), d_status.as<i32>());
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    i32 st = 0;
    C_TRY(hipMemcpy(&st, d_status.p, 4, hipMemcpyDeviceToHost));
    if (st == TBA_OK) C_TRY(hipMemcpy(seq_poss, d_out.p, (size_t)(n_bases + 1) * 8, hipMemcpyDeviceToHost));
    return st;
}

extern "C" int tba_c_base_z_scores(tba_engine *e, const double *b_sig, int64_t n, double ref_mean,
    double ref_sd, int do_winsorize_z, double max_half_z_score, double *out)
{
    if (!e || !b_sig || !out || n < 0) return set_err(TBA_E_ARG, "bad arguments");
    if (n == 0) return TBA_OK;
    HIP_TRY(hipSetDevice(e->device));
    Tmp d_in, d_out;
    if (d_in.alloc((size_t)n * 8) || d_out.alloc((size_t)n * 8)) return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_in.p, b_sig, (size_t)n * 8, hipMemcpyHostToDevice));
    k_c_base_z_scores<<<grid_for(n), 256, 0, e->stream>>>(d_in.as<double>(), n, ref_mean, ref_sd,
                                                          do_winsorize_z, max_half_z_score, d_out.as<double>());
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    C_TRY(hipMemcpy(out, d_out.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return TBA_OK;
}

extern "C" int tba_c_new_means(tba_engine *e, const double *norm_signal, int64_t n_sig,
    const int64_t *new_segs, int64_t n_segs, double *means)
{
    if (!e || !norm_signal || !new_segs || !means || n_segs < 0 || n_sig < 0)
        return set_err(TBA_E_ARG, "bad arguments");
    if (n_segs == 0) return TBA_OK;
    for (i64 i = 0; i <= n_segs; i++)
        if (new_segs[i] < 0 || new_segs[i] > n_sig || (i > 0 && new_segs[i] < new_segs[i - 1]))
            return set_err(TBA_E_ARG, "segment boundaries outside the signal");
    HIP_TRY(hipSetDevice(e->device));
    // the batch pipeline's own kernel on a one-read batch (k_event_means: wave-cooperative, software-
    // pipelined segment sums, k_select.h) -- the slice size picked as the batch pipeline picks it,
    // by the mean segment length (+ 64 bytes: a 16-byte access may touch the element past an odd end)
    Tmp d_sig, d_segs, d_out, d_rs, d_dp;
    if (d_sig.alloc((size_t)n_sig * 8 + 64) || d_segs.alloc((size_t)(n_segs + 1) * 8) || d_out.alloc((size_t)n_segs * 8) ||
        d_rs.alloc(sizeof(ReadState)) || d_dp.alloc(sizeof(DevParams)))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    ReadState r;
    memset(&r, 0, sizeof(r));
    r.n_raw = n_sig; r.n_cpts = n_segs + 1; r.status = TBA_OK;
    DevParams dp;
    memset(&dp, 0, sizeof(dp));
    C_TRY(hipMemset(d_sig.p, 0, (size_t)n_sig * 8 + 64));
    C_TRY(hipMemcpy(d_sig.p, norm_signal, (size_t)n_sig * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_segs.p, new_segs, (size_t)(n_segs + 1) * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_rs.p, &r, sizeof(r), hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_dp.p, &dp, sizeof(dp), hipMemcpyHostToDevice));
    const unsigned g = (unsigned)std::min<i64>(std::max<i64>((n_segs + 255) / 256, 1), 128);
    if (n_sig >= 10 * n_segs)
        k_event_means<double, 1280><<<dim3(g, 1), 256, 0, e->stream>>>(d_rs.as<ReadState>(), d_dp.as<DevParams>(), d_sig.as<double>(), d_segs.as<i64>(), d_out.as<double>(), 0);
    else
        k_event_means<double><<<dim3(g, 1), 256, 0, e->stream>>>(d_rs.as<ReadState>(), d_dp.as<DevParams>(), d_sig.as<double>(), d_segs.as<i64>(), d_out.as<double>(), 0);
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    C_TRY(hipMemcpy(means, d_out.p, (size_t)n_segs * 8, hipMemcpyDeviceToHost));
    return TBA_OK;
}

extern "C" int tba_c_apply_outlier_thresh(tba_engine *e, const double *sig, int64_t n,
    double lower_lim, double upper_lim, double *out)
{
    if (!e || !sig || !out || n < 0) return set_err(TBA_E_ARG, "bad arguments");
    if (n == 0) return TBA_OK;
    HIP_TRY(hipSetDevice(e->device));
    Tmp d_in, d_out;
    if (d_in.alloc((size_t)n * 8) || d_out.alloc((size_t)n * 8)) return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_in.p, sig, (size_t)n * 8, hipMemcpyHostToDevice));
    k_c_clip<<<grid_for(n), 256, 0, e->stream>>>(d_in.as<double>(), n, lower_lim, upper_lim, d_out.as<double>());
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    C_TRY(hipMemcpy(out, d_out.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return TBA_OK;
}

// the two change-point detectors run the batch kernels on a one-read batch
static int c_valid_cpts(tba_engine *e, const double *sig, int64_t n, int64_t min_base_obs,
                        int64_t width, int64_t num_cpts, int64_t *cpts, int ttest)
{
    if (!e || !sig || !cpts || n < 1 || min_base_obs < 1 || width < 1 || num_cpts < 1)
        return set_err(TBA_E_ARG, "bad arguments");
    if ((ttest ? n - 2 * width : n + 1 - 2 * width) <= 0) return TBA_INTERNAL;
    HIP_TRY(hipSetDevice(e->device));
    Tmp d_sig, d_csum, d_score, d_state, d_cpts, d_rs, d_dp;
    if (d_sig.alloc((size_t)n * 8) || d_csum.alloc((size_t)(n + 1) * 8) || d_score.alloc((size_t)n * 8) ||
        d_state.alloc((size_t)n) || d_cpts.alloc((size_t)num_cpts * 8) || d_rs.alloc(sizeof(ReadState)) ||
        d_dp.alloc(sizeof(DevParams)))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    ReadState r;
    memset(&r, 0, sizeof(r));
    r.n_raw = n; r.num_events = num_cpts; r.status = TBA_OK;
    DevParams dp;
    memset(&dp, 0, sizeof(dp));
    dp.p.running_stat_width = width; dp.p.min_obs_per_base = min_base_obs;
    C_TRY(hipMemcpy(d_sig.p, sig, (size_t)n * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_rs.p, &r, sizeof(r), hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_dp.p, &dp, sizeof(dp), hipMemcpyHostToDevice));
    hipStream_t s = e->stream;
    const unsigned g = grid_for(n) > 128 ? 128 : grid_for(n);
    // The score-free kernels of the batch pipeline (k_detect.h) when the engine's dispatch sends a
    // batch of one read through the throughput form (tba_engine_set_dispatch(0, ...): the parity tests
    // of this entry in both forms) and for the t-test scores at RNA's defaults; the kernels that keep
    // the scores then run on what those left (flagged reads) -- exactly the batch pipeline's sequence.
    int only_flagged = 0;
    if (!ttest && e->small_batch < 1 && 2 * width <= DT_W2MAX && min_base_obs == 3) {
        // (shift 0, scale 1, no limits: the loader's normalised copy of the signal is the signal)
        Tmp d_norm;
        if (d_norm.alloc((size_t)(n + 2) * 8 + 64)) return set_err(TBA_E_NOMEM, "hipMalloc failed");
        r.shift = 0.0; r.scale = 1.0;
        r.is_long = n > TBA_LONG_RAW;
        C_TRY(hipMemcpy(d_rs.p, &r, sizeof(r), hipMemcpyHostToDevice));
        k_detect<2, double><<<1, 256, 0, s>>>(d_rs.as<ReadState>(), 1, d_dp.as<DevParams>(), d_sig.as<double>(), d_norm.as<double>(), d_csum.as<double>(), d_score.as<double>(), n);
        k_pick<<<1, SEL_NT, 0, s>>>(d_rs.as<ReadState>(), d_dp.as<DevParams>(), d_csum.as<double>(), d_score.as<double>(), d_cpts.as<i64>(), 0);
        C_TRY(hipStreamSynchronize(s)); // (d_norm is released at the end of this scope)
        only_flagged = 1;
    } else if (ttest && e->small_batch < 1 && min_base_obs == 6 && width <= TT_MAXW) {
        if (width == 12) k_detect_tt<5, 12, double><<<1, SEL_NT, 0, s>>>(d_rs.as<ReadState>(), d_dp.as<DevParams>(), d_sig.as<double>(), d_csum.as<double>(), d_score.as<double>());
        else k_detect_tt<5, 0, double><<<1, SEL_NT, 0, s>>>(d_rs.as<ReadState>(), d_dp.as<DevParams>(), d_sig.as<double>(), d_csum.as<double>(), d_score.as<double>());
        k_pick<<<1, SEL_NT, 0, s>>>(d_rs.as<ReadState>(), d_dp.as<DevParams>(), d_csum.as<double>(), d_score.as<double>(), d_cpts.as<i64>(), 1);
        only_flagged = 1;
    }
    if (!ttest && 2 * width <= 64) { // the batch pipeline's fused form
        k_cumsum_scores<32><<<1, 256, 0, s>>>(d_rs.as<ReadState>(), 1, d_dp.as<DevParams>(), d_sig.as<double>(), d_score.as<double>(), only_flagged);
    } else if (!ttest) {
        k_cumsum<<<1, 64, 0, s>>>(d_rs.as<ReadState>(), 1, d_sig.as<double>(), d_csum.as<double>());
        k_scores_dna<<<dim3(g, 1), 256, 0, s>>>(d_rs.as<ReadState>(), d_dp.as<DevParams>(), d_csum.as<double>(), d_score.as<double>());
    } else {
        k_scores_ttest<double><<<dim3(g, 1), 256, 0, s>>>(d_rs.as<ReadState>(), d_dp.as<DevParams>(), d_sig.as<double>(), d_score.as<double>(), only_flagged);
    }
    launch_peaks(min_base_obs, 1, s, d_rs.as<ReadState>(), d_dp.as<DevParams>(), d_score.as<double>(),
                                 d_state.as<unsigned char>(), d_csum.as<double>(), d_cpts.as<i64>(), ttest, only_flagged,
                                 ttest ? TBA_ED_FORM_TTEST_PEAKS : TBA_ED_FORM_SCORES_PEAKS);
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(s));
    C_TRY(hipMemcpy(&r, d_rs.p, sizeof(r), hipMemcpyDeviceToHost));
    if (r.status == TBA_OK) C_TRY(hipMemcpy(cpts, d_cpts.p, (size_t)num_cpts * 8, hipMemcpyDeviceToHost));
    e->last_c_ed_form = r.ed_form;
    return r.status;
}

// which kernels produced the result of the last tba_c_valid_cpts_w_cap[_t_test] call (TBA_ED_FORM_*)
extern "C" int tba_c_last_ed_form(tba_engine *e) { return e ? e->last_c_ed_form : 0; }

extern "C" int tba_c_valid_cpts_w_cap(tba_engine *e, const double *sig, int64_t n,
    int64_t min_base_obs, int64_t running_stat_width, int64_t num_cpts, int64_t *cpts)
{
    return c_valid_cpts(e, sig, n, min_base_obs, running_stat_width, num_cpts, cpts, 0);
}

extern "C" int tba_c_valid_cpts_w_cap_t_test(tba_engine *e, const double *sig, int64_t n,
    int64_t min_base_obs, int64_t running_stat_width, int64_t num_cpts, int64_t *cpts)
{
    return c_valid_cpts(e, sig, n, min_base_obs, running_stat_width, num_cpts, cpts, 1);
}

extern "C" int tba_c_new_mean_stds(tba_engine *e, const double *norm_signal, int64_t n_sig,
    const int64_t *new_segs, int64_t n_segs, double *means, double *stds)
{
    if (!e || !norm_signal || !new_segs || !means || !stds || n_segs < 0 || n_sig < 0)
        return set_err(TBA_E_ARG, "bad arguments");
    if (n_segs == 0) return TBA_OK;
    for (i64 i = 0; i <= n_segs; i++)
        if (new_segs[i] < 0 || new_segs[i] > n_sig || (i > 0 && new_segs[i] < new_segs[i - 1]))
            return set_err(TBA_E_ARG, "segment boundaries outside the signal");
    HIP_TRY(hipSetDevice(e->device));
    Tmp d_sig, d_segs, d_m, d_s;
    if (d_sig.alloc((size_t)n_sig * 8) || d_segs.alloc((size_t)(n_segs + 1) * 8) ||
        d_m.alloc((size_t)n_segs * 8) || d_s.alloc((size_t)n_segs * 8))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_sig.p, norm_signal, (size_t)n_sig * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_segs.p, new_segs, (size_t)(n_segs + 1) * 8, hipMemcpyHostToDevice));
    k_c_new_mean_stds<<<grid_for(n_segs), 256, 0, e->stream>>>(d_sig.as<double>(),
        d_segs.as<i64>(), n_segs, d_m.as<double>(), d_s.as<double>());
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    C_TRY(hipMemcpy(means, d_m.p, (size_t)n_segs * 8, hipMemcpyDeviceToHost));
    C_TRY(hipMemcpy(stds, d_s.p, (size_t)n_segs * 8, hipMemcpyDeviceToHost));
    return TBA_OK;
}

extern "C" int tba_c_compute_slopes(tba_engine *e, const double *r_event_means,
    const double *r_model_means, int64_t n, double max_slope, double *slopes)
{
    if (!e || !r_event_means || !r_model_means || !slopes || n < 0)
        return set_err(TBA_E_ARG, "bad arguments");
    if (n < 2) return TBA_OK;
    if (n > 65535) return set_err(TBA_E_ARG, "too many points");
    const size_t ns = (size_t)n * (size_t)(n - 1) / 2;
    HIP_TRY(hipSetDevice(e->device));
    Tmp d_ev, d_md, d_out;
    if (d_ev.alloc((size_t)n * 8) || d_md.alloc((size_t)n * 8) || d_out.alloc(ns * 8))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_ev.p, r_event_means, (size_t)n * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_md.p, r_model_means, (size_t)n * 8, hipMemcpyHostToDevice));
    k_c_compute_slopes<<<dim3((unsigned)(n - 1)), 256, 0, e->stream>>>(d_ev.as<double>(),
        d_md.as<double>(), n, max_slope, d_out.as<double>());
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    C_TRY(hipMemcpy(slopes, d_out.p, ns * 8, hipMemcpyDeviceToHost));
    return TBA_OK;
}

extern "C" int tba_c_reg_z_scores(tba_engine *e, const double *r_sig, int64_t n_sig,
    const double *r_ref_means, const double *r_ref_sds, int64_t n_bases,
    const int64_t *r_b_starts, int64_t n_b_starts, int64_t reg_start, int64_t reg_end,
    int64_t max_base_shift, int64_t min_obs_per_base, int do_winsorize_z,
    double max_half_z_score, int64_t *bounds, int64_t *z_off, double *z, int64_t z_cap)
{
    if (!e || !r_sig || !r_ref_means || !r_ref_sds || !r_b_starts || !bounds || !z_off || !z ||
        n_sig < 0 || z_cap < 0)
        return set_err(TBA_E_ARG, "bad arguments");
    const i64 reg_len = reg_end - reg_start;
    if (reg_start < 0 || reg_len <= 0 || reg_end > n_bases || reg_end >= n_b_starts ||
        max_base_shift < 0)
        return set_err(TBA_E_ARG, "region outside the bases / base starts");
    HIP_TRY(hipSetDevice(e->device));
    Tmp d_sig, d_mu, d_sd, d_bs, d_ss, d_se, d_off, d_z, d_st;
    if (d_sig.alloc((size_t)n_sig * 8) || d_mu.alloc((size_t)n_bases * 8) ||
        d_sd.alloc((size_t)n_bases * 8) || d_bs.alloc((size_t)n_b_starts * 8) ||
        d_ss.alloc((size_t)reg_len * 8) || d_se.alloc((size_t)reg_len * 8) ||
        d_off.alloc((size_t)(reg_len + 1) * 8) || d_z.alloc((size_t)z_cap * 8) || d_st.alloc(4))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_sig.p, r_sig, (size_t)n_sig * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_mu.p, r_ref_means, (size_t)n_bases * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_sd.p, r_ref_sds, (size_t)n_bases * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_bs.p, r_b_starts, (size_t)n_b_starts * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemsetAsync(d_st.p, 0, 4, e->stream));
    k_c_reg_bounds<<<1, 64, 0, e->stream>>>(d_bs.as<i64>(), reg_start, reg_end, max_base_shift,
        min_obs_per_base, d_ss.as<i64>(), d_se.as<i64>(), d_off.as<i64>());
    k_c_reg_z<<<dim3((unsigned)reg_len), 64, 0, e->stream>>>(d_sig.as<double>(), n_sig,
        d_mu.as<double>(), d_sd.as<double>(), reg_start, d_ss.as<i64>(), d_se.as<i64>(),
        d_off.as<i64>(), z_cap, do_winsorize_z, max_half_z_score, d_z.as<double>(),
        d_st.as<i32>());
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    std::vector<i64> ss(reg_len), se(reg_len);
    i32 st = 0;
    C_TRY(hipMemcpy(&st, d_st.p, 4, hipMemcpyDeviceToHost));
    C_TRY(hipMemcpy(ss.data(), d_ss.p, (size_t)reg_len * 8, hipMemcpyDeviceToHost));
    C_TRY(hipMemcpy(se.data(), d_se.p, (size_t)reg_len * 8, hipMemcpyDeviceToHost));
    C_TRY(hipMemcpy(z_off, d_off.p, (size_t)(reg_len + 1) * 8, hipMemcpyDeviceToHost));
    if (z_off[reg_len] > z_cap) return set_err(TBA_E_ARG, "z-score buffer too small");
    if (st != 0) return set_err(TBA_E_ARG, "base intervals outside the signal");
    const i64 base = r_b_starts[reg_start];
    for (i64 i = 0; i < reg_len; i++) { bounds[2 * i] = ss[i] - base; bounds[2 * i + 1] = se[i] - base; }
    if (z_off[reg_len] > 0)
        C_TRY(hipMemcpy(z, d_z.p, (size_t)z_off[reg_len] * 8, hipMemcpyDeviceToHost));
    return TBA_OK;
}

extern "C" int tba_c_base_forward_pass(tba_engine *e, const double *b_data, int64_t b_start,
    int64_t b_end, const double *prev_b_data, int64_t prev_b_start, int64_t prev_b_end,
    const double *prev_b_fwd_data, const int64_t *prev_b_last_diag, int64_t min_obs_per_base,
    double *b_fwd_data, int64_t *b_last_diag)
{
    if (!e || !b_data || !prev_b_data || !prev_b_fwd_data || !prev_b_last_diag || !b_fwd_data ||
        !b_last_diag)
        return set_err(TBA_E_ARG, "bad arguments");
    const i64 b_len = b_end - b_start, plen = prev_b_end - prev_b_start;
    if (b_len <= 0 || plen <= 0) return set_err(TBA_E_ARG, "empty base interval");
    HIP_TRY(hipSetDevice(e->device));
    Tmp d_b, d_pb, d_pf, d_pl, d_cum, d_f, d_l, d_st;
    if (d_b.alloc((size_t)b_len * 8) || d_pb.alloc((size_t)plen * 8) ||
        d_pf.alloc((size_t)plen * 8) || d_pl.alloc((size_t)plen * 8) ||
        d_cum.alloc((size_t)plen * 8) || d_f.alloc((size_t)b_len * 8) ||
        d_l.alloc((size_t)b_len * 8) || d_st.alloc(4))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_b.p, b_data, (size_t)b_len * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_pb.p, prev_b_data, (size_t)plen * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_pf.p, prev_b_fwd_data, (size_t)plen * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_pl.p, prev_b_last_diag, (size_t)plen * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemsetAsync(d_st.p, 0, 4, e->stream));
    k_c_base_forward_pass<<<1, 64, 0, e->stream>>>(d_b.as<double>(), b_start, b_end,
        d_pb.as<double>(), prev_b_start, prev_b_end, d_pf.as<double>(), d_pl.as<i64>(),
        min_obs_per_base, d_cum.as<double>(), d_f.as<double>(), d_l.as<i64>(), d_st.as<i32>());
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    i32 st = 0;
    C_TRY(hipMemcpy(&st, d_st.p, 4, hipMemcpyDeviceToHost));
    if (st != 0) return st;
    C_TRY(hipMemcpy(b_fwd_data, d_f.p, (size_t)b_len * 8, hipMemcpyDeviceToHost));
    C_TRY(hipMemcpy(b_last_diag, d_l.p, (size_t)b_len * 8, hipMemcpyDeviceToHost));
    return TBA_OK;
}

extern "C" int tba_c_base_traceback(tba_engine *e, const double *curr_b_data, int64_t curr_len,
    int64_t curr_start, const double *next_b_data, int64_t next_len, int64_t next_start,
    int64_t next_end, int64_t sig_start, int64_t min_obs_per_base, int64_t *sig_pos)
{
    if (!e || !curr_b_data || !next_b_data || !sig_pos || curr_len <= 0 || next_len <= 0)
        return set_err(TBA_E_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(e->device));
    Tmp d_c, d_n, d_out, d_st;
    if (d_c.alloc((size_t)curr_len * 8) || d_n.alloc((size_t)next_len * 8) || d_out.alloc(8) ||
        d_st.alloc(4))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_c.p, curr_b_data, (size_t)curr_len * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_n.p, next_b_data, (size_t)next_len * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemsetAsync(d_st.p, 0, 4, e->stream));
    k_c_base_traceback<<<1, 64, 0, e->stream>>>(d_c.as<double>(), curr_len, curr_start,
        d_n.as<double>(), next_len, next_start, next_end, sig_start, min_obs_per_base,
        d_out.as<i64>(), d_st.as<i32>());
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    i32 st = 0;
    C_TRY(hipMemcpy(&st, d_st.p, 4, hipMemcpyDeviceToHost));
    if (st != 0) return st;
    C_TRY(hipMemcpy(sig_pos, d_out.p, 8, hipMemcpyDeviceToHost));
    return TBA_OK;
}

extern "C" int tba_llh_ratio_windows(tba_engine *e, int kind, const double *means,
    const double *ref_means, const double *alt_means, const double *ref_vars,
    const double *alt_vars, int64_t n_values, int64_t width, const int64_t *starts,
    int64_t n_windows, const double *par, double *out)
{
    if (!e || !means || !ref_means || !alt_means || !ref_vars || !starts || !out || kind < 0 ||
        kind > 2 || (kind == 0 && !alt_vars) || (kind == 2 && !par) || n_values < 0 || width < 0 ||
        n_windows < 0)
        return set_err(TBA_E_ARG, "bad arguments");
    if (n_windows == 0) return TBA_OK;
    for (i64 i = 0; i < n_windows; i++)
        if (starts[i] < 0 || starts[i] + width > n_values || (kind != 0 && starts[i] >= n_values))
            return set_err(TBA_E_ARG, "window outside the arrays");
    HIP_TRY(hipSetDevice(e->device));
    const size_t nb = (size_t)n_values * 8;
    Tmp d_m, d_r, d_a, d_rv, d_av, d_s, d_o;
    if (d_m.alloc(nb) || d_r.alloc(nb) || d_a.alloc(nb) || d_rv.alloc(nb) || d_av.alloc(nb) ||
        d_s.alloc((size_t)n_windows * 8) || d_o.alloc((size_t)n_windows * 8))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_m.p, means, nb, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_r.p, ref_means, nb, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_a.p, alt_means, nb, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_rv.p, ref_vars, nb, hipMemcpyHostToDevice));
    if (alt_vars) C_TRY(hipMemcpy(d_av.p, alt_vars, nb, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_s.p, starts, (size_t)n_windows * 8, hipMemcpyHostToDevice));
    k_c_llh_windows<<<grid_for(n_windows), 256, 0, e->stream>>>(kind, d_m.as<double>(),
        d_r.as<double>(), d_a.as<double>(), d_rv.as<double>(), d_av.as<double>(), width,
        d_s.as<i64>(), n_windows, par ? par[0] : 0.0, par ? par[1] : 0.0, par ? par[2] : 0.0,
        d_o.as<double>());
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    C_TRY(hipMemcpy(out, d_o.p, (size_t)n_windows * 8, hipMemcpyDeviceToHost));
    return TBA_OK;
}

extern "C" int tba_read_pvals(tba_engine *e, const double *means, const double *ref_means,
    const double *ref_sds, const int64_t *off, int64_t n_reads, int64_t fm_offset, int floor_out,
    double smallest_pval, double *pvals)
{
    if (!e || !means || !ref_means || !ref_sds || !off || !pvals || n_reads < 0 || fm_offset < 0 ||
        fm_offset > 64)
        return set_err(TBA_E_ARG, "bad arguments");
    if (n_reads == 0) return TBA_OK;
    if (off[0] != 0) return set_err(TBA_E_ARG, "offset arrays must start at 0");
    for (i64 i = 0; i < n_reads; i++)
        if (off[i + 1] < off[i]) return set_err(TBA_E_ARG, "offset arrays must be non-decreasing");
    const i64 total = off[n_reads];
    if (total == 0) return TBA_OK;
    HIP_TRY(hipSetDevice(e->device));
    const size_t nb = (size_t)total * 8;
    Tmp d_m, d_r, d_s, d_off, d_o;
    if (d_m.alloc(nb) || d_r.alloc(nb) || d_s.alloc(nb) || d_off.alloc((size_t)(n_reads + 1) * 8) || d_o.alloc(nb))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_m.p, means, nb, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_r.p, ref_means, nb, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_s.p, ref_sds, nb, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_off.p, off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice));
    k_read_pvals<<<grid_for(total), 256, 0, e->stream>>>(d_m.as<double>(), d_r.as<double>(),
        d_s.as<double>(), d_off.as<i64>(), n_reads, total, fm_offset, floor_out, smallest_pval, d_o.as<double>());
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    C_TRY(hipMemcpy(pvals, d_o.p, nb, hipMemcpyDeviceToHost));
    return TBA_OK;
}

// testable slice of every read -> CSR offsets into a packed copy of (means, levels); one thread
// per read writes its (start, count), the host scans (a handful of values per read)
__global__ void k_denovo_pack(const ReadState *rs, i64 n_reads, const DevParams *dp,
    const double *bm, const double *refm, const double *refs, const i64 *pk_off, double *pm,
    double *pr, double *ps)
{
    const ReadState &r = rs[blockIdx.y];
    const i64 cp = dp->central_pos, dn = dp->kmer_width - dp->central_pos - 1;
    const i64 cnt = pk_off[blockIdx.y + 1] - pk_off[blockIdx.y];
    (void)n_reads;
    for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < cnt; k += (i64)gridDim.x * 256) {
        const i64 src = r.ref_off + cp + k, dst = pk_off[blockIdx.y] + k;
        pm[dst] = bm[src]; pr[dst] = refm[src]; ps[dst] = refs[src];
    }
    (void)dn;
}
__global__ void k_denovo_unpack(const ReadState *rs, const DevParams *dp, const i64 *pk_off,
    const double *pp, double *out)
{
    const ReadState &r = rs[blockIdx.y];
    const i64 cp = dp->central_pos;
    const i64 cnt = pk_off[blockIdx.y + 1] - pk_off[blockIdx.y];
    for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < r.B; k += (i64)gridDim.x * 256) {
        const i64 q = k - cp;
        out[r.ref_off + k] = (q >= 0 && q < cnt) ? pp[pk_off[blockIdx.y] + q] : NAN;
    }
}

extern "C" int tba_batch_de_novo_stats(tba_engine *e, int64_t fm_offset, double smallest_pval,
                                       double *pvals, int64_t n_values)
{
    if (!e || !e->have_batch || !e->finished) return set_err(TBA_E_STATE, "no finished batch");
    if (!pvals || n_values < e->B_tot || fm_offset < 0 || fm_offset > 64) return set_err(TBA_E_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->B_tot == 0) return 0;
    const size_t N = (size_t)e->n_reads;
    const i64 K = e->hp.kmer_width, cp = e->hp.central_pos, dn = K - cp - 1;
    // per-base means of the final signal (the Events table's norm_mean), on the device
    if (e->d_stat.ensure((size_t)e->B_tot * 16)) return set_err(TBA_E_NOMEM, "hipMalloc failed");
    double *d_m = e->d_stat.as<double>(), *d_s = d_m + e->B_tot;
    const unsigned gB = (unsigned)std::min<i64>(std::max<i64>((e->max_B + 255) / 256, 1), 128);
    k_base_stats<<<dim3(gB, (unsigned)N), 256, 0, e->stream>>>(e->d_rs.as<ReadState>(),
        e->d_dp.as<DevParams>(), e->hp.o.skip_norm_out ? nullptr : e->d_norm_out.as<double>(),
        e->d_norm.as<double>(), e->d_segs.as<i64>(), d_m, d_s);
    // testable positions of every successful read, packed
    std::vector<ReadState> rs(N);
    HIP_TRY(hipMemcpy(rs.data(), e->d_rs.p, N * sizeof(ReadState), hipMemcpyDeviceToHost));
    std::vector<i64> pk(N + 1, 0);
    for (size_t i = 0; i < N; i++) {
        i64 cnt = rs[i].status == TBA_OK ? rs[i].B - cp - dn : 0;
        // a read shorter than one Fisher window: "P-values vector too short" in the reference
        if (cnt < 2 * fm_offset + 1 || cnt < 1) cnt = 0;
        pk[i + 1] = pk[i] + cnt;
    }
    const i64 total = pk[N];
    Tmp d_pk, d_pm, d_pr, d_ps, d_pp, d_out;
    if (d_pk.alloc((N + 1) * 8) || d_pm.alloc((size_t)total * 8) || d_pr.alloc((size_t)total * 8) ||
        d_ps.alloc((size_t)total * 8) || d_pp.alloc((size_t)total * 8) || d_out.alloc((size_t)e->B_tot * 8))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    HIP_TRY(hipMemcpyAsync(d_pk.p, pk.data(), (N + 1) * 8, hipMemcpyHostToDevice, e->stream));
    if (total > 0) {
        k_denovo_pack<<<dim3(gB, (unsigned)N), 256, 0, e->stream>>>(e->d_rs.as<ReadState>(), e->n_reads,
            e->d_dp.as<DevParams>(), d_m, e->d_refm.as<double>(), e->d_refs.as<double>(),
            d_pk.as<i64>(), d_pm.as<double>(), d_pr.as<double>(), d_ps.as<double>());
        k_read_pvals<<<grid_for(total), 256, 0, e->stream>>>(d_pm.as<double>(), d_pr.as<double>(),
            d_ps.as<double>(), d_pk.as<i64>(), (i64)N, total, fm_offset, 1, smallest_pval, d_pp.as<double>());
    }
    k_denovo_unpack<<<dim3(gB, (unsigned)N), 256, 0, e->stream>>>(e->d_rs.as<ReadState>(),
        e->d_dp.as<DevParams>(), d_pk.as<i64>(), d_pp.as<double>(), d_out.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(pvals, d_out.p, (size_t)e->B_tot * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int tba_selftest_division(tba_engine *e, const double *a, const double *b, int64_t n,
                                     double *out)
{
    if (!e || !a || !b || !out || n < 1) return set_err(TBA_E_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(e->device));
    Tmp d_a, d_b, d_o;
    if (d_a.alloc((size_t)n * 8) || d_b.alloc((size_t)n * 8) || d_o.alloc((size_t)n * 8))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_a.p, a, (size_t)n * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_b.p, b, (size_t)n * 8, hipMemcpyHostToDevice));
    k_c_div_check<<<grid_for(n), 256, 0, e->stream>>>(d_a.as<double>(), d_b.as<double>(), n, d_o.as<double>());
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    C_TRY(hipMemcpy(out, d_o.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return TBA_OK;
}

extern "C" int tba_selftest_approx_quotient(tba_engine *e, const double *a, const double *b,
                                            int64_t n, double *out)
{
    if (!e || !a || !b || !out || n < 1) return set_err(TBA_E_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(e->device));
    Tmp d_a, d_b, d_o;
    if (d_a.alloc((size_t)n * 8) || d_b.alloc((size_t)n * 8) || d_o.alloc((size_t)n * 8))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    C_TRY(hipMemcpy(d_a.p, a, (size_t)n * 8, hipMemcpyHostToDevice));
    C_TRY(hipMemcpy(d_b.p, b, (size_t)n * 8, hipMemcpyHostToDevice));
    k_c_rcp_check<<<grid_for(n), 256, 0, e->stream>>>(d_a.as<double>(), d_b.as<double>(), n, d_o.as<double>());
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    C_TRY(hipMemcpy(out, d_o.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return TBA_OK;
}

// ---- ts.identify_stalls (tombo_stats.py:269-368), one read, host buffers --------------------
extern "C" int tba_identify_stalls(tba_engine *e, const void *raw, int raw_dtype, int64_t n,
    int64_t window_size, int64_t n_windows, int64_t mini_window_size, double threshold,
    int64_t min_consecutive_obs, int64_t edge_buffer, int64_t *ints, int64_t cap, int64_t *n_ints)
{
    if (!e || !raw || n < 1 || !n_ints || (cap > 0 && !ints)) return set_err(TBA_E_ARG, "bad arguments");
    if (raw_dtype < TBA_RAW_F64 || raw_dtype > TBA_RAW_I16) return set_err(TBA_E_ARG, "unknown raw dtype");
    if (n_windows < 2 || n_windows > 16 || mini_window_size < 1 ||
        window_size != n_windows * mini_window_size || min_consecutive_obs < 0)
        return set_err(TBA_E_ARG, "bad stall detection parameters");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    *n_ints = 0;
    if (n < window_size) return TBA_OK; // tombo_stats.py:305-308
    ReadState r;
    memset(&r, 0, sizeof(r));
    r.n_raw = n;
    r.status = TBA_OK;
    DevParams hp;
    memset(&hp, 0, sizeof(hp));
    hp.o.detect_stalls = 1;
    hp.o.stall_window_size = window_size; hp.o.stall_n_windows = n_windows;
    hp.o.stall_mini_window_size = mini_window_size; hp.o.stall_threshold = threshold;
    hp.o.stall_min_consecutive_obs = min_consecutive_obs; hp.o.stall_edge_buffer = edge_buffer;
    const i64 dev_cap = n / (min_consecutive_obs + 1) + 2;
    Tmp d_r, d_p, d_raw, d_csum, d_bits, d_ints;
    if (d_r.alloc(sizeof(r)) || d_p.alloc(sizeof(hp)) || d_raw.alloc((size_t)n * raw_elem_bytes(raw_dtype)) ||
        d_csum.alloc((size_t)(n + 2) * 8) || d_bits.alloc((size_t)(n / 64 + 2) * 8) ||
        d_ints.alloc((size_t)dev_cap * 16))
        return set_err(TBA_E_NOMEM, "hipMalloc failed");
    hipStream_t s = e->stream;
    C_TRY(hipMemcpyAsync(d_r.p, &r, sizeof(r), hipMemcpyHostToDevice, s));
    C_TRY(hipMemcpyAsync(d_p.p, &hp, sizeof(hp), hipMemcpyHostToDevice, s));
    C_TRY(hipMemcpyAsync(d_raw.p, raw, (size_t)n * raw_elem_bytes(raw_dtype), hipMemcpyHostToDevice, s));
    ReadState *rs = d_r.as<ReadState>();
    const DevParams *dp = d_p.as<DevParams>();
    if (raw_dtype == TBA_RAW_I16 && window_size <= SI_MAXW) {
        const unsigned gq = (unsigned)std::min<i64>(std::max<i64>((n + SI_T - 1) / SI_T, 1), 1024);
        if (n_windows == 7) k_stall_metric_i16<7><<<dim3(gq, 1), 256, 0, s>>>(rs, dp, d_raw.as<int16_t>(), d_bits.as<u64>());
        else k_stall_metric_i16<0><<<dim3(gq, 1), 256, 0, s>>>(rs, dp, d_raw.as<int16_t>(), d_bits.as<u64>());
    } else {
        RAW_DISPATCH(raw_dtype, (k_cumsum_scores<32, RT, 1><<<1, 256, 0, s>>>(rs, 1, dp, d_raw.as<RT>(), d_csum.as<double>())));
        const unsigned gq2 = (unsigned)std::min<i64>(std::max<i64>((n + SM_T - 1) / SM_T, 1), 1024);
        if (n_windows == 7) k_stall_metric<7><<<dim3(gq2, 1), 256, 0, s>>>(rs, dp, d_csum.as<double>(), d_bits.as<u64>());
        else k_stall_metric<0><<<dim3(gq2, 1), 256, 0, s>>>(rs, dp, d_csum.as<double>(), d_bits.as<u64>());
    }
    k_stall_runs<<<dim3(grid_for(n / 64 + 1), 1), 256, 0, s>>>(rs, dp, d_bits.as<u64>(), d_ints.as<i64>());
    k_stall_merge<<<1, 64, 0, s>>>(rs, 1, dp, d_ints.as<i64>());
    C_TRY(hipGetLastError());
    C_TRY(hipMemcpyAsync(&r, d_r.p, sizeof(r), hipMemcpyDeviceToHost, s));
    C_TRY(hipStreamSynchronize(s));
    if (r.status != TBA_OK) return r.status;
    *n_ints = r.n_stall;
    if (r.n_stall > cap) return set_err(TBA_E_ARG, "interval buffer too small (n_ints holds the count)");
    if (r.n_stall > 0) C_TRY(hipMemcpy(ints, d_ints.p, (size_t)r.n_stall * 16, hipMemcpyDeviceToHost));
    return TBA_OK;
}

// self-test of the device-side subsample: out[t] = image of t under the keyed permutation of
// [0, n) that read `read_index` of a batch would use under `seed` (t < count <= n)
__global__ void k_c_perm_check(i64 n, u64 seed, i64 read_index, i64 count, i64 *out)
{
    const u64 key = subsample_key(seed, read_index);
    for (i64 t = (i64)blockIdx.x * 256 + threadIdx.x; t < count; t += (i64)gridDim.x * 256)
        out[t] = keyed_perm(t, n, key);
}
extern "C" int tba_selftest_subsample(tba_engine *e, int64_t n, uint64_t seed, int64_t read_index,
                                      int64_t count, int64_t *out)
{
    if (!e || !out || n < 1 || count < 1 || count > n) return set_err(TBA_E_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(e->device));
    Tmp d_o;
    if (d_o.alloc((size_t)count * 8)) return set_err(TBA_E_NOMEM, "hipMalloc failed");
    k_c_perm_check<<<grid_for(count), 256, 0, e->stream>>>(n, seed, read_index, count, d_o.as<i64>());
    C_TRY(hipGetLastError());
    C_TRY(hipStreamSynchronize(e->stream));
    C_TRY(hipMemcpy(out, d_o.p, (size_t)count * 8, hipMemcpyDeviceToHost));
    return TBA_OK;
}

// ---- host-side packer: per-read arrays -> the CSR buffers of tba_batch_upload_async ----------
// (the reader side of the reference's worker pool, resquiggle.py:1385-1486: one Python thread
// cannot copy 100 k reads/s into a batch; this does it with n_threads native threads, GIL released)
#include <thread>
extern "C" int tba_pack_reads(int64_t n_reads, const void *const *raw_ptrs, int raw_dtype,
    int reverse, const int64_t *raw_off, void *raw_out, const char *const *seq_ptrs,
    const int64_t *seq_off, uint8_t *seq_out, int n_threads)
{
    if (n_reads < 0 || !raw_off || !seq_off || (n_reads > 0 && (!raw_ptrs || !seq_ptrs || !raw_out || !seq_out)))
        return set_err(TBA_E_ARG, "bad arguments");
    if (raw_dtype < TBA_RAW_F64 || raw_dtype > TBA_RAW_I16) return set_err(TBA_E_ARG, "unknown raw dtype");
    const size_t eb = raw_elem_bytes(raw_dtype);
    // ACGT -> 0..3, anything else 255 (the engine reports TBA_INVALID_SEQ).  A function-local static
    // with an initialiser is built once under the C++11 guard: callers pack from several threads
    // (ctypes releases the GIL; ReadFeeder.prefetch runs on a helper thread).
    struct CodeTable {
        unsigned char c[256];
        CodeTable() { for (int i = 0; i < 256; i++) c[i] = 255; c[(int)'A'] = 0; c[(int)'C'] = 1; c[(int)'G'] = 2; c[(int)'T'] = 3; }
    };
    static const CodeTable table;
    const unsigned char *code = table.c;
    auto work = [&](i64 a, i64 b) {
        for (i64 i = a; i < b; i++) {
            const i64 n = raw_off[i + 1] - raw_off[i], m = seq_off[i + 1] - seq_off[i];
            char *dst = (char *)raw_out + (size_t)raw_off[i] * eb;
            const char *src = (const char *)raw_ptrs[i];
            if (!reverse) memcpy(dst, src, (size_t)n * eb);
            else if (eb == 2) { const int16_t *q = (const int16_t *)src; int16_t *d = (int16_t *)dst; for (i64 k = 0; k < n; k++) d[k] = q[n - 1 - k]; }
            else if (eb == 4) { const float *q = (const float *)src; float *d = (float *)dst; for (i64 k = 0; k < n; k++) d[k] = q[n - 1 - k]; }
            else { const double *q = (const double *)src; double *d = (double *)dst; for (i64 k = 0; k < n; k++) d[k] = q[n - 1 - k]; }
            const unsigned char *sq = (const unsigned char *)seq_ptrs[i];
            uint8_t *so = seq_out + seq_off[i];
            for (i64 k = 0; k < m; k++) so[k] = code[sq[k]];
        }
    };
    int nt = std::max(1, std::min<int>(n_threads, (int)std::min<i64>(n_reads, 256)));
    if (nt == 1) { work(0, n_reads); return 0; }
    // cut by bytes, not by reads: the reads of a sorted batch differ in length
    std::vector<std::thread> th;
    const i64 tot = raw_off[n_reads] * (i64)eb + seq_off[n_reads];
    i64 a = 0;
    for (int t = 0; t < nt; t++) {
        i64 b = a;
        const i64 goal = tot / nt * (t + 1);
        while (b < n_reads && (t == nt - 1 || raw_off[b + 1] * (i64)eb + seq_off[b + 1] <= goal)) b++;
        if (t == nt - 1) b = n_reads;
        if (b > a) th.emplace_back(work, a, b);
        a = b;
    }
    for (auto &x : th) x.join();
    return 0;
}

// sizeof of the ABI structs, so that a binding can check its mirrors without a C compiler
// ---- synthetic reads on the device (k_synth.h) --------------------------------------------------
struct tba_synth {
    int device = 0;
    hipStream_t stream = nullptr;
    i64 kmer_width = 0;
    i64 n_reads = 0, S_tot = 0, seq_tot = 0;
    int raw_dtype = TBA_RAW_I16;
    DevBuf d_kmeans, d_sp, d_seq_off, d_base_off, d_raw_off, d_seq, d_starts, d_nraw, d_raw;
    PinBuf h_sp, h_off, h_nraw;
};

extern "C" int tba_synth_create(int device, const double *kmer_means, int64_t kmer_width, tba_synth **out)
{
    if (!out || !kmer_means || kmer_width < 1 || kmer_width > 12) return set_err(TBA_E_ARG, "bad arguments");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
        return set_err(TBA_E_HIP, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    tba_synth *g = new tba_synth();
    g->device = device;
    g->kmer_width = kmer_width;
    const size_t n = (size_t)1 << (2 * kmer_width);
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess || g->d_kmeans.ensure(n * 8) ||
        hipMemcpy(g->d_kmeans.p, kmer_means, n * 8, hipMemcpyHostToDevice) != hipSuccess) {
        g->d_kmeans.release();
        if (g->stream) (void)hipStreamDestroy(g->stream);
        delete g;
        return set_err(TBA_E_HIP, "tba_synth_create: stream / model upload failed");
    }
    *out = g;
    return 0;
}

extern "C" void tba_synth_destroy(tba_synth *g)
{
    if (!g) return;
    (void)hipSetDevice(g->device);
    (void)hipStreamSynchronize(g->stream);
    for (DevBuf *b : {&g->d_kmeans, &g->d_sp, &g->d_seq_off, &g->d_base_off, &g->d_raw_off, &g->d_seq,
                      &g->d_starts, &g->d_nraw, &g->d_raw})
        b->release();
    g->h_sp.release(); g->h_off.release(); g->h_nraw.release();
    (void)hipStreamDestroy(g->stream);
    delete g;
}

// the dwell thresholds and the noise constant of k_synth.h (plain IEEE operations only: the numpy
// restatement builds the same values)
static void synth_fill(SynthParams &sp, const tba_synth_params *p, i64 kmer_width)
{
    memset(&sp, 0, sizeof(sp));
    sp.mean_dwell = p->mean_dwell; sp.min_dwell = p->min_dwell; sp.n_lead = p->n_lead; sp.n_trail = p->n_trail;
    sp.scale = p->scale; sp.offset = p->offset; sp.noise_sd = p->noise_sd;
    sp.dac_per_pa = p->dac_per_pa; sp.dac_offset = p->dac_offset;
    sp.noise_norm = 1.0 / std::sqrt((65536.0 * 65536.0 - 1.0) / 3.0);
    sp.reverse = p->reverse; sp.kmer_width = (i32)kmer_width;
    const double q = 1.0 - 1.0 / (double)p->mean_dwell;
    double t = 1.0;
    for (int k = 0; k < SYNTH_DWELL_MAX; k++) {
        t = t * q;                                   // q^(k+1) = P(dwell > k + 1)
        const double v = std::floor(4294967296.0 * (1.0 - t));
        sp.thr[k] = v >= 4294967295.0 ? 0xffffffffu : (u32)v;
    }
}

extern "C" int tba_synth_dwell_thresholds(const tba_synth_params *p, uint32_t *thr, int64_t n, double *noise_norm)
{
    if (!p || !thr || n < 0 || p->mean_dwell < 1) return set_err(TBA_E_ARG, "bad arguments");
    SynthParams sp;
    synth_fill(sp, p, 1);
    for (i64 k = 0; k < n && k < SYNTH_DWELL_MAX; k++) thr[k] = sp.thr[k];
    if (noise_norm) *noise_norm = sp.noise_norm;
    return 0;
}

extern "C" int tba_synth_generate(tba_synth *g, const tba_synth_params *p, uint64_t seed, int64_t first_read,
                                  int64_t n_reads, const int64_t *n_bases, int raw_dtype,
                                  int64_t *raw_off, int64_t *seq_off, const void **d_raw, const uint8_t **d_seq)
{
    if (!g || !p || n_reads <= 0 || !n_bases || !raw_off || !seq_off || !d_raw || !d_seq)
        return set_err(TBA_E_ARG, "bad arguments");
    if (raw_dtype != TBA_RAW_I16 && raw_dtype != TBA_RAW_F64) return set_err(TBA_E_ARG, "raw dtype must be TBA_RAW_I16 or TBA_RAW_F64");
    if (p->mean_dwell < 1 || p->min_dwell < 1 || p->min_dwell > SYNTH_DWELL_MAX || p->n_lead < 0 || p->n_trail < 0)
        return set_err(TBA_E_ARG, "bad synthesis parameters");
    HIP_TRY(hipSetDevice(g->device));
    const i64 n = n_reads, K = g->kmer_width;
    hipStream_t s = g->stream;
    HIP_TRY(hipStreamSynchronize(s));
    const size_t N = (size_t)n;
    if (g->h_sp.ensure(sizeof(SynthParams)) || g->h_off.ensure(3 * (N + 1) * 8) || g->h_nraw.ensure(N * 8)) return TBA_E_NOMEM;
    synth_fill(*g->h_sp.as<SynthParams>(), p, K);
    i64 *h_seq_off = g->h_off.as<i64>(), *h_base_off = h_seq_off + (n + 1), *h_raw_off = h_base_off + (n + 1);
    h_seq_off[0] = h_base_off[0] = 0;
    for (i64 i = 0; i < n; i++) {
        if (n_bases[i] < 1 || n_bases[i] * SYNTH_DWELL_MAX > 0x7fff0000ll) return set_err(TBA_E_ARG, "bad read length");
        h_seq_off[i + 1] = h_seq_off[i] + n_bases[i] + K - 1;
        h_base_off[i + 1] = h_base_off[i] + n_bases[i];
    }
    const i64 B_tot = h_base_off[n];
    if (g->d_sp.ensure(sizeof(SynthParams)) || g->d_seq_off.ensure((N + 1) * 8) || g->d_base_off.ensure((N + 1) * 8) ||
        g->d_raw_off.ensure((N + 1) * 8) || g->d_seq.ensure((size_t)h_seq_off[n]) ||
        g->d_starts.ensure((size_t)(B_tot + n) * 4) || g->d_nraw.ensure(N * 8))
        return TBA_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(g->d_sp.p, g->h_sp.p, sizeof(SynthParams), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(g->d_seq_off.p, h_seq_off, (N + 1) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(g->d_base_off.p, h_base_off, (N + 1) * 8, hipMemcpyHostToDevice, s));
    k_synth_plan<<<(unsigned)n, SYNTH_NT, 0, s>>>(g->d_sp.as<SynthParams>(), seed, first_read, g->d_seq_off.as<i64>(),
        g->d_base_off.as<i64>(), g->d_seq.as<uint8_t>(), g->d_starts.as<i32>(), g->d_nraw.as<i64>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(g->h_nraw.p, g->d_nraw.p, N * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const i64 *nr = g->h_nraw.as<i64>();
    h_raw_off[0] = 0;
    i64 max_b = 0;
    for (i64 i = 0; i < n; i++) { h_raw_off[i + 1] = h_raw_off[i] + nr[i]; max_b = std::max(max_b, n_bases[i]); }
    const i64 S_tot = h_raw_off[n];
    if (g->d_raw.ensure((size_t)S_tot * raw_elem_bytes(raw_dtype) + 64)) return TBA_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(g->d_raw_off.p, h_raw_off, (N + 1) * 8, hipMemcpyHostToDevice, s));
    const unsigned gx = (unsigned)std::min<i64>(std::max<i64>((max_b + 4 * SYNTH_NT - 1) / (4 * SYNTH_NT), 1), 64);
    if (raw_dtype == TBA_RAW_I16)
        k_synth_raw<int16_t><<<dim3(gx, (unsigned)n), SYNTH_NT, 0, s>>>(g->d_sp.as<SynthParams>(), seed, first_read,
            g->d_seq_off.as<i64>(), g->d_base_off.as<i64>(), g->d_raw_off.as<i64>(), g->d_seq.as<uint8_t>(),
            g->d_starts.as<i32>(), g->d_kmeans.as<double>(), g->d_raw.as<int16_t>());
    else
        k_synth_raw<double><<<dim3(gx, (unsigned)n), SYNTH_NT, 0, s>>>(g->d_sp.as<SynthParams>(), seed, first_read,
            g->d_seq_off.as<i64>(), g->d_base_off.as<i64>(), g->d_raw_off.as<i64>(), g->d_seq.as<uint8_t>(),
            g->d_starts.as<i32>(), g->d_kmeans.as<double>(), g->d_raw.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    memcpy(raw_off, h_raw_off, (N + 1) * 8);
    memcpy(seq_off, h_seq_off, (N + 1) * 8);
    g->n_reads = n; g->S_tot = S_tot; g->seq_tot = h_seq_off[n]; g->raw_dtype = raw_dtype;
    *d_raw = g->d_raw.p;
    *d_seq = g->d_seq.as<uint8_t>();
    return 0;
}

extern "C" int tba_synth_download(tba_synth *g, void *raw, uint8_t *seq)
{
    if (!g || g->n_reads <= 0) return set_err(TBA_E_STATE, "nothing generated");
    HIP_TRY(hipSetDevice(g->device));
    if (raw) HIP_TRY(hipMemcpy(raw, g->d_raw.p, (size_t)g->S_tot * raw_elem_bytes(g->raw_dtype), hipMemcpyDeviceToHost));
    if (seq) HIP_TRY(hipMemcpy(seq, g->d_seq.p, (size_t)g->seq_tot, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int tba_abi_sizes(int64_t *out, int64_t n)
{
    if (!out || n < 3) return set_err(TBA_E_ARG, "bad arguments");
    out[0] = (int64_t)sizeof(tba_params); out[1] = (int64_t)sizeof(tba_opts);
    out[2] = (int64_t)sizeof(tba_read_result);
    if (n >= 4) out[3] = TBA_ABI_VERSION;
    return 0;
}

// the other direction: slices of one flat result array -> one destination array per read
extern "C" int tba_unpack_reads(int64_t n_reads, const void *src, int64_t elem_bytes,
    const int64_t *src_off, const int64_t *count, void *const *dst_ptrs, int n_threads)
{
    if (n_reads < 0 || elem_bytes < 1 || (n_reads > 0 && (!src || !src_off || !count || !dst_ptrs)))
        return set_err(TBA_E_ARG, "bad arguments");
    auto work = [&](i64 a, i64 b) {
        for (i64 i = a; i < b; i++)
            if (count[i] > 0 && dst_ptrs[i])
                memcpy(dst_ptrs[i], (const char *)src + (size_t)src_off[i] * (size_t)elem_bytes,
                       (size_t)count[i] * (size_t)elem_bytes);
    };
    const int nt = std::max(1, std::min<int>(n_threads, (int)std::min<i64>(n_reads, 256)));
    if (nt == 1) { work(0, n_reads); return 0; }
    std::vector<i64> acc((size_t)n_reads + 1, 0);
    for (i64 i = 0; i < n_reads; i++) acc[(size_t)i + 1] = acc[(size_t)i] + std::max<i64>(count[i], 0) + 64;
    std::vector<std::thread> th;
    i64 a = 0;
    for (int t = 0; t < nt; t++) {
        const i64 goal = acc[(size_t)n_reads] / nt * (t + 1);
        i64 b = t == nt - 1 ? n_reads : (i64)(std::upper_bound(acc.begin(), acc.end(), goal) - acc.begin()) - 1;
        b = std::max(a, std::min(b, n_reads));
        if (b > a) th.emplace_back(work, a, b);
        a = b;
    }
    for (auto &x : th) x.join();
    return 0;
}

extern "C" int tba_engine_set_dispatch(tba_engine *e, int64_t small_batch_reads, int64_t tb_wave_below)
{
    if (!e) return set_err(TBA_E_ARG, "engine is NULL");
    if (small_batch_reads >= 0) e->small_batch = small_batch_reads;
    if (tb_wave_below >= 0) e->tb_wave_below = tb_wave_below;
    return 0;
}
extern "C" int tba_engine_get_dispatch(tba_engine *e, int64_t *small_batch_reads, int64_t *tb_wave_below)
{
    if (!e) return set_err(TBA_E_ARG, "engine is NULL");
    if (small_batch_reads) *small_batch_reads = e->small_batch;
    if (tb_wave_below) *tb_wave_below = e->tb_wave_below;
    return 0;
}

extern "C" int tba_engine_set_side_stream(tba_engine *e, int mode)
{
    if (!e || mode < -1 || mode > 1) return set_err(TBA_E_ARG, "bad arguments");
    e->side_mode = mode;
    return 0;
}
extern "C" int tba_engine_last_side_stream(tba_engine *e)
{
    return e ? (e->last_side ? 1 : 0) : 0;
}

extern "C" int tba_engine_set_sharing(tba_engine *e, int n_engines)
{
    if (!e || n_engines < 1) return set_err(TBA_E_ARG, "bad arguments");
    e->n_sharing = n_engines;
    return 0;
}

// device bytes this engine's grow-only buffers hold right now (a planner budgets against the
// free memory PLUS this: a batch that fitted before still fits)
extern "C" int tba_engine_held_bytes(tba_engine *e, int64_t *bytes)
{
    if (!e || !bytes) return set_err(TBA_E_ARG, "bad arguments");
    size_t tot = 0;
    DevBuf *all[] = {&e->d_rs, &e->d_dp, &e->d_kmeans, &e->d_ksds, &e->d_raw, &e->d_norm, &e->d_norm_out, &e->d_csum,
                     &e->d_score, &e->d_state, &e->d_cpts, &e->d_evm, &e->d_seq, &e->d_refm, &e->d_refs, &e->d_bst,
                     &e->d_lo, &e->d_hi, &e->d_readtb, &e->d_dpsegs, &e->d_segs, &e->d_win, &e->d_absz,
                     &e->d_sv_in, &e->d_samp, &e->d_stall, &e->d_lastrow, &e->d_startvals, &e->d_smoves,
                     &e->d_moves, &e->d_dscr, &e->d_wide, &e->d_stat, &e->d_res, &e->d_segs32, &e->d_skipq,
                     &e->d_stall_csum, &e->d_stall_bits};
    for (DevBuf *b : all) tot += b->cap;
    *bytes = (int64_t)tot;
    return 0;
}
