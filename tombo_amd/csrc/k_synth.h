// Synthetic reads made on the device (bench / test support: no counterpart in the reference, whose
// benchmark input is FAST5 files; SURVEY.md section 8d describes the workload these stand in for).
// A job of a million distinct 10 kb reads is 190 GB of int16 samples: made by numpy on the host cores
// (tombo_amd/synth.py, ~1 ms per read) that is most of an hour, so the multi-GPU job of
// BASELINE.json's cfg5 draws every batch here instead, from a counter-based generator keyed by
// (batch seed, read, element) -- any rank produces the same batch for the same seed, no state is
// carried from batch to batch.
//
// The read is the one synth.synth_read describes -- uniform ACGT sequence, per-base level = the
// model's k-mer mean, dwell = max(min_dwell, Geometric(1 / mean_dwell)), level + noise per sample,
// n_lead samples around +0.5 before and n_trail around -0.5 after -- with every draw made of
// integer operations and a handful of IEEE double operations in a fixed order, so that the numpy
// restatement (synth.device_reads_reference) reproduces the samples bit for bit:
//   hash(x)   = splitmix64 finaliser
//   key(read) = hash(hash(seed) + read index)
//   draw(stream, i) = hash(key + (stream << 40) + i)   stream 0 bases, 1 dwells, 2 noise
//   base code = (draw >> 11) & 3
//   dwell     = 1 + #{k : thr[k] <= draw >> 32}, thr[k] = floor(2^32 (1 - q^(k+1))), q = 1 - 1/mean_dwell
//               built by repeated multiplication (no pow), SYNTH_DWELL_MAX entries
//   noise     = (sum of the draw's four 16-bit fields - 131070) * c, c = 1 / sqrt((65536^2 - 1) / 3)
//               (Irwin-Hall of four uniforms: unit variance, tails cut at 3.46)
//   sample    = ((level + noise * sd) * scale + offset) [* dac_per_pa + dac_offset, rint -> int16]
#pragma once
#include "tba_common.h"

#define SYNTH_DWELL_MAX 256
#define SYNTH_NT 256

struct SynthParams {
    i64 mean_dwell, min_dwell, n_lead, n_trail;
    double scale, offset, noise_sd, dac_per_pa, dac_offset;
    double noise_norm;              // c above
    i32 reverse, kmer_width;        // reverse: samples written last to first (RNA acquisition order)
    u32 thr[SYNTH_DWELL_MAX];
};

__host__ __device__ __forceinline__ u64 synth_hash(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ u64 synth_key(u64 seed, i64 read) { return synth_hash(synth_hash(seed) + (u64)read); }
__device__ __forceinline__ u64 synth_draw(u64 key, int stream, i64 i) { return synth_hash(key + ((u64)stream << 40) + (u64)i); }

__device__ __forceinline__ double synth_noise(u64 key, i64 s, double c)
{
    const u64 h = synth_draw(key, 2, s);
    const i32 t = (i32)(h & 0xffff) + (i32)((h >> 16) & 0xffff) + (i32)((h >> 32) & 0xffff) + (i32)(h >> 48);
    return (double)(t - 131070) * c;
}

template <class RT> __device__ __forceinline__ RT synth_out(double x, const SynthParams &sp);
template <> __device__ __forceinline__ double synth_out<double>(double x, const SynthParams &sp)
{
    return x * sp.scale + sp.offset;
}
template <> __device__ __forceinline__ int16_t synth_out<int16_t>(double x, const SynthParams &sp)
{
    const double pa = x * sp.scale + sp.offset;
    double d = rint(pa * sp.dac_per_pa + sp.dac_offset);
    d = d > 32767.0 ? 32767.0 : (d < -32768.0 ? -32768.0 : d);
    return (int16_t)d;
}

// One workgroup per read: base codes, dwells and their running sum (the first sample of every base,
// counted from the end of the lead), the read's sample count.
// seq_off: n + 1 offsets of the code arrays (n_bases + K - 1 codes per read); base_off: n + 1 offsets
// of the per-base arrays (n_bases + 1 starts per read -> base_off[i] + i).
__global__ __launch_bounds__(SYNTH_NT) void k_synth_plan(const SynthParams *spp, u64 seed, i64 read0,
    const i64 *seq_off, const i64 *base_off, uint8_t *seq, i32 *starts, i64 *n_raw)
{
    __shared__ i32 s_w[SYNTH_NT / 64];
    __shared__ i32 s_run;
    const SynthParams &sp = *spp;
    const i64 rd = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u64 key = synth_key(seed, read0 + rd);
    const i64 so = seq_off[rd], n_codes = seq_off[rd + 1] - so;
    const i64 bo = base_off[rd] + rd, B = base_off[rd + 1] - base_off[rd];
    for (i64 j = tid; j < n_codes; j += SYNTH_NT) seq[so + j] = (uint8_t)((synth_draw(key, 0, j) >> 11) & 3);
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (i64 b0 = 0; b0 < B; b0 += SYNTH_NT) {
        const i64 b = b0 + tid;
        i32 dw = 0;
        if (b < B) {
            const u32 u = (u32)(synth_draw(key, 1, b) >> 32);
            int lo = 0, hi = SYNTH_DWELL_MAX;         // first k with thr[k] > u
            while (lo < hi) { const int m = (lo + hi) >> 1; if (sp.thr[m] <= u) lo = m + 1; else hi = m; }
            dw = 1 + lo;
            dw = dw < (i32)sp.min_dwell ? (i32)sp.min_dwell : dw;
        }
        i32 inc = dw;
        for (int d = 1; d < 64; d <<= 1) { const i32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        i32 base = s_run;
        for (int q = 0; q < w; q++) base += s_w[q];
        if (b < B) starts[bo + b] = base + inc - dw;
        __syncthreads();
        if (tid == SYNTH_NT - 1) s_run = base + inc;
        __syncthreads();
    }
    if (tid == 0) { starts[bo + B] = s_run; n_raw[rd] = sp.n_lead + (i64)s_run + sp.n_trail; }
}

// The samples: grid (x: slices of a read, y: read).  A thread per base writes that base's samples
// (consecutive threads, consecutive runs), then the lead and the trail.
template <class RT>
__global__ __launch_bounds__(SYNTH_NT) void k_synth_raw(const SynthParams *spp, u64 seed, i64 read0,
    const i64 *seq_off, const i64 *base_off, const i64 *raw_off, const uint8_t *seq, const i32 *starts,
    const double *kmer_means, RT *raw)
{
    const SynthParams &sp = *spp;
    const i64 rd = blockIdx.y;
    const u64 key = synth_key(seed, read0 + rd);
    const i64 bo = base_off[rd] + rd, B = base_off[rd + 1] - base_off[rd];
    const i64 S = raw_off[rd + 1] - raw_off[rd];
    RT *out = raw + raw_off[rd];
    const uint8_t *codes = seq + seq_off[rd];
    const int K = sp.kmer_width;
    const i64 stride = (i64)gridDim.x * SYNTH_NT;
    const double c = sp.noise_norm;
    for (i64 b = (i64)blockIdx.x * SYNTH_NT + threadIdx.x; b < B; b += stride) {
        i64 idx = 0;
        for (int j = 0; j < K; j++) idx = idx * 4 + codes[b + j];
        const double level = kmer_means[idx];
        const i64 s0 = sp.n_lead + starts[bo + b], s1 = sp.n_lead + starts[bo + b + 1];
        for (i64 s = s0; s < s1; s++) {
            const double x = level + synth_noise(key, s, c) * sp.noise_sd;
            out[sp.reverse ? S - 1 - s : s] = synth_out<RT>(x, sp);
        }
    }
    const i64 body_end = S - sp.n_trail;
    for (i64 t = (i64)blockIdx.x * SYNTH_NT + threadIdx.x; t < sp.n_lead + sp.n_trail; t += stride) {
        const bool head = t < sp.n_lead;
        const i64 s = head ? t : body_end + (t - sp.n_lead);
        const double x = (head ? 0.5 : -0.5) + synth_noise(key, s, c);
        out[sp.reverse ? S - 1 - s : s] = synth_out<RT>(x, sp);
    }
}
