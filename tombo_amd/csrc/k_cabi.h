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

// c_new_means, _c_helper.pyx:59-71
__global__ void k_c_new_means(const double *sig, const i64 *segs, i64 n_segs, double *means)
{
    for (i64 s = (i64)blockIdx.x * blockDim.x + threadIdx.x; s < n_segs; s += (i64)gridDim.x * blockDim.x) {
        double acc = 0;
        for (i64 j = segs[s]; j < segs[s + 1]; j++) acc += sig[j];
        means[s] = acc / (double)(segs[s + 1] - segs[s]);
    }
}

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
        out[i] = div_by_recip(a[i], b[i], y);
    }
}
