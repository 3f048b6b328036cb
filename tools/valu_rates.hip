// Issue cost of the VALU instructions k_dp is made of, measured: waves per SIMD x independent
// instructions of one kind in a loop, cycles per wave-instruction per SIMD out of s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -w -o alt_builds/valu_rates tools/valu_rates.hip && alt_builds/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int OP>
__global__ __launch_bounds__(256) void k(long long *out, int iters, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = seed * 0.5, c = seed * 0.25;
    float f0 = (float)a0, f1 = (float)a1;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    unsigned long long m = 0;
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 1) { REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 2) { REP8(asm volatile("v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %4\n v_max_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 3) { REP8(asm volatile("v_cmp_gt_f64 vcc, %0, %4\n v_cmp_gt_f64 vcc, %1, %4\n v_cmp_gt_f64 vcc, %2, %4\n v_cmp_gt_f64 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 4) { REP8(asm volatile("v_cmp_ne_u64 vcc, %0, %4\n v_cmp_ne_u64 vcc, %1, %4\n v_cmp_ne_u64 vcc, %2, %4\n v_cmp_ne_u64 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 5) { REP8(asm volatile("v_cmp_ne_u32 vcc, %0, %4\n v_cmp_ne_u32 vcc, %1, %4\n v_cmp_ne_u32 vcc, %2, %4\n v_cmp_ne_u32 vcc, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0) : "vcc");) }
        if (OP == 6) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        if (OP == 7) { REP8(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        if (OP == 8) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : : "vcc");) }
        if (OP == 9) { REP8(asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(f0), "+v"(f1) : "v"(f0));) }
        if (OP == 10) { REP8(asm volatile("v_cvt_f32_f64 %0, %2\n v_cvt_f32_f64 %1, %3\n v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5" : "+v"(f0), "+v"(f1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if (OP == 11) { REP8(asm volatile("v_min_f64 %0, |%0|, %4\n v_min_f64 %1, |%1|, %4\n v_min_f64 %2, |%2|, %4\n v_min_f64 %3, |%3|, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 12) { REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 13) { REP8(asm volatile("v_lshl_or_b32 %0, %1, 2, %0\n v_lshl_or_b32 %1, %2, 2, %1\n v_lshl_or_b32 %2, %3, 2, %2\n v_lshl_or_b32 %3, %0, 2, %3" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        if (OP == 14) { REP8(asm volatile("v_cmp_gt_f64 s[20:21], %0, %4\n v_cmp_gt_f64 s[22:23], %1, %4\n v_cmp_gt_f64 s[24:25], %2, %4\n v_cmp_gt_f64 s[26:27], %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        // dependent chains (latency): one accumulator
        if (OP == 20) { REP32(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(b));) }
        if (OP == 21) { REP32(asm volatile("v_max_f64 %0, %0, %1" : "+v"(a0) : "v"(b));) }
        if (OP == 22) { REP32(asm volatile("v_mov_b32 %0, %0" : "+v"(i0));) }
        if (OP == 23) { REP32(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));) }
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (a0 + a1 + a2 + a3 + f0 + f1 + i0 + i1 + i2 + i3 + (double)m == 12345.678) out[1] = 1;
}

template <int OP>
static void run(const char *name, long long *d, int waves_per_simd)
{
    const int iters = 20000;
    // one workgroup of 256 threads = one wavefront on every SIMD of a CU; `waves_per_simd` such
    // workgroups per CU over all 256 CUs.  Wall time of the whole launch (long against its ramp):
    // every SIMD executes waves_per_simd x iters x 32 instructions of the kind.
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256 * waves_per_simd, 256>>>(d, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<256 * waves_per_simd, 256>>>(d, iters, 1.0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long cyc = 0;
    hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
    const double n_inst = (double)iters * 32.0;
    const double clk_ghz = (double)cyc / (ms * 1e6);                       // block 0's own cycles over the launch's wall time
    printf("%-34s waves/SIMD %d: wave 0: %6.2f cycles per instruction; launch: %7.3f ms = %6.2f ns per instruction per SIMD "
           "(= %5.2f cycles at 2.4 GHz; wave-0 cycles / wall = %.2f GHz)\n", name, waves_per_simd, (double)cyc / n_inst,
           ms, ms * 1e6 / (n_inst * waves_per_simd), ms * 1e6 / (n_inst * waves_per_simd) * 2.4, clk_ghz);
}

int main()
{
    long long *d;
    hipMalloc(&d, 64);
    for (int w : {1, 4}) {
        run<0>("v_add_f64 (independent x4)", d, w);
        run<1>("v_fma_f64", d, w);
        run<12>("v_mul_f64", d, w);
        run<2>("v_max_f64", d, w);
        run<11>("v_min_f64 |x|", d, w);
        run<3>("v_cmp_gt_f64 -> vcc", d, w);
        run<14>("v_cmp_gt_f64 -> sgpr pair", d, w);
        run<4>("v_cmp_ne_u64 -> vcc", d, w);
        run<5>("v_cmp_ne_u32 -> vcc", d, w);
        run<6>("v_mov_b32", d, w);
        run<7>("v_mov_b32_dpp wave_shr (+s_nop/4)", d, w);
        run<8>("v_cndmask_b32", d, w);
        run<13>("v_lshl_or_b32", d, w);
        run<9>("v_add_f32", d, w);
        run<10>("v_cvt_f32_f64", d, w);
        run<20>("v_add_f64 dependent chain", d, w);
        run<21>("v_max_f64 dependent chain", d, w);
        run<23>("v_fma_f64 dependent chain", d, w);
        run<22>("v_mov_b32 dependent chain", d, w);
    }
    return 0;
}
