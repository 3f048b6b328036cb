// Development aid: resident workgroups per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor) and
// register / LDS use (hipFuncGetAttributes) of the batch kernels, from the same headers the
// engine is built from.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o alt_builds/occupancy tools/occupancy.hip
//   gpurun -- alt_builds/occupancy
#include "../tombo_amd/csrc/tba_common.h"
#include "../tombo_amd/csrc/k_select.h"
#include "../tombo_amd/csrc/k_segment.h"
#include "../tombo_amd/csrc/k_detect.h"
#include "../tombo_amd/csrc/k_prep_raw.h"
#include "../tombo_amd/csrc/k_dp.h"
#include "../tombo_amd/csrc/k_tb_par.h"
#include "../tombo_amd/csrc/k_dp_multi.h"
#include "../tombo_amd/csrc/k_long.h"
#include "../tombo_amd/csrc/k_tail.h"
#include <cstdio>

template <class K> static void report(const char *name, K kernel, int threads)
{
    int nb = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, 0);
    hipFuncAttributes a{};
    hipError_t e2 = hipFuncGetAttributes(&a, reinterpret_cast<const void *>(kernel));
    printf("%-28s threads %4d  workgroups/CU %2d (waves/CU %2d)  vgpr %3d  lds %6zu  scratch %4zu  %s %s\n",
           name, threads, nb, nb * threads / 64, a.numRegs, a.sharedSizeBytes, a.localSizeBytes,
           e == hipSuccess ? "" : hipGetErrorString(e), e2 == hipSuccess ? "" : hipGetErrorString(e2));
}

int main()
{
    report("k_normalize<double>", k_normalize<double>, SEL_NT);
    report("k_normalize<int16_t>", k_normalize<int16_t>, SEL_NT);
    report("k_cumsum_scores<20>", k_cumsum_scores<20>, 256);
    report("k_cumsum_scores<32>", k_cumsum_scores<32>, 256);
    report("k_scores_ttest<double>", k_scores_ttest<double>, 256);
    report("k_peaks<2>", k_peaks<2>, SEL_NT);
    report("k_peaks<5>", k_peaks<5>, SEL_NT);
    report("k_detect<2,double>", k_detect<2, double>, 256);
    report("k_detect<2,int16_t>", k_detect<2, int16_t>, 256);
    report("k_pick", k_pick, SEL_NT);
    report("k_detect_tt<5,12,double>", k_detect_tt<5, 12, double>, SEL_NT);
    report("k_event_means<double>", k_event_means<double, 448>, 256);
    report("k_event_means<double, 1280>", k_event_means<double, 1280>, 256);
    report("k_dp<8,false>", k_dp<8, false>, 64);
    report("k_dp8_lowreg", k_dp8_lowreg, 64);
    report("k_dp<5,false>", k_dp<5, false>, 64);
    report("k_dp<12,false>", k_dp<12, false>, 64);
    report("k_dp_multi<4,2>", k_dp_multi<4, 2>, 64);
    report("k_dp_multi<8,4>", k_dp_multi<8, 4>, 64);
    report("k_dp_multi<8,2>", k_dp_multi<8, 2>, 64);
    report("k_dp_multi<10,2>", k_dp_multi<10, 2>, 64);
    report("k_cumsum_scores<32,i16,1>", k_cumsum_scores<32, int16_t, 1>, 256);
    report("k_stall_metric<7>", k_stall_metric<7>, 256);
    report("k_cumsum_scores_long<f64,0>", k_cumsum_scores_long<double, 0>, 256);
    report("k_main_tb_long", k_main_tb_long, 64);
    report("k_main_tb", k_main_tb, 64);
    report("k_main_tb_par<16>", k_main_tb_par<16>, 64);
    report("k_main_tb_par<64>", k_main_tb_par<64>, 64);
    report("k_tb_par_verify<16>", k_tb_par_verify<16>, 64);
    report("k_skip_dp<true>", k_skip_dp<true>, 64);
    report("k_skip_dp<false>", k_skip_dp<false>, 64);
    report("k_skip_plan", k_skip_plan, 64);
    report("k_theil_sen", k_theil_sen, SEL_NT);
    report("k_rescale_absz<true>", k_rescale_absz<true>, 256);
    report("k_rescale_absz<false>", k_rescale_absz<false>, 256);
    return 0;
}
