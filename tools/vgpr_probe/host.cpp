// host harness of the VGPR clobber probe: hipModuleLoad(<hsaco>), launch <kernel> over <n_wg> one-wavefront workgroups,
// report the waves whose registers did not hold their signature at the end
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv)
{
    const char *path = argv[1], *name = argv[2];
    unsigned n_wg = atoi(argv[3]); int base = atoi(argv[4]); int reps = argc > 5 ? atoi(argv[5]) : 1;
    hipModule_t mod; hipFunction_t fn;
    CK(hipModuleLoad(&mod, path));
    CK(hipModuleGetFunction(&fn, mod, name));
    unsigned *d_out; CK(hipMalloc((void **)&d_out, (size_t)(1 << 24) + (2 << 20)));   // results (<= 16 MiB) + 2 MiB of 0x5a5a5a5a for the LOADS kernels
    if ((size_t)n_wg * 16 > (size_t)(1 << 24)) { printf("too many waves\n"); return 1; }
    CK(hipMemset((char *)d_out + (1 << 24), 0x5a, 2 << 20));
    std::vector<unsigned> out((size_t)n_wg * 4);
    long bad_total = 0;
    for (int r = 0; r < reps; r++) {
        CK(hipMemset(d_out, 0xEE, (size_t)n_wg * 16));
        struct { void *out; int base; int pad; } args = {d_out, base, 0};
        size_t sz = sizeof(args);
        void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        CK(hipModuleLaunchKernel(fn, n_wg, 1, 1, 64, 1, 1, 0, 0, nullptr, cfg));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), d_out, (size_t)n_wg * 16, hipMemcpyDeviceToHost));
        long bad = 0, unwritten = 0;
        for (unsigned w = 0; w < n_wg; w++) {
            unsigned *o = &out[(size_t)w * 4];
            if (o[2] == 0xEEEEEEEEu) { unwritten++; continue; }
            if (o[2] != 0) {
                if (bad < 12) printf("  rep %d wave %u: %u registers changed; first v%u = 0x%08x (expected 0x%08x) hw_id 0x%08x\n", r, w, o[2], o[0], o[1], (0x40000000u | (w << 10)) + o[0], o[3]);
                bad++;
            }
        }
        bad_total += bad;
        printf("%s rep %d: %u waves, %ld with clobbered registers, %ld unwritten\n", name, r, n_wg, bad, unwritten);
    }
    printf("PROBE %s total clobbered waves %ld\n", name, bad_total);
    return 0;
}
