#include <hip/hip_runtime.h>
#include <cstdio>
int main(int argc, char **argv)
{
    float *out, *in; (void)hipMalloc(&out, 256 * 1024 * sizeof(float)); (void)hipMalloc(&in, 64 * 64 * sizeof(float)); (void)hipMemset(in, 0, 64 * 64 * sizeof(float));
    for (int a = 1; a < argc; ++a) {
        hipModule_t mod; hipFunction_t fn;
        if (hipModuleLoad(&mod, argv[a]) != hipSuccess || hipModuleGetFunction(&fn, mod, "k_regs") != hipSuccess) { printf("%s: load failed\n", argv[a]); continue; }
        for (int threads = 256; threads <= 1024; threads *= 2) {
            int iters = 2; void *args[] = {&out, &in, &iters};
            (void)hipModuleLaunchKernel(fn, 256, 1, 1, threads, 1, 1, 0, 0, args, 0); (void)hipDeviceSynchronize();
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            iters = 64;
            (void)hipEventRecord(e0); (void)hipModuleLaunchKernel(fn, 256, 1, 1, threads, 1, 1, 0, 0, args, 0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double nmfma = 256.0 * (threads / 64) * iters * 128.0;
            printf("%s waves/SIMD %d: %.3f ms  %.1f cycles/MFMA/SIMD @2.4GHz\n", argv[a], threads / 256, ms, (ms * 1e-3) * 2.4e9 / (nmfma / 1024.0)); fflush(stdout);
        }
    }
    return 0;
}
