// Micro-benchmark (see profiles/r01_notes.md): COMPILED twins of tools/ubench/mfma_regs.hip -- the same 8 chains x 16 distinct operand
// registers of v_mfma_f32_16x16x4_f32, with the operands (a) loaded from memory in front of the loop (the compiler leaves s_waitcnt vmcnt(N)
// between the MFMAs), (b) loaded and waited for in front of the loop, (c) made up in registers (no memory instruction in the kernel).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_compiled tools/ubench/mfma_compiled.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void k_regs(float *out, const float *in, int iters)
{
    const int lane = threadIdx.x & 63;
    float av[16], bv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (MODE == 2 || MODE == 4 || MODE == 6 || MODE == 7) {
            av[i] = (float)(lane * 0 + i) * 0.f;
            bv[i] = (float)(lane * 0 + i + 16) * 0.f;
            asm volatile("" : "+v"(av[i]), "+v"(bv[i]));
        } else {
            av[i] = in[i * 64 + lane];
            bv[i] = in[(16 + i) * 64 + lane];
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(av[i]), "+v"(bv[i]));
    }
    if (MODE == 3) {   // everything has landed before the loop, but the compiler does not know: it still places its vmcnt(N) in the loop
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)");
        __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 4) {   // registers only, but the waves of a SIMD enter the loop staggered
        if ((threadIdx.x >> 8) & 1) __builtin_amdgcn_s_sleep(100);
        if ((threadIdx.x >> 9) & 1) __builtin_amdgcn_s_sleep(50);
    }
    if (MODE == 6) {   // registers only, waves of a SIMD one s_sleep unit (64 clocks = two MFMAs) apart
        const int k = threadIdx.x >> 8;
        if (k & 1) __builtin_amdgcn_s_sleep(1);
        if (k & 2) __builtin_amdgcn_s_sleep(2);
    }
    if (MODE == 7) {   // ... 20 clocks apart
        const int k = threadIdx.x >> 8;
        if (k & 1) { asm volatile("s_nop 9"); asm volatile("s_nop 9"); }
        if (k & 2) { asm volatile("s_nop 9"); asm volatile("s_nop 9"); asm volatile("s_nop 9"); asm volatile("s_nop 9"); }
    }
    f32x4 acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (MODE == 5) {   // memory operands as in mode 0, first pass peeled, then the workgroup re-aligned by a barrier
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(i + c) % 16], bv[(i * 3 + c) % 16], acc[c], 0, 0, 0);
        __syncthreads();
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(i + c) % 16], bv[(i * 3 + c) % 16], acc[c], 0, 0, 0);
    }
    float r = 0.f;
    for (int c = 0; c < 8; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(int threads, float *out, const float *in)
{
    const int iters = 64, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_regs<MODE><<<blocks, threads>>>(out, in, 2);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k_regs<MODE><<<blocks, threads>>>(out, in, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double nmfma = (double)blocks * (threads / 64) * iters * 128.0;
    printf("mode %d waves/SIMD %d: %.3f ms  %.1f cycles/MFMA/SIMD @2.4GHz\n", MODE, threads / 256, ms, (ms * 1e-3) * 2.4e9 / (nmfma / 1024.0)); fflush(stdout);
}
int main()
{
    float *out, *in; (void)hipMalloc(&out, 256 * 1024 * sizeof(float)); (void)hipMalloc(&in, 64 * 64 * sizeof(float));
    (void)hipMemset(in, 0, 64 * 64 * sizeof(float));
    run<0>(256, out, in); run<0>(512, out, in); run<0>(1024, out, in);
    run<1>(256, out, in); run<1>(512, out, in); run<1>(1024, out, in);
    run<2>(256, out, in); run<2>(512, out, in); run<2>(1024, out, in);
    run<3>(256, out, in); run<3>(512, out, in); run<3>(1024, out, in);
    run<4>(256, out, in); run<4>(512, out, in); run<4>(1024, out, in);
    run<5>(256, out, in); run<5>(512, out, in); run<5>(1024, out, in);
    run<6>(256, out, in); run<6>(512, out, in); run<6>(1024, out, in);
    run<7>(256, out, in); run<7>(512, out, in); run<7>(1024, out, in);
    return 0;
}
