// Micro-benchmark: sustained rate of v_mfma_f32_16x16x4_f32 on gfx950 as a function of
// waves per SIMD and of the number of independent accumulator chains per wave.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/ubench/mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, bool LDSB, int NVALU = 0>
__global__ __launch_bounds__(1024) void k(float *out, int iters)
{
    __shared__ float w[27 * 256];
    for (int i = threadIdx.x; i < 27 * 256; i += blockDim.x) w[i] = 1e-3f * (i & 63);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + lane * 1e-6f, b = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 27; ++t) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float bb = LDSB ? w[t * 256 + s * 64 + lane] : b;
                // NVALU independent-of-the-chain VALU ops feeding the A operand (the BN + ReLU of conv2)
#pragma unroll
                for (int u = 0; u < NVALU; ++u) a = fmaxf(fmaf(a, 0.999f, b), 0.25f);
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc[c], 0, 0, 0);
            }
        }
    }
    float r = 0.f;
    for (int c = 0; c < CHAINS; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int CHAINS, bool LDSB, int NVALU = 0>
void run(int threads, int blocks_per_cu, float *out)
{
    const int iters = 64, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<CHAINS, LDSB, NVALU><<<blocks, threads>>>(out, 2);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<CHAINS, LDSB, NVALU><<<blocks, threads>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double nmfma = (double)blocks * (threads / 64) * iters * 108.0 * CHAINS;
    const double tf = nmfma * 2048.0 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD at 2.4 GHz: 1024 SIMDs
    const double cyc = (ms * 1e-3) * 2.4e9 / (nmfma / 1024.0);
    printf("valu-pairs %d chains %d ldsB %d threads %4d blocks/CU %d: %.3f ms  %.1f TFLOP/s  %.1f cycles/MFMA/SIMD @2.4GHz\n", NVALU, CHAINS, (int)LDSB, threads,
           blocks_per_cu, ms, tf, cyc);
}

// Operands from DISTINCT registers (the split-K / conv kernels' situation), 8 accumulator chains, optionally a long run
// (sustained clock) -- does the pipe still take one MFMA per ~32 cycles?
template <int NREG>
__global__ __launch_bounds__(1024) void k_regs(float *out, const float *in, int iters)
{
    const int lane = threadIdx.x & 63;
    float av[NREG], bv[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) { av[i] = in[i * 64 + lane]; bv[i] = in[(NREG + i) * 64 + lane]; }
    f32x4 acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NREG; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(i + c) % NREG], bv[(i * 3 + c) % NREG], acc[c], 0, 0, 0);
    }
    float r = 0.f;
    for (int c = 0; c < 8; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// Same, but the operands are RE-LOADED every pass (register double buffer: NL 16-byte requests per lane in flight while the
// 8 * NREG MFMAs of the previous pass run; source = a cache-resident buffer) -- the conv kernels' steady state.
// UPFRONT: one s_waitcnt for the whole operand set in front of each MFMA block instead of the compiler's descending vmcnt(N) between the MFMAs.
constexpr int vmcnt_imm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }
template <int NREG, int NL, bool UPFRONT = false>
__global__ __launch_bounds__(1024) void k_regs_ld(float *out, const float *in, int iters)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float4 *src = reinterpret_cast<const float4 *>(in) + (size_t)(blockIdx.x % 64) * 4096 + wv * 64 + lane;
    float4 va[NL], vb[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) va[i] = src[i * 1024];
    f32x4 acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto consume = [&](const float4 (&v)[NL]) {
        const float *f = reinterpret_cast<const float *>(v);
#pragma unroll
        for (int i = 0; i < NREG; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[(i + c) % (4 * NL)], f[(i * 3 + c + 1) % (4 * NL)], acc[c], 0, 0, 0);
    };
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int i = 0; i < NL; ++i) vb[i] = src[i * 1024 + ((it + 1) & 3) * 256];
        __builtin_amdgcn_sched_barrier(0);
        if (UPFRONT) { __builtin_amdgcn_s_waitcnt(vmcnt_imm(NL)); __builtin_amdgcn_sched_barrier(0); }
        consume(va);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NL; ++i) va[i] = src[i * 1024 + ((it + 2) & 3) * 256];
        __builtin_amdgcn_sched_barrier(0);
        if (UPFRONT) { __builtin_amdgcn_s_waitcnt(vmcnt_imm(NL)); __builtin_amdgcn_sched_barrier(0); }
        consume(vb);
        __builtin_amdgcn_sched_barrier(0);
    }
    float r = 0.f;
    for (int c = 0; c < 8; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NREG, int NL, bool UPFRONT = false>
void run_regs_ld(int threads, int blocks_per_cu, int iters, float *out, const float *in)
{
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_regs_ld<NREG, NL, UPFRONT><<<blocks, threads>>>(out, in, 2);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_regs_ld<NREG, NL, UPFRONT><<<blocks, threads>>>(out, in, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double nmfma = (double)blocks * (threads / 64) * iters * 8.0 * NREG;
    printf("%soperands re-loaded each pass: %2d x 16 B per lane per %3d MFMAs, threads %4d blocks/CU %d: %.3f ms  %.1f TFLOP/s  %.1f cycles/MFMA/SIMD @2.4GHz\n", UPFRONT ? "[one up-front wait] " : "", NL,
           8 * NREG, threads, blocks_per_cu, ms, nmfma * 2048.0 / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 / (nmfma / 1024.0));
}

template <int NREG>
void run_regs(int threads, int blocks_per_cu, int iters, float *out, const float *in)
{
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_regs<NREG><<<blocks, threads>>>(out, in, 2);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_regs<NREG><<<blocks, threads>>>(out, in, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double nmfma = (double)blocks * (threads / 64) * iters * 8.0 * NREG;
    printf("distinct operand registers %2d, 8 chains, threads %4d blocks/CU %d, %8.0f MFMA/wave: %.3f ms  %.1f TFLOP/s  %.1f cycles/MFMA/SIMD @2.4GHz\n", NREG,
           threads, blocks_per_cu, (double)iters * 8 * NREG, ms, nmfma * 2048.0 / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 / (nmfma / 1024.0));
}

int main()
{
    {
        float *out, *in; (void)hipMalloc(&out, 256 * 8 * 1024 * sizeof(float)); (void)hipMalloc(&in, 64 * 64 * sizeof(float));
        (void)hipMemset(in, 0, 64 * 64 * sizeof(float));
        run_regs<16>(256, 1, 64, out, in);      // 1 wave / SIMD, short
        run_regs<16>(512, 1, 64, out, in);      // 2 waves / SIMD
        run_regs<16>(512, 1, 4096, out, in);    // 2 waves / SIMD, ~50 ms: sustained clock
        run_regs<16>(1024, 1, 4096, out, in);   // 4 waves / SIMD, long
        float *big; (void)hipMalloc(&big, (size_t)64 * 4096 * 16 + 65536 * 16); (void)hipMemset(big, 0, (size_t)64 * 4096 * 16 + 65536 * 16);
        run_regs_ld<16, 4>(256, 1, 512, out, big);    // 1 wave / SIMD, 4 KiB per wave per 128 MFMAs
        run_regs_ld<16, 16>(256, 1, 512, out, big);   // 16 KiB per wave per 128 MFMAs (conv2_fwd: 27 KiB per 108)
        run_regs_ld<4, 8>(256, 1, 2048, out, big);    // 8 KiB per wave per 32 MFMAs (conv2_fwd's ratio)
        run_regs_ld<16, 4>(1024, 1, 512, out, big);   // 4 waves / SIMD
        run_regs_ld<4, 8>(1024, 1, 2048, out, big);
        run_regs_ld<16, 16, true>(256, 1, 512, out, big);
        run_regs_ld<4, 8, true>(256, 1, 2048, out, big);
        run_regs_ld<16, 4, true>(1024, 1, 512, out, big);
        run_regs_ld<4, 8, true>(1024, 1, 2048, out, big);
        (void)hipFree(big);
        (void)hipFree(out); (void)hipFree(in);
    }
    float *out; (void)hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
    run<1, false>(256, 1, out);   // 1 wave / SIMD
    run<1, false>(512, 1, out);   // 2
    run<1, false>(1024, 1, out);  // 4
    run<1, false>(1024, 2, out);  // 8
    run<2, false>(256, 1, out);
    run<4, false>(256, 1, out);
    run<4, false>(1024, 1, out);
    run<1, true>(256, 1, out);
    run<1, true>(1024, 1, out);
    run<1, true>(1024, 2, out);
    run<2, true>(1024, 1, out);
    run<4, true>(1024, 1, out);
    run<1, true, 1>(256, 1, out);
    run<1, true, 1>(1024, 1, out);
    run<1, true, 1>(1024, 2, out);
    run<1, true, 2>(1024, 1, out);
    run<1, true, 4>(1024, 1, out);
    run<2, true, 1>(1024, 1, out);
    return 0;
}
