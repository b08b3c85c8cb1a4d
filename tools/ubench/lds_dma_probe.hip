// LDS-DMA on gfx950: (1) semantics probe -- does `global_load_lds_dwordx4` write lane l's 16 bytes at M0 + 16 l for an M0 ABOVE 64 KiB
// (the conv kernels' ring is 126 KiB)? -- and (2) the stream rate of one workgroup per CU pulling a large fp32 tensor through
//   A: global_load_dwordx4 -> VGPR -> ds_write_b128   (the register transport of the round-5 conv kernels)
//   B: global_load_lds_dwordx4 nt                     (LDS-DMA, nothing through the registers)
// with the same request pattern as k_conv2_wgrad_split (2 KiB chunks 62 KiB apart, 36 KiB per step, 8 staging waves, a barrier per
// step) and NO compute, i.e. the transport alone.    hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_dma_probe tools/ubench/lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// (1) one workgroup: every wave DMAs 1 KiB to LDS offset `base + 1024 * wave`, then the LDS content is copied out
__global__ void k_probe(const uint32_t *src, uint32_t *out, uint32_t base)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 150 * 1024 / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    glds16(src + wv * 256 + lane * 4, __builtin_amdgcn_readfirstlane(a + base + 1024 * wv));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 150 * 1024 / 4; i += blockDim.x) out[i] = reinterpret_cast<uint32_t *>(lds)[i];
}

constexpr int kRow = 2048, kPlanes = 9, kRows = 31, kSteps = 16;

// (2A) register transport: 8 waves, 5 x 16 B per lane and step, two steps ahead, stored to a 90 KiB ring
__global__ __launch_bounds__(512) void k_stream_regs(const float *y1, int nsamples, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, half = tid >> 7;  // region rs = 4 k + half
    float acc = 0.f;
    for (int s = blockIdx.x; s < nsamples * 4; s += gridDim.x) {
        const int b = s >> 2, grp = s & 3;
        const float *base = y1 + ((size_t)b * kRows + 8 * grp) * kRows * 512 + (tid & 127) * 4;
        float4 ra[5], rb[5];
        auto load = [&](float4 (&r)[5], int j) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int rs = 4 * k + half, pi = min(rs >> 1, 8), row = min(max(2 * j + 1 + (rs & 1), 0), kRows - 1);
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(base + ((size_t)pi * kRows + row) * 512));
                r[k] = make_float4(v[0], v[1], v[2], v[3]);
            }
        };
        auto store = [&](const float4 (&r)[5], int j) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int rs = 4 * k + half;
                if (rs >= 18) continue;
                *reinterpret_cast<float4 *>(lds + ((rs >> 1) * 5 + (2 * j + 1 + (rs & 1) + 5) % 5) * kRow + (tid & 127) * 16) = r[k];
            }
        };
        load(ra, 0);
        load(rb, 1);
        for (int t = 0; t < kSteps; t += 2) {
            store(ra, t);
            load(ra, t + 2);
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            store(rb, t + 1);
            load(rb, t + 3);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
        acc += ra[0].x + rb[0].x;
    }
    acc += reinterpret_cast<float *>(lds)[tid];
    if (acc == 123.456f) sink[0] = acc;
}

// (2B) LDS-DMA transport: 8 waves, 5 x 1 KiB per wave and step (the last one of waves 4-7 a duplicate), one step ahead, 126 KiB ring
__global__ __launch_bounds__(512) void k_stream_dma(const float *y1, int nsamples, float *sink, int ahead)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, pw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    float acc = 0.f;
    for (int s = blockIdx.x; s < nsamples * 4; s += gridDim.x) {
        const int b = s >> 2, grp = s & 3;
        const float *base = y1 + ((size_t)b * kRows + 8 * grp) * kRows * 512 + lane * 4;
        auto issue = [&](int j) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int h = min(pw + 8 * k, 35), pi = h >> 2, row = min(max(2 * j + 1 + ((h >> 1) & 1), 0), kRows - 1);
                glds16(base + ((size_t)pi * kRows + row) * 512 + (h & 1) * 256,
                       __builtin_amdgcn_readfirstlane(a + (uint32_t)((pi * 7 + (2 * j + 1 + ((h >> 1) & 1) + 14) % 7) * kRow + (h & 1) * 1024)));
            }
        };
        issue(0);
        if (ahead == 2) issue(1);
        for (int t = 0; t < kSteps; ++t) {
            issue(t + ahead);
            if (ahead == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            acc += *reinterpret_cast<const float *>(lds + ((pw >> 2) * 7 + (2 * t + 1 + 14) % 7) * kRow + lane * 4);  // (touch what landed)
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main()
{
    // ---- (1) semantics ----
    std::vector<uint32_t> h(16 * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x10000000u + (uint32_t)i;
    uint32_t *dsrc, *dout;
    CK(hipMalloc(&dsrc, h.size() * 4));
    CK(hipMalloc(&dout, 150 * 1024));
    CK(hipMemcpy(dsrc, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    std::vector<uint32_t> o(150 * 1024 / 4);
    int ok_all = 1;
    for (uint32_t base : {0u, 60u * 1024, 64u * 1024, 100u * 1024, 128u * 1024}) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(1024), 150 * 1024, 0, dsrc, dout, base);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), dout, 150 * 1024, hipMemcpyDeviceToHost));
        size_t bad = 0, stray = 0;
        for (size_t i = 0; i < o.size(); ++i) {
            const bool inside = i >= base / 4 && i < base / 4 + 16 * 256;
            if (inside) bad += o[i] != h[i - base / 4];
            else stray += o[i] != 0xdeadbeefu;
        }
        printf("probe: M0 base %6u: %zu wrong words inside the 16 KiB target, %zu words changed outside\n", base, bad, stray);
        ok_all &= bad == 0 && stray == 0;
    }
    printf("probe: %s\n", ok_all ? "LDS-DMA writes lane l at M0 + 16 l for every base tried (incl. > 64 KiB)" : "MISMATCH");
    // ---- (2) transport rate ----
    const int nsamples = 128;
    const size_t n = (size_t)nsamples * 31 * 31 * 512;
    float *y1, *sink;
    CK(hipMalloc(&y1, n * 4 + 4096));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(y1, 0, n * 4 + 4096));
    CK(hipFuncSetAttribute((const void *)k_stream_regs, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute((const void *)k_stream_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 135 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double bytes = (double)nsamples * 4 * kSteps * 36 * 1024;  // what the staging waves request
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            for (int it = 0; it < 22; ++it) {
                if (it == 2) CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_stream_regs, dim3(512), dim3(512), 100 * 1024, 0, y1, nsamples, sink);
                else hipLaunchKernelGGL(k_stream_dma, dim3(512), dim3(512), 135 * 1024, 0, y1, nsamples, sink, mode);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / 20;
            printf("stream: %-34s %7.1f us per launch, %5.2f TB/s requested, %5.2f B/clk/CU at 2.4 GHz\n",
                   mode == 0 ? "registers + ds_write (2 ahead)" : mode == 1 ? "LDS-DMA nt (1 step ahead)" : "LDS-DMA nt (2 steps ahead)", us,
                   bytes / us / 1e6, bytes / (us * 1e-6) / 256 / 2.4e9);
        }
    return ok_all ? 0 : 1;
}
