// Micro-benchmark (see profiles/r01_notes.md): rate of wave-wide 16-byte-per-lane global loads as a function of the lane -> address
// mapping, data L2-resident.  (a) lane l reads 16 B at l * 16 (fully contiguous 1 KiB); (b) the conv kernels' MFMA-operand mapping:
// lane = 16 kq + m reads voxel m (64 B apart), quad kq (16 B inside the voxel) -- the same 1 KiB, but the four lanes of a hardware quad
// touch four different 64-byte segments; (c) dword loads, lanes 4 B apart; (d) dword loads, lanes 64 B apart.
//   hipcc --offload-arch=gfx950 -O3 -o vmem_quad_rate tools/ubench/vmem_quad_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k_load(const float *__restrict__ src, float *__restrict__ out, int iters)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    // each wave walks its own 64 KiB window (L2 / L1 resident after the first pass)
    const float *base = src + ((size_t)(blockIdx.x % 64) * 4 + wv) * 16384;
    int off;
    if (MODE == 0) off = lane * 4;
    else if (MODE == 1) off = m * 16 + kq * 4;
    else if (MODE == 2) off = lane;
    else off = lane * 16;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float accs = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int step = (MODE < 2 ? 256 : (MODE == 2 ? 64 : 1024)) * u;   // next 1 KiB (or 256 B / 4 KiB) block
            if (MODE < 2) {
                const float4 v = *reinterpret_cast<const float4 *>(base + off + step);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            } else {
                accs += base[off + (step & 16383)];
            }
        }
        off += (int)(acc.x + accs);  // the source is all zeros: keeps the loads inside the loop without changing the addresses
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w + accs;
}

template <int MODE> void run(const char *name, const float *src, float *out)
{
    const int iters = 256, blocks = 256 * 2;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_load<MODE><<<blocks, 256>>>(src, out, 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k_load<MODE><<<blocks, 256>>>(src, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ninst = (double)blocks * 4 * iters * 16;   // wave-level load instructions
    printf("%-44s %.3f ms  %.1f cycles per wave-load per CU @2.1GHz  (%.1f B/clk/CU)\n", name, ms, ms * 1e-3 * 2.1e9 / (ninst / 256.0),
           (MODE < 2 ? 1024.0 : 256.0) * (ninst / 256.0) / (ms * 1e-3 * 2.1e9));
}
int main()
{
    float *src, *out; (void)hipMalloc(&src, 64 * 4 * 16384 * sizeof(float) + 65536); (void)hipMalloc(&out, 512 * 256 * sizeof(float));
    (void)hipMemset(src, 0, 64 * 4 * 16384 * sizeof(float) + 65536);
    run<0>("dwordx4, lanes 16 B apart (contiguous)", src, out);
    run<1>("dwordx4, lane = 16kq+m: m 64 B apart, kq 16 B", src, out);
    run<2>("dword, lanes 4 B apart", src, out);
    run<3>("dword, lanes 64 B apart", src, out);
    return 0;
}
