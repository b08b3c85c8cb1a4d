#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(1024) void k_regs(float *out, const float *in, int iters)
{
    const int lane = threadIdx.x & 63;
    float av[16], bv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { av[i] = in[i * 64 + lane]; bv[i] = in[(16 + i) * 64 + lane]; }
    f32x4 acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
