// Shared helpers for the gfx950 kernels (device + host).  CDNA4 only: wave = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#define GNBV_API extern "C" __attribute__((visibility("default")))

#define GNBV_CHECK_ARG(cond) \
    do {                     \
        if (!(cond)) return (int)hipErrorInvalidValue; \
    } while (0)

static inline int gnbv_launch_status() { return (int)hipGetLastError(); }

static inline hipStream_t gnbv_stream(void *s) { return (hipStream_t)s; }

constexpr int kWave = 64;

// ---- wave / block primitives ------------------------------------------------
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        int o = __shfl_up(v, d, kWave);
        if (lane >= d) v += o;
    }
    return v;
}

__device__ __forceinline__ int wave_reduce_sum(int v)
{
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_down(v, d, kWave);
    return v;
}

__device__ __forceinline__ float wave_reduce_sum(float v)
{
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_down(v, d, kWave);
    return v;
}
