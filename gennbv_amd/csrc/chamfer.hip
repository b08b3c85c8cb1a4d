// Chamfer distance between two point clouds: the reconstruction-accuracy metric of the evaluation env
// (gennbv/env/env_eval_gennbv.py:253-262 calls pytorch3d.loss.chamfer_distance(x[None], y[None]) with its
// defaults; pytorch3d 0.7.8, the version the reference's README names: point_reduction = "mean",
// batch_reduction = "mean", norm = 2, i.e. SQUARED nearest-neighbour distances):
//     cd(x, y) = mean_i min_j |x_i - y_j|^2 + mean_j min_i |x_i - y_j|^2
// pytorch3d is not in this image (third-party, absent): the kernel restates the published definition; the
// oracle is a float64 brute force of the same formula (oracle/oracle.py::chamfer_distance_ref).
//
// Exact brute-force 1-NN in difference form ((a-b)^2 summed, no |a|^2+|b|^2-2ab expansion: at centimetre
// distances between points metres from the origin the expansion loses every significant digit in fp32).
// One thread owns kPtsPerThread query points; the other cloud streams through LDS in tiles of 1024 points
// that every lane reads at the same address (broadcast, conflict-free).  VALU-bound: 7 ops per pair.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

namespace {

constexpr int kNNThreads = 256, kPtsPerThread = 4, kTile = 1024;

__global__ __launch_bounds__(kNNThreads) void k_nn_sqdist(const float *__restrict__ q, int nq, const float *__restrict__ r, int nr,
                                                        float *__restrict__ out /*[nq]*/)
{
    __shared__ float4 tile[kTile];
    float qx[kPtsPerThread], qy[kPtsPerThread], qz[kPtsPerThread], best[kPtsPerThread];
    const int base = (blockIdx.x * kNNThreads + threadIdx.x) * kPtsPerThread;
#pragma unroll
    for (int p = 0; p < kPtsPerThread; ++p) {
        const int i = min(base + p, nq - 1);
        qx[p] = q[3 * (size_t)i]; qy[p] = q[3 * (size_t)i + 1]; qz[p] = q[3 * (size_t)i + 2];
        best[p] = FLT_MAX;
    }
    for (int t0 = 0; t0 < nr; t0 += kTile) {
        const int cnt = min(kTile, nr - t0);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += kNNThreads) {
            const float *s = r + 3 * (size_t)(t0 + j);
            tile[j] = make_float4(s[0], s[1], s[2], 0.0f);
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < cnt; ++j) {
            const float4 y = tile[j];
#pragma unroll
            for (int p = 0; p < kPtsPerThread; ++p) {
                const float dx = qx[p] - y.x, dy = qy[p] - y.y, dz = qz[p] - y.z;
                best[p] = fminf(best[p], __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx)));
            }
        }
    }
#pragma unroll
    for (int p = 0; p < kPtsPerThread; ++p)
        if (base + p < nq) out[base + p] = best[p];
}

// fixed-order fp64 sums: stage 1 = one partial per workgroup, stage 2 = one workgroup over the partials
__global__ __launch_bounds__(256) void k_sum_f64(const float *__restrict__ v, int64_t n, double *__restrict__ partial)
{
    __shared__ double s[256];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += (double)v[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(256) void k_chamfer_finish(const double *__restrict__ px, int bx, double nx, const double *__restrict__ py, int by,
                                                      double ny, float *__restrict__ out)
{
    __shared__ double s[2][256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < bx; i += 256) a += px[i];
    for (int i = threadIdx.x; i < by; i += 256) b += py[i];
    s[0][threadIdx.x] = a;
    s[1][threadIdx.x] = b;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            s[0][threadIdx.x] += s[0][threadIdx.x + d];
            s[1][threadIdx.x] += s[1][threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(s[0][0] / nx + s[1][0] / ny);
}

constexpr int kSumBlocks = 512;

}  // namespace

GNBV_API size_t gnbv_chamfer_workspace_bytes(int n, int m)
{
    if (n <= 0 || m <= 0) return 0;
    return ((size_t)n + (size_t)m) * sizeof(float) + 2 * kSumBlocks * sizeof(double) + 512;
}

GNBV_API int gnbv_chamfer_distance(const float *x, int n, const float *y, int m, float *out, void *workspace, size_t workspace_bytes,
                                   void *stream)
{
    GNBV_CHECK_ARG(x && y && out && workspace && n > 0 && m > 0);
    GNBV_CHECK_ARG(workspace_bytes >= gnbv_chamfer_workspace_bytes(n, m) && ((uintptr_t)workspace & 15) == 0);
    hipStream_t st = gnbv_stream(stream);
    float *dx = (float *)workspace, *dy = dx + n;
    double *px = (double *)(((uintptr_t)(dy + m) + 255) & ~(uintptr_t)255), *py = px + kSumBlocks;
    const int per = kNNThreads * kPtsPerThread;
    hipLaunchKernelGGL(k_nn_sqdist, dim3((n + per - 1) / per), dim3(kNNThreads), 0, st, x, n, y, m, dx);
    hipLaunchKernelGGL(k_nn_sqdist, dim3((m + per - 1) / per), dim3(kNNThreads), 0, st, y, m, x, n, dy);
    int err;
    if ((err = gnbv_launch_status())) return err;
    hipLaunchKernelGGL(k_sum_f64, dim3(kSumBlocks), dim3(256), 0, st, (const float *)dx, (int64_t)n, px);
    hipLaunchKernelGGL(k_sum_f64, dim3(kSumBlocks), dim3(256), 0, st, (const float *)dy, (int64_t)m, py);
    hipLaunchKernelGGL(k_chamfer_finish, dim3(1), dim3(256), 0, st, (const double *)px, kSumBlocks, (double)n, (const double *)py, kSumBlocks,
                       (double)m, out);
    return gnbv_launch_status();
}
