// Split-K fp32 MFMA kernel for the one skinny, K-dominated linear of the encoder:
//   Hybrid_Encoder.output_layer_grid = Linear(16*o2^3, 256) + ReLU   (gennbv/network/hybrid_encoder.py:39-42, :87)
// At G = 64 the layer is [M = minibatch 128] x [K = 54 000] x [N = 256]: 55 MB of weights, 3.5 GFLOP.
// A library GEMM tiles M x N (only 8 x 16 MFMA tiles) and leaves most CUs idle or runs a generic
// split-K with a slow epilogue (119 us measured for the TunableOp-selected rocBLAS/hipBLASLt kernel);
// here every CU streams its own K-chunk of W exactly once:
//   stage 1  k_linear_splitk : workgroup = (N-tile of 64 columns, K-chunk), 4 waves x (2 row-tiles x
//            4 column-tiles) accumulators, operands loaded 16 bytes per lane ALONG K (both x and W are
//            K-contiguous), register double buffer over 32-k steps  -> partial [chunk][M][N]
//   stage 2  k_linear_reduce : out = relu(bias + sum_chunks partial), chunks in ascending order
// Deterministic (no atomics).  MFMA operand roles (v_mfma_f32_16x16x4_f32: A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15]): lane (i, kq) holds x[row i][k0 + 4 kq + s], s = 0..3, and the matching
// W[col i][k0 + 4 kq + s]; MFMA s contracts k in {4 kq + s}: the four MFMAs of a 16-k step cover all 16.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kLinThreads = 256;  // 4 waves
constexpr int kTileN = 64;        // columns per workgroup
constexpr int kRowTilesPerWave = 2;

__device__ __forceinline__ float4 ld4g(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// blockIdx.y = slab of 128 rows (8 row-tiles, 2 per wave); rows >= M are clamped for the loads and never stored.
//
// The W tile of a trip (64 columns x 32 k = 8 KiB) is the SAME for the four waves of the workgroup: it is fetched once
// (two 16-byte requests per lane), staged in LDS (double-buffered, one barrier per trip) and read back as MFMA operands with
// ds_read_b128; only the x rows of a wave come straight from global memory.  With every wave loading W itself the workgroup
// issued 48 vector-memory instructions per 64-MFMA trip = 47 B/clk per CU of the 64 B/clk the CU's address path moves: the
// kernel ran at 46 % of the matrix pipe's rate (profiles/r02_notes.md).
__global__ __launch_bounds__(kLinThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_linear_splitk(
    const float *__restrict__ x /*[M][K]*/, const float *__restrict__ w /*[N][K]*/, int M, int N, int K, int nchunks,
    float *__restrict__ partial /*[nchunks][M][N]*/)
{
    __shared__ float4 s_b[2][2][4][4][17];  // [buffer][h: 16-k step of the trip][column tile][kq][column i (+1 pad: the staging stores of 8 lanes differ in kq)]: 2 x 8.5 KiB
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int i = lane & 15, kq = lane >> 4, mbase = blockIdx.y * 128;
    // block -> (chunk, N-tile): the N-tiles of one chunk sit on the same XCD (block b runs on XCD b % 8)
    // so that the x chunk they share is fetched from HBM once per XCD
    const int ntn = N / kTileN;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int nt = slot % ntn, chunk = (slot / ntn) * 8 + xcd;
    if (chunk >= nchunks) return;
    const int nstep16 = (K + 15) / 16;  // 16-k steps
    const int s0 = (int)((int64_t)chunk * nstep16 / nchunks), s1 = (int)((int64_t)(chunk + 1) * nstep16 / nchunks);
    const float *xr[kRowTilesPerWave];
#pragma unroll
    for (int rt = 0; rt < kRowTilesPerWave; ++rt) xr[rt] = x + (size_t)min(mbase + (wv * kRowTilesPerWave + rt) * 16 + i, M - 1) * K + 4 * kq;
    // W staging: thread t fetches, for r = 0, 1, the float4 W[col = 32 r + t / 8][k = trip k0 + 4 (t % 8) .. + 3]
    // (8 threads = 128 contiguous bytes of one column) and stores it at s_b[buf][h = (t % 8) / 4][ct = col / 16][kq = t % 4][col % 16]
    const int t8 = threadIdx.x & 7, tcol = threadIdx.x >> 3;
    const float *wst[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) wst[r] = w + (size_t)(nt * kTileN + 32 * r + tcol) * K + 4 * t8;
    f32x4 acc[kRowTilesPerWave][4];
#pragma unroll
    for (int rt = 0; rt < kRowTilesPerWave; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // Requests are UNCONDITIONAL (clamped addresses past the chunk end): a branch around a request group makes the compiler's
    // s_waitcnt insertion assume the worst case at the join and wait for the prefetched group as well.
    auto request_a = [&](int s, float4 (&a)[2][kRowTilesPerWave]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kc = min(s + h, s1 - 1) * 16;
#pragma unroll
            for (int rt = 0; rt < kRowTilesPerWave; ++rt) a[h][rt] = ld4g(xr[rt] + min(kc, K - 4 - 4 * kq));
        }
    };
    auto request_b = [&](int s, float4 (&b)[2]) {
        const int kc = min(s + (t8 >> 2), s1 - 1) * 16 - 16 * (t8 >> 2);  // this thread's 16-k step of the trip, clamped to the chunk
#pragma unroll
        for (int r = 0; r < 2; ++r) b[r] = ld4g(wst[r] + min(kc, K - 4 - 4 * t8));
    };
    auto stage_b = [&](int buf, const float4 (&b)[2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r) s_b[buf][t8 >> 2][2 * r + (tcol >> 4)][t8 & 3][tcol & 15] = b[r];
    };
    auto consume = [&](int s, int buf, float4 (&a)[2][kRowTilesPerWave]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool ok = (s + h) < s1 && (s + h) * 16 + 4 * kq + 3 < K;  // past the chunk / past K: zeros
            float4 b[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) b[ct] = s_b[buf][h][ct][kq][i];
#pragma unroll
            for (int rt = 0; rt < kRowTilesPerWave; ++rt) {
                if (!ok) a[h][rt] = zero4;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    acc[rt][ct] = mfma4(a[h][rt].x, b[ct].x, acc[rt][ct]);
                    acc[rt][ct] = mfma4(a[h][rt].y, b[ct].y, acc[rt][ct]);
                    acc[rt][ct] = mfma4(a[h][rt].z, b[ct].z, acc[rt][ct]);
                    acc[rt][ct] = mfma4(a[h][rt].w, b[ct].w, acc[rt][ct]);
                }
            }
        }
    };
    if (s0 >= s1) return;  // (never: nchunks <= number of steps)
    float4 a0[2][kRowTilesPerWave], a1[2][kRowTilesPerWave], bq[2];
    request_a(s0, a0);
    request_b(s0, bq);
    stage_b(0, bq);
    __syncthreads();
    int buf = 0;
    for (int s = s0; s < s1; s += 4) {
        // trip s (a0, LDS buffer `buf`) while trip s + 2's operands are in flight; then trip s + 2 while s + 4's are
        request_a(s + 2, a1);
        request_b(s + 2, bq);
        __builtin_amdgcn_sched_barrier(0);
        consume(s, buf, a0);
        __builtin_amdgcn_sched_barrier(0);
        stage_b(buf ^ 1, bq);
        __syncthreads();
        request_a(s + 4, a0);
        request_b(s + 4, bq);
        __builtin_amdgcn_sched_barrier(0);
        consume(s + 2, buf ^ 1, a1);
        __builtin_amdgcn_sched_barrier(0);
        stage_b(buf, bq);
        __syncthreads();
    }
    // D[i = 4 kq + r][j = lane & 15]: row = tile row 4 kq + r, column = tile column i
    float *out = partial + (size_t)chunk * M * N;
#pragma unroll
    for (int rt = 0; rt < kRowTilesPerWave; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = mbase + (wv * kRowTilesPerWave + rt) * 16 + 4 * kq + r;
            if (row < M) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) out[(size_t)row * N + nt * kTileN + ct * 16 + i] = acc[rt][ct][r];
            }
        }
}

// out[m][n] = act(bias[n] + sum_c partial[c][m][n]); one thread per 4 consecutive columns, the chunk
// loop is split over `kRedSplit` threads whose sub-sums are combined in a fixed order through LDS.
constexpr int kRedSplit = 4;
__global__ __launch_bounds__(256) void k_linear_reduce(const float *__restrict__ partial, const float *__restrict__ bias, int M, int N,
                                                       int nchunks, int relu, float *__restrict__ out)
{
    __shared__ float4 sub[kRedSplit][64];
    const int q = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int64_t e4 = (int64_t)blockIdx.x * 64 + q, total4 = (int64_t)M * N / 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e4 < total4) {
        const int per = (nchunks + kRedSplit - 1) / kRedSplit, c0 = part * per, c1 = min(nchunks, c0 + per);
        for (int c = c0; c < c1; ++c) {
            const float4 v = ld4g(partial + ((size_t)c * M * N) + e4 * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sub[part][q] = s;
    __syncthreads();
    if (part == 0 && e4 < total4) {
        float4 t = sub[0][q];
#pragma unroll
        for (int p = 1; p < kRedSplit; ++p) {
            t.x += sub[p][q].x; t.y += sub[p][q].y; t.z += sub[p][q].z; t.w += sub[p][q].w;
        }
        const int n = (int)((e4 * 4) % N);
        const float4 b = ld4g(bias + n);
        t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
        if (relu) {
            t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
        }
        *reinterpret_cast<float4 *>(out + e4 * 4) = t;
    }
}

inline int pick_chunks(int K)
{
    const int nstep16 = (K + 15) / 16;
    int c = 128;  // 2 workgroups of 4 waves per CU at N = 256
    while (c > 1 && nstep16 / c < 8) c >>= 1;  // keep >= 8 steps per chunk
    return c;
}

}  // namespace

GNBV_API size_t gnbv_linear_workspace_bytes(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (size_t)pick_chunks(K) * M * N * sizeof(float) + 256;
}

GNBV_API int gnbv_linear_forward(const float *x, const float *w, const float *bias, int M, int N, int K, int relu, float *out,
                                 void *workspace, size_t workspace_bytes, void *stream)
{
    GNBV_CHECK_ARG(x && w && bias && out && workspace);
    GNBV_CHECK_ARG(M > 0 && N > 0 && N % kTileN == 0 && K > 0 && K % 4 == 0);
    GNBV_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)out | (uintptr_t)workspace) & 15) == 0);
    GNBV_CHECK_ARG(workspace_bytes >= gnbv_linear_workspace_bytes(M, N, K));
    hipStream_t st = gnbv_stream(stream);
    const int nchunks = pick_chunks(K), ntn = N / kTileN;
    const int blocks = ((nchunks + 7) / 8) * 8 * ntn;
    hipLaunchKernelGGL(k_linear_splitk, dim3(blocks, (M + 127) / 128), dim3(kLinThreads), 0, st, x, w, M, N, K, nchunks, (float *)workspace);
    int err;
    if ((err = gnbv_launch_status())) return err;
    const int64_t total4 = (int64_t)M * N / 4;
    hipLaunchKernelGGL(k_linear_reduce, dim3((unsigned)((total4 + 63) / 64)), dim3(256), 0, st, (const float *)workspace, bias, M, N, nchunks,
                       relu, out);
    return gnbv_launch_status();
}
