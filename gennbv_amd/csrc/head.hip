// Policy head of the GenNBV actor-critic, fused (M = minibatch rows, F = 256 features):
//   feat   = relu([fa | fg] W_out^T + b_out)      Hybrid_Encoder.output_layer   (gennbv/network/hybrid_encoder.py:51-54, :89)
//   logits = feat W_act^T + b_act                 ActorCriticPolicy.action_net  (stable_baselines3/common/policies.py:975, :1024)
//   values = feat W_val^T + b_val                 ActorCriticPolicy.value_net   (policies.py:979, :1011)
// and its backward.  Through the library these are 6 launches forward (cat, GEMM, clamp, GEMM, add, GEMV)
// and ~16 backward, each a few microseconds of work behind a ~5 us dependent-launch floor inside the
// captured minibatch graph; here: 2 launches forward, 2 backward.  The GEMMs are tiny (<= 34 MFLOP),
// so the kernels are organised for latency, not bandwidth: one wave per 16x16 output tile, the whole
// contraction in that wave (deterministic, no atomics, no split-K), operands straight from L2.
//
// v_mfma_f32_16x16x4_f32 operand roles: lane (x = l & 15, kq = l >> 4) supplies A[i = x][k] and
// B[k][j = x] for k = k0 + 4 kq + s in MFMA s of a 16-k group (s = 0..3), and owns D[i = 4 kq + r][j = x].
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kHeadThreads = 256;  // 4 waves = 4 tiles per workgroup

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// ---- "NT" contraction: both operands k-contiguous, 16-byte requests along k ------------------------
// arow / brow point at (row i, k = 0) / (col j, k = 0) of this lane (already offset by 4 kq); K % 16 == 0.
// A may switch to a second segment (the concat of two inputs) at k = K1.
// U = 16-k groups whose requests are issued together: the contraction is a chain of dependent L2 round trips (~1 us each) with a
// handful of MFMAs in between, and these launches are a few dozen single-wave-per-SIMD workgroups -- registers are free.  U = 16
// (256 k per round trip, 128 VGPRs of operands) makes output_layer's K = 512 two round trips instead of eight (round 3; the
// order of the accumulation, k ascending, is unchanged: bit-identical results).
template <int U>
__device__ __forceinline__ f32x4 contract_nt(const float *arow, const float *arow2, int K1, const float *brow, int K, bool a_ok, bool b_ok)
{
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < K; k0 += 16 * U) {
        float4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(k0 + 16 * u, K - 16);
            a[u] = ld4(k < K1 ? arow + k : arow2 + (k - K1));
            b[u] = ld4(brow + k);
        }
        __builtin_amdgcn_sched_barrier(0);  // all 2 U requests of the step before its first MFMA
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = k0 + 16 * u < K;
            const float4 av = (a_ok && live) ? a[u] : z, bv = b_ok ? b[u] : z;
            acc = mfma4(av.x, bv.x, acc);
            acc = mfma4(av.y, bv.y, acc);
            acc = mfma4(av.z, bv.z, acc);
            acc = mfma4(av.w, bv.w, acc);
        }
    }
    return acc;
}
constexpr int kHeadU = 16;  // 16-k groups per round trip in the forward contractions
constexpr int kHeadUB = 4;  // 32-k (or 32-row) groups per round trip in the backward contractions (4-byte requests: 16 per group)

// ===========================================================================================
// forward 1: feat[M][F] = relu([fa | fg] W_out^T + b_out)
// ===========================================================================================
__global__ __launch_bounds__(kHeadThreads) void k_head_fwd_feat(const float *__restrict__ fa, const float *__restrict__ fg, int M, int K1, int K2,
                                                              const float *__restrict__ W_out, const float *__restrict__ b_out, int F,
                                                              float *__restrict__ feat)
{
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int x = lane & 15, kq = lane >> 4;
    const int tiles_j = F / 16, tile = blockIdx.x * (int)(blockDim.x / kWave) + wv, ti = tile / tiles_j, tj = tile % tiles_j;
    if (ti * 16 >= M) return;
    const int i = min(ti * 16 + x, M - 1), j = tj * 16 + x;
    const f32x4 acc = contract_nt<kHeadU>(fa + (size_t)i * K1 + 4 * kq, fg + (size_t)i * K2 + 4 * kq, K1, W_out + (size_t)j * (K1 + K2) + 4 * kq,
                                          K1 + K2, true, true);
    const float bias = b_out[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + 4 * kq + r;
        if (row < M) feat[(size_t)row * F + j] = fmaxf(acc[r] + bias, 0.0f);
    }
}

// ===========================================================================================
// forward 2: logits[M][A] = feat W_act^T + b_act ; values[M] = feat W_val^T + b_val  (column A of the tile grid)
// ===========================================================================================
__global__ __launch_bounds__(kHeadThreads) void k_head_fwd_out(const float *__restrict__ feat, int M, int F, const float *__restrict__ W_act,
                                                             const float *__restrict__ b_act, int A, const float *__restrict__ W_val,
                                                             const float *__restrict__ b_val, float *__restrict__ logits,
                                                             float *__restrict__ values)
{
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int x = lane & 15, kq = lane >> 4;
    const int tiles_j = (A + 1 + 15) / 16, tile = blockIdx.x * (int)(blockDim.x / kWave) + wv, ti = tile / tiles_j, tj = tile % tiles_j;
    if (ti * 16 >= M) return;
    const int i = min(ti * 16 + x, M - 1), j = tj * 16 + x;
    const float *brow = j < A ? W_act + (size_t)j * F : W_val;  // column A = the value head
    const f32x4 acc = contract_nt<kHeadU>(feat + (size_t)i * F + 4 * kq, feat, F, brow + 4 * kq, F, true, j <= A);
    const float bias = j < A ? b_act[j] : (j == A ? b_val[0] : 0.0f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + 4 * kq + r;
        if (row < M) {
            if (j < A) logits[(size_t)row * A + j] = acc[r] + bias;
            else if (j == A) values[row] = acc[r] + bias;
        }
    }
}

// ===========================================================================================
// backward 1: dH[M][F] = [feat > 0] * (d_logits W_act + d_values (x) W_val)
//   contraction over k = action index (A + 1 entries, the last one the value head); operands by 4-byte
//   requests (d_logits rows are A floats apart: not 16-byte aligned)
// ===========================================================================================
__global__ __launch_bounds__(kHeadThreads) void k_head_bwd_dh(const float *__restrict__ d_logits, const float *__restrict__ d_values, int M, int A,
                                                            const float *__restrict__ W_act, const float *__restrict__ W_val, int F,
                                                            const float *__restrict__ feat, float *__restrict__ dH)
{
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int x = lane & 15, kq = lane >> 4;
    const int tiles_j = F / 16, tile = blockIdx.x * (int)(blockDim.x / kWave) + wv, ti = tile / tiles_j, tj = tile % tiles_j;
    if (ti * 16 >= M) return;
    const int i = min(ti * 16 + x, M - 1), j = tj * 16 + x, K = A + 1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float dv = d_values[i], wv_j = W_val[j];
    for (int k0 = 0; k0 < K; k0 += 32 * kHeadUB) {  // (kHeadUB x 16 requests per round trip: A + 1 = 241 -> two instead of eight)
        float a[2 * kHeadUB][4], b[2 * kHeadUB][4];
#pragma unroll
        for (int u = 0; u < 2 * kHeadUB; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = k0 + 16 * u + 4 * kq + s, kc = min(k, A - 1);
                const float av = d_logits[(size_t)i * A + kc], bv = W_act[(size_t)kc * F + j];
                a[u][s] = k < A ? av : (k == A ? dv : 0.0f);
                b[u][s] = k < A ? bv : (k == A ? wv_j : 0.0f);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2 * kHeadUB; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma4(a[u][s], b[u][s], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + 4 * kq + r;
        if (row < M) dH[(size_t)row * F + j] = feat[(size_t)row * F + j] > 0.0f ? acc[r] : 0.0f;
    }
}

// ===========================================================================================
// backward 2 (one launch, three kinds of tiles):
//   kind 0: d[fa | fg][M][K1+K2] = dH W_out                       contraction over F
//   kind 1: gW_out[F][K1+K2] = dH^T [fa | fg], gb_out = colsum dH   contraction over the M rows
//   kind 2: gW_act[A][F] = d_logits^T feat, gb_act; gW_val = d_values^T feat, gb_val (row A of the tile grid)
// ===========================================================================================
__global__ __launch_bounds__(kHeadThreads) void k_head_bwd_rest(const float *__restrict__ fa, const float *__restrict__ fg, int M, int K1, int K2,
                                                              const float *__restrict__ feat, const float *__restrict__ dH, int F,
                                                              const float *__restrict__ d_logits, const float *__restrict__ d_values, int A,
                                                              const float *__restrict__ W_out, float *__restrict__ d_fa, float *__restrict__ d_fg,
                                                              float *__restrict__ gW_out, float *__restrict__ gb_out, float *__restrict__ gW_act,
                                                              float *__restrict__ gb_act, float *__restrict__ gW_val, float *__restrict__ gb_val,
                                                              int n0, int n1)
{
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int x = lane & 15, kq = lane >> 4, KC = K1 + K2;
    int tile = blockIdx.x * (int)(blockDim.x / kWave) + wv;
    if (tile < n0) {
        // ---- kind 0: rows = minibatch rows, columns = concat inputs ----
        const int tiles_j = KC / 16, ti = tile / tiles_j, tj = tile % tiles_j;
        const int i = min(ti * 16 + x, M - 1), j = tj * 16 + x;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < F; k0 += 32 * kHeadUB) {
            float4 a[2 * kHeadUB];
            float b[2 * kHeadUB][4];
#pragma unroll
            for (int u = 0; u < 2 * kHeadUB; ++u) {
                const int k = min(k0 + 16 * u, F - 16) + 4 * kq;
                a[u] = ld4(dH + (size_t)i * F + k);
#pragma unroll
                for (int s = 0; s < 4; ++s) b[u][s] = W_out[(size_t)(k + s) * KC + j];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2 * kHeadUB; ++u) {
                if (k0 + 16 * u >= F) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                acc = mfma4(a[u].x, b[u][0], acc);
                acc = mfma4(a[u].y, b[u][1], acc);
                acc = mfma4(a[u].z, b[u][2], acc);
                acc = mfma4(a[u].w, b[u][3], acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = ti * 16 + 4 * kq + r;
            if (row < M) {
                if (j < K1) d_fa[(size_t)row * K1 + j] = acc[r];
                else d_fg[(size_t)row * K2 + (j - K1)] = acc[r];
            }
        }
        return;
    }
    tile -= n0;
    // ---- kinds 1, 2: contraction over the M minibatch rows; A(i, m) = P[m][i], B(m, j) = Q[m][j] ----
    const bool k1 = tile < n1;
    if (!k1) tile -= n1;
    const int tiles_j = k1 ? KC / 16 : F / 16, ti = tile / tiles_j, tj = tile % tiles_j;
    const int rows_out = k1 ? F : A + 1;  // kind 2: row A = the value head
    if (ti * 16 >= rows_out) return;
    const int i = ti * 16 + x, j = tj * 16 + x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float asum = 0.0f;
    for (int m0 = 0; m0 < M; m0 += 32 * kHeadUB) {
        float a[2 * kHeadUB][4], b[2 * kHeadUB][4];
#pragma unroll
        for (int u = 0; u < 2 * kHeadUB; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int m = m0 + 16 * u + 4 * kq + s, mc = min(m, M - 1);
                float av, bv;
                if (k1) {
                    av = dH[(size_t)mc * F + i];
                    bv = j < K1 ? fa[(size_t)mc * K1 + j] : fg[(size_t)mc * K2 + (j - K1)];
                } else {
                    av = i < A ? d_logits[(size_t)mc * A + min(i, A - 1)] : (i == A ? d_values[mc] : 0.0f);
                    bv = feat[(size_t)mc * F + j];
                }
                a[u][s] = m < M ? av : 0.0f;
                b[u][s] = bv;
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2 * kHeadUB; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc = mfma4(a[u][s], b[u][s], acc);
                asum += a[u][s];
            }
    }
    // bias gradient = column sums of the A operand (tiles of the first column only)
    if (tj == 0) {
        asum += __shfl_xor(asum, 16, kWave);
        asum += __shfl_xor(asum, 32, kWave);
        if (kq == 0) {
            if (k1) gb_out[i] = asum;
            else if (i < A) gb_act[i] = asum;
            else if (i == A) gb_val[0] = asum;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + 4 * kq + r;
        if (k1) gW_out[(size_t)row * KC + j] = acc[r];
        else if (row < A) gW_act[(size_t)row * F + j] = acc[r];
        else if (row == A) gW_val[j] = acc[r];
    }
}

}  // namespace

GNBV_API int gnbv_policy_head_forward(const float *fa, const float *fg, int M, int K1, int K2, const float *W_out, const float *b_out, int F,
                                      const float *W_act, const float *b_act, int A, const float *W_val, const float *b_val, float *feat,
                                      float *logits, float *values, void *stream)
{
    GNBV_CHECK_ARG(fa && fg && W_out && b_out && W_act && b_act && W_val && b_val && feat && logits && values);
    GNBV_CHECK_ARG(M > 0 && A > 0 && F >= 16 && F % 16 == 0 && K1 >= 16 && K1 % 16 == 0 && K2 >= 16 && K2 % 16 == 0);
    GNBV_CHECK_ARG((((uintptr_t)fa | (uintptr_t)fg | (uintptr_t)W_out | (uintptr_t)W_act | (uintptr_t)W_val | (uintptr_t)feat) & 15) == 0);
    hipStream_t st = gnbv_stream(stream);
    const int mt = (M + 15) / 16;
    // (Both as ONE launch -- a workgroup per 16 rows, the features handed over in LDS -- was measured in round 3: 578-582 -> 601-606 us per
    // minibatch.  Eight workgroups pull all of W_out, 512 KiB each, through eight CUs' load paths; 32 workgroups share it out.)
    // ONE wave (tile) per workgroup in the three launches with at most a few hundred tiles: a tile pulls its 64 KiB of operands through
    // its CU's load path (~23 GB/s per CU), so 128 tiles on 128 CUs instead of four each on 32
    hipLaunchKernelGGL(k_head_fwd_feat, dim3(mt * (F / 16)), dim3(kWave), 0, st, fa, fg, M, K1, K2, W_out, b_out, F, feat);
    int err;
    if ((err = gnbv_launch_status())) return err;
    hipLaunchKernelGGL(k_head_fwd_out, dim3(mt * ((A + 1 + 15) / 16)), dim3(kWave), 0, st, (const float *)feat, M, F, W_act, b_act,
                       A, W_val, b_val, logits, values);
    return gnbv_launch_status();
}

GNBV_API int gnbv_policy_head_backward(const float *fa, const float *fg, int M, int K1, int K2, const float *feat, const float *d_logits,
                                       const float *d_values, const float *W_out, int F, const float *W_act, int A, const float *W_val,
                                       float *dH_scratch, float *d_fa, float *d_fg, float *gW_out, float *gb_out, float *gW_act,
                                       float *gb_act, float *gW_val, float *gb_val, void *stream)
{
    GNBV_CHECK_ARG(fa && fg && feat && d_logits && d_values && W_out && W_act && W_val && dH_scratch && d_fa && d_fg);
    GNBV_CHECK_ARG(gW_out && gb_out && gW_act && gb_act && gW_val && gb_val);
    GNBV_CHECK_ARG(M > 0 && A > 0 && F >= 16 && F % 16 == 0 && K1 >= 16 && K1 % 16 == 0 && K2 >= 16 && K2 % 16 == 0);
    GNBV_CHECK_ARG((((uintptr_t)dH_scratch) & 15) == 0);
    hipStream_t st = gnbv_stream(stream);
    const int mt = (M + 15) / 16, KC = K1 + K2;
    hipLaunchKernelGGL(k_head_bwd_dh, dim3(mt * (F / 16)), dim3(kWave), 0, st, d_logits, d_values, M, A, W_act, W_val, F, feat,
                       dH_scratch);
    int err;
    if ((err = gnbv_launch_status())) return err;
    const int n0 = mt * (KC / 16), n1 = (F / 16) * (KC / 16), n2 = ((A + 1 + 15) / 16) * (F / 16);
    hipLaunchKernelGGL(k_head_bwd_rest, dim3((n0 + n1 + n2 + 3) / 4), dim3(kHeadThreads), 0, st, fa, fg, M, K1, K2, feat, (const float *)dH_scratch,
                       F, d_logits, d_values, A, W_out, d_fa, d_fg, gW_out, gb_out, gW_act, gb_act, gW_val, gb_val, n0, n1);
    return gnbv_launch_status();
}
