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
#include <cstdlib>

#include "common.h"
#include "../../include/gennbv_hip.h"

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

// ---------------------------------------------------------------------------
// The same stage 1 on the f16 matrix pipe with SPLIT fp32 operands (see csrc/conv_split.h for the arithmetic): x = hi + lo with
// hi = f16(x), lo = f16(x - hi), products accumulated in fp32 as hi.hi + lo.hi + hi.lo -- three `v_mfma_f32_16x16x32_f16`
// (16 cycles each) per 32 k instead of eight `v_mfma_f32_16x16x4_f32` (32 cycles each).  The fp32 kernel above is bound by the
// fp32 matrix pipe (3.5 GFLOP = 22 us at the peak, 40-53 us measured); this one by the 83 MB stream.
// Same grid, chunking and staging; a trip = 32 k: lane (row i, g) loads x[row][k0 + 8 g .. + 7] (two 16-byte requests), thread t
// loads W[col][k0 + 4 (t % 8) .. + 3] for two columns, splits them and stores them as B operands [column tile][g][column][8 halfs].
// Exact power-of-two scalings keep the `lo` parts in f16's normal range (x 2^6 for x, |x| clamped to 1015; x 2^12 for W, |W|
// clamped to 15.8 -- both far outside what a normalised activation / a 54 000-input linear layer holds).
// ---------------------------------------------------------------------------
typedef _Float16 lh8 __attribute__((ext_vector_type(8)));
typedef _Float16 lh4 __attribute__((ext_vector_type(4)));
constexpr float kLinXScale = 64.0f, kLinWScale = 4096.0f, kLinMax = 65000.0f;
__device__ __forceinline__ void lsplit(float v, float scale, _Float16 &hi, _Float16 &lo)
{
    const float t = __builtin_amdgcn_fmed3f(v * scale, -kLinMax, kLinMax);
    hi = (_Float16)t;
    lo = (_Float16)(t - (float)hi);
}
__device__ __forceinline__ void lsplit8(const float4 &p, const float4 &q, float scale, lh8 &hi, lh8 &lo)
{
    const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        _Float16 a, b;
        lsplit(v[e], scale, a, b);
        hi[e] = a;
        lo[e] = b;
    }
}
// FOLD (round 3): x is a BatchNorm pre-activation [M][C][P] and the operand is relu(scale[c] x + shift[c]), c = column / P -- the
// BN2 + ReLU pass in front of fc_grid (k_bn_relu_apply: a 27 MB write, a 27 MB read and a launch on the update's critical path)
// folded into this kernel's operand load.  P >= 512 (the caller checks): a chunk of <= 512 columns then meets at most one channel
// boundary, and a 32-column trip that does not contain it takes the scalar path (one fma + one max per element).  The largest
// operand is tracked for the range guard (bit 4 of *range_flag when it passes 1000: the split clamps x at 1015).
// WAVES = 4: a workgroup = 128 rows (the minibatch); WAVES = 8 (round 5): 256 rows (the rollout's policy evaluation) behind ONE staged W tile --
// two 128-row workgroups staged (fetched, split, stored) every W tile twice: 66 us for a launch whose operands take 20 at the HBM rate.
// Same per-row arithmetic in the same order: bit-identical rows.
template <bool FOLD, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_linear_splitk_split(
    const float *__restrict__ x /*[M][K]*/, const float *w /*[N][K]*/, int M, int N, int K, int nchunks,
    float *__restrict__ partial /*[nchunks][M][N]*/, const float *__restrict__ xs_scale = nullptr, const float *__restrict__ xs_shift = nullptr,
    int xs_P = 0, int *__restrict__ range_flag = nullptr)
{
    __shared__ __attribute__((aligned(16))) char s_b[2][2][4096];  // [buffer][hi | lo][column tile 4][g 4][column 16][8 halfs]
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    // 1-D grid: the row slabs (128 rows each) of one (chunk, column tile) sit next to each other in dispatch order and on ONE XCD, so
    // that the second slab's W requests meet the first one's lines in that XCD's L2 (the rollout's 256 rows: W was fetched twice)
    constexpr int kThreads = 64 * WAVES, kRows = 32 * WAVES, kNR = 512 / kThreads;  // rows per workgroup; staging repetitions per thread
    static_assert(WAVES == 4 || WAVES == 8, "128 or 256 rows per workgroup");
    const int nslab = (M + kRows - 1) / kRows;
    const int xcd = blockIdx.x & 7, t_ = blockIdx.x >> 3, slot = t_ / nslab;
    const int i = lane & 15, g = lane >> 4, mbase = (t_ - slot * nslab) * kRows;
    const int ntn = N / kTileN;
    const int nt = slot % ntn, chunk = (slot / ntn) * 8 + xcd;
    if (chunk >= nchunks) return;
    const int nstep32 = (K + 31) / 32;  // 32-k trips
    const int s0 = (int)((int64_t)chunk * nstep32 / nchunks), s1 = (int)((int64_t)(chunk + 1) * nstep32 / nchunks);
    if (s0 >= s1) {  // (a chunk without work still owns its partial slab)
        float *out = partial + (size_t)chunk * M * N;
        for (int r = threadIdx.x; r < kRows * kTileN; r += kThreads) {
            const int row = mbase + r / kTileN;
            if (row < M) out[(size_t)row * N + nt * kTileN + r % kTileN] = 0.0f;
        }
        return;
    }
    const float *xr[kRowTilesPerWave];
#pragma unroll
    for (int rt = 0; rt < kRowTilesPerWave; ++rt) xr[rt] = x + (size_t)min(mbase + (wv * kRowTilesPerWave + rt) * 16 + i, M - 1) * K + 8 * g;
    const int t8 = threadIdx.x & 7, tcol = threadIdx.x >> 3;
    const float *wst[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) wst[r] = w + (size_t)(nt * kTileN + (kNR == 2 ? 32 * r : 0) + tcol) * K + 4 * t8;  // (WAVES = 8: one column per thread, r = 1 unused)
    // LDS byte offsets: staging store of this thread's (column 32 r + tcol, k-quad t8); B-operand read of this lane
    uint32_t st_off[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int col = (kNR == 2 ? 32 * r : 0) + tcol;
        st_off[r] = (uint32_t)((((col >> 4) * 4 + (t8 >> 1)) * 16 + (col & 15)) * 16 + (t8 & 1) * 8);
    }
    const uint32_t rd_off = (uint32_t)((g * 16 + i) * 16);
    f32x4 acc[kRowTilesPerWave][4];
#pragma unroll
    for (int rt = 0; rt < kRowTilesPerWave; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // requests are UNCONDITIONAL (clamped addresses): see the fp32 kernel
    struct Trip { float4 a00, a01, a10, a11, b0, b1; };  // (named members: arrays handed to lambdas ended up in scratch memory elsewhere)
    static_assert(kRowTilesPerWave == 2, "two row tiles per wave");
    auto request_a = [&](int s, Trip &t) {
        const int kc = min(min(s, s1 - 1) * 32, K - 8 - 8 * g);
        t.a00 = ld4g(xr[0] + kc);
        t.a01 = ld4g(xr[0] + kc + 4);
        t.a10 = ld4g(xr[1] + kc);
        t.a11 = ld4g(xr[1] + kc + 4);
    };
    auto request_b = [&](int s, Trip &t) {
        const int kc = min(min(s, s1 - 1) * 32, K - 4 - 4 * t8);
        t.b0 = ld4g(wst[0] + kc);
        if (kNR == 2) t.b1 = ld4g(wst[1] + kc);
    };
    auto stage_one = [&](int buf, bool ok, const float4 &b, uint32_t off) {
        const float v[4] = {b.x, b.y, b.z, b.w};
        lh4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 p, q;
            lsplit(ok ? v[e] : 0.0f, kLinWScale, p, q);
            hi[e] = p;
            lo[e] = q;
        }
        *reinterpret_cast<lh4 *>(&s_b[buf][0][off]) = hi;
        *reinterpret_cast<lh4 *>(&s_b[buf][1][off]) = lo;
    };
    auto stage_b = [&](int buf, int s, Trip &t) {
        const bool ok = s < s1 && s * 32 + 4 * t8 + 3 < K;  // past the chunk / past K: zeros
        stage_one(buf, ok, t.b0, st_off[0]);
        if (kNR == 2) stage_one(buf, ok, t.b1, st_off[1]);
    };
    // FOLD: the channel of the chunk's first column, the first column of the next channel, both channels' (scale, shift)
    float f_sc[2] = {0.f, 0.f}, f_sh[2] = {0.f, 0.f}, zmax = 0.0f;
    int f_kb = 0x7fffffff;
    if (FOLD) {
        const int c_lo = (s0 * 32) / xs_P, nch = (K + xs_P - 1) / xs_P;
        f_kb = (c_lo + 1) * xs_P;
        f_sc[0] = xs_scale[c_lo]; f_sh[0] = xs_shift[c_lo];
        f_sc[1] = xs_scale[min(c_lo + 1, nch - 1)]; f_sh[1] = xs_shift[min(c_lo + 1, nch - 1)];
    }
    auto fold8 = [&](float4 &p, float4 &q, int k0) {  // the eight operands of columns k0 .. k0 + 7
        float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool hi = k0 + e >= f_kb;
            v[e] = fmaxf(fmaf(hi ? f_sc[1] : f_sc[0], v[e], hi ? f_sh[1] : f_sh[0]), 0.0f);
        }
        zmax = fmaxf(zmax, fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]))));
        p = make_float4(v[0], v[1], v[2], v[3]);
        q = make_float4(v[4], v[5], v[6], v[7]);
    };
    auto consume = [&](int s, int buf, Trip &t) {
        const bool ok = s < s1 && s * 32 + 8 * g + 7 < K;
        if (FOLD) {
            const int k0 = min(min(s, s1 - 1) * 32, K - 8 - 8 * g) + 8 * g;  // (the columns request_a fetched for this lane)
            if (s * 32 + 32 <= f_kb || s * 32 >= f_kb) {  // (wave-uniform: the trip lies inside one channel)
                const int u = s * 32 >= f_kb ? 1 : 0;
                const float sc = f_sc[u], sh = f_sh[u];
                float *vv[4] = {&t.a00.x, &t.a01.x, &t.a10.x, &t.a11.x};
#pragma unroll
                for (int h = 0; h < 4; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float z = fmaxf(fmaf(sc, vv[h][e], sh), 0.0f);
                        vv[h][e] = z;
                        zmax = fmaxf(zmax, z);
                    }
            } else {
                fold8(t.a00, t.a01, k0);
                fold8(t.a10, t.a11, k0);
            }
        }
        lh8 ah[kRowTilesPerWave], al[kRowTilesPerWave];
        lsplit8(t.a00, t.a01, ok ? kLinXScale : 0.0f, ah[0], al[0]);
        lsplit8(t.a10, t.a11, ok ? kLinXScale : 0.0f, ah[1], al[1]);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const lh8 bh = *reinterpret_cast<const lh8 *>(&s_b[buf][0][ct * 1024 + rd_off]);
            const lh8 bl = *reinterpret_cast<const lh8 *>(&s_b[buf][1][ct * 1024 + rd_off]);
#pragma unroll
            for (int rt = 0; rt < kRowTilesPerWave; ++rt) {
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], bh, acc[rt][ct], 0, 0, 0);
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rt], bh, acc[rt][ct], 0, 0, 0);
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], bl, acc[rt][ct], 0, 0, 0);
            }
        }
    };
    // Two trips per iteration.  On entry: LDS buffer 0 = W of trip s; t0.a / t1.a = x of trips s / s + 1; t1.b = W of trip s + 1.
    // Every request is issued one trip before its data is needed (x before the other trip's MFMAs, W before the other
    // trip's staging), all unconditional.
    Trip t0, t1;
    request_a(s0, t0);
    request_b(s0, t0);
    request_a(s0 + 1, t1);
    request_b(s0 + 1, t1);
    stage_b(0, s0, t0);
    __syncthreads();
    for (int s = s0; s < s1; s += 2) {
        request_b(s + 2, t0);
        __builtin_amdgcn_sched_barrier(0);
        consume(s, 0, t0);
        __builtin_amdgcn_sched_barrier(0);
        request_a(s + 2, t0);
        stage_b(1, s + 1, t1);
        __syncthreads();
        request_b(s + 3, t1);
        __builtin_amdgcn_sched_barrier(0);
        consume(s + 1, 1, t1);
        __builtin_amdgcn_sched_barrier(0);
        request_a(s + 3, t1);
        stage_b(0, s + 2, t0);
        __syncthreads();
    }
    if (FOLD && range_flag != nullptr && __any(zmax > 1000.0f) && lane == 0) atomicOr(range_flag, 4);
    const float unscale = 1.0f / (kLinXScale * kLinWScale);
    float *out = partial + (size_t)chunk * M * N;
#pragma unroll
    for (int rt = 0; rt < kRowTilesPerWave; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = mbase + (wv * kRowTilesPerWave + rt) * 16 + 4 * g + r;
            if (row < M) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) out[(size_t)row * N + nt * kTileN + ct * 16 + i] = acc[rt][ct][r] * unscale;
            }
        }
}

// out[m][n] = act(bias[n] + sum_c partial[c][m][n]); one thread per 4 consecutive columns, the chunk
// loop is split over `kRedSplit` threads whose sub-sums are combined in a fixed order through LDS.
constexpr int kRedSplit = 4;
constexpr int kRedQ = 16;
__global__ __launch_bounds__(kRedSplit * kRedQ) void k_linear_reduce(const float *__restrict__ partial, const float *__restrict__ bias, int M, int N,
                                                       int nchunks, int relu, float *__restrict__ out)
{
    // kRedQ float4 elements x kRedSplit parts per workgroup (round 3: 16 x 4 = one wave; with 64 x 4 a workgroup pulled 128 chunks x 1 KiB
    // through its CU's load path, ~23 GB/s per CU: the 6 us of this launch at fc_grid's shape; 512 one-wave workgroups share it out.
    // Same sums in the same order.)
    __shared__ float4 sub[kRedSplit][kRedQ];
    const int q = threadIdx.x % kRedQ, part = threadIdx.x / kRedQ;
    const int64_t e4 = (int64_t)blockIdx.x * kRedQ + q, total4 = (int64_t)M * N / 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e4 < total4) {
        const int per = (nchunks + kRedSplit - 1) / kRedSplit, c0 = part * per, c1 = min(nchunks, c0 + per);
        // kU chunks requested before the first is added (clamped duplicates past the end: unconditional requests) -- as a plain
        // loop this was one round trip per chunk, 32 in a row on the update's critical path; same order of additions
        constexpr int kU = 16;  // (round 3: sixteen -- fc_grid's 128 chunks over four parts are two round trips instead of four)
        for (int c = c0; c < c1; c += kU) {
            float4 v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) v[u] = ld4g(partial + ((size_t)min(c + u, c1 - 1) * M * N) + e4 * 4);
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (c + u < c1) {
                    s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
                }
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

// ---------------------------------------------------------------------------
// Backward of the same layer on the split-f16 path:  g = d_out * (out > 0);  dx = g W  [M][K];  dW = g^T x  [N][K];  db = sum_m g.
// Both products are skinny GEMMs whose streamed operand (W [N][K] for dx, x [M][K] for dW) is row-major ALONG THE OUTPUT
// dimension, i.e. the contraction index is its ROW: one kernel serves both,
//     C[i][k] = sum_c A[i][c] B[c][k],   B row-major [C][K] streamed once, A small (g or g^T) and pre-split,
// with B's 32 x 64 tile staged per k-step (fp32 -> f16 hi | lo -> LDS [k block of 16][c][16]) and handed to the MFMA by the
// transposing LDS read (ds_read_b64_tr_b16, see csrc/conv_split.h), three `v_mfma_f32_16x16x32_f16` per operand pair.
// The library GEMMs they replace run at 70 % of the fp32 matrix peak (32 us / 60 us); these are bound by the 83 / 138 MB they move.
//   k_fc_bwd_prep: one wave per row of A: rows m of g (for dx) and rows n of g^T (for dW; also db).  A gradient has no fixed
//                  range, so every row gets its own power-of-two scale (a row's scale factors out of its output row), then is
//                  split and written in MFMA fragment order [row tile][k-step][hi | lo][lane][8]; k slots of lane group q:
//                  c = 32 s + 4 q + j and 32 s + 16 + 4 q + j (j < 4) -- the order the transposing read delivers B in.
// ---------------------------------------------------------------------------
typedef short ls4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lh8 ltr_pair(const char *p0, const char *p1)
{
    typedef __attribute__((address_space(3))) ls4 *lds_s4;
    const ls4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)p0), b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)p1);
    typedef short ls8 __attribute__((ext_vector_type(8)));
    const ls8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return *reinterpret_cast<const lh8 *>(&r);
}
__device__ __forceinline__ float pow2_scale_for(float amax)  // the power of two that puts amax into [2^13, 2^14); 1 for 0 / inf / nan
{
    int e = 0;
    (void)frexpf(amax, &e);
    return amax > 0.0f && amax < 3.0e38f ? ldexpf(1.0f, 14 - e) : 1.0f;
}
// fragment element offset (in halfs, within one [hi | lo] plane pair start) of row i, contraction index c
__device__ __forceinline__ size_t frag_pos(int row, int c, int ksteps, int hl)
{
    const int rt = row >> 4, i = row & 15, s = c >> 5, w = c & 31, half = w >> 4, ww = w & 15, q = ww >> 2, j = (ww & 3) + 4 * half;
    return ((((size_t)rt * ksteps + s) * 2 + hl) * 64 + 16 * q + i) * 8 + j;
}
__global__ __launch_bounds__(256) void k_fc_bwd_prep(const float *__restrict__ d_out, const float *__restrict__ out, int M, int N,
                                                     _Float16 *__restrict__ fragA_dx /*rows m, c = n*/, float *__restrict__ inv_dx /*[M]*/,
                                                     _Float16 *__restrict__ fragA_dw /*rows n, c = m*/, float *__restrict__ inv_dw /*[N]*/,
                                                     float *__restrict__ db /*[N] or NULL*/)
{
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool is_dx = wave < M;
    const int row = is_dx ? wave : wave - M;
    if (!is_dx && row >= N) return;
    const int C = is_dx ? N : M;                  // contraction length
    const int Cp = (C + 31) & ~31, ksteps = Cp / 32;
    // C <= 256 (gnbv_linear_bwd_prep): at most four values per lane, ALL requested before the first is used and kept for the second
    // pass (the two run-time loops were up to four dependent round trips each in a launch that sits on the update's critical path;
    // same operations in the same order per lane: the same bits)
    float gvs[4];
    {
        float o[4], d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = min(lane + 64 * u, C - 1);
            const size_t idx = is_dx ? (size_t)row * N + c : (size_t)c * N + row;
            o[u] = out[idx];
            d[u] = d_out[idx];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) gvs[u] = (lane + 64 * u < C && o[u] > 0.0f) ? d[u] : 0.0f;
    }
    float amax = 0.0f, sum = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < C) {
            amax = fmaxf(amax, fabsf(gvs[u]));
            sum += gvs[u];
        }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        amax = fmaxf(amax, __shfl_xor(amax, d, 64));
        sum += __shfl_xor(sum, d, 64);
    }
    const float sc = pow2_scale_for(amax);
    _Float16 *frag = is_dx ? fragA_dx : fragA_dw;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = lane + 64 * u;
        if (c < Cp) {
            _Float16 hi, lo;
            const float t = gvs[u] * sc;  // (gvs[u] is 0 for c >= C: the padding columns of the last k-step)
            hi = (_Float16)t;
            lo = (_Float16)(t - (float)hi);
            frag[frag_pos(row, c, ksteps, 0)] = hi;
            frag[frag_pos(row, c, ksteps, 1)] = lo;
        }
    }
    if (lane == 0) {
        (is_dx ? inv_dx : inv_dw)[row] = 1.0f / sc;
        if (!is_dx && db != nullptr) db[row] = sum;
    }
}

// C[i][k] = inv_a[i] / bscale * sum_c Afrag[i][c] (bscale B[c][k]);  workgroup = 64 columns k, 4 waves x RT row tiles
// FOLD: B[c][k] is a BatchNorm pre-activation and the operand is relu(scale[k / P] B + shift[k / P]) (see k_linear_splitk_split);
// a thread stages the same four columns in every step, so its four (scale, shift) pairs are loop constants.
template <int RT, bool FOLD = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_skinny_gemm_split(const uint4 *__restrict__ fragA, const float *__restrict__ inv_a, const float *__restrict__ Bm /*[C][K]*/,
                                                           int rows /*of C*/, int Cc /*contraction length*/, int K, float bscale, float *__restrict__ Cout /*[rows][K]*/,
                                                           double *__restrict__ sq_partial = nullptr /*[gridDim.x]: sum of the squares this workgroup stored*/,
                                                           const float *__restrict__ xs_scale = nullptr, const float *__restrict__ xs_shift = nullptr, int xs_P = 0)
{
    __shared__ __attribute__((aligned(16))) char s_b[2][2][4096];  // [buffer][hi | lo][k block 4][c 32][16 k] f16
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
    const int i = lane & 15, q = lane >> 4;
    const int k0 = blockIdx.x * 64;
    const int ksteps = (Cc + 31) / 32;
    // staging: thread t, piece u: c row cc = (t + 256 u) / 16, k quad kq = (t + 256 u) % 16
    const int cc0 = threadIdx.x >> 4, kq = threadIdx.x & 15;
    const int kcol = min(k0 + 4 * kq, K - 4);  // (clamped: a partial last slab re-reads valid columns; its stores are masked)
    const uint32_t st_off = (uint32_t)((((kq >> 2) * 32 + cc0) * 16 + (kq & 3) * 4) * 2);
    float f_sc[4] = {1.f, 1.f, 1.f, 1.f}, f_sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (FOLD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (kcol + e) / xs_P;
            f_sc[e] = xs_scale[c];
            f_sh[e] = xs_shift[c];
        }
    }
    f32x4 acc[RT][4];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[r][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto request_b = [&](int s, float4 &b0, float4 &b1) {
        const int c = min(32 * min(s, ksteps - 1) + cc0, Cc - 1), c2 = min(32 * min(s, ksteps - 1) + cc0 + 16, Cc - 1);
        b0 = ld4g(Bm + (size_t)c * K + kcol);
        b1 = ld4g(Bm + (size_t)c2 * K + kcol);
    };
    auto stage_b = [&](int buf, int s, const float4 &b0, const float4 &b1) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float4 &b = u ? b1 : b0;
            const bool ok = s < ksteps && 32 * s + cc0 + 16 * u < Cc;
            float v[4] = {b.x, b.y, b.z, b.w};
            if (FOLD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(f_sc[e], v[e], f_sh[e]), 0.0f);
            }
            lh4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 a, c;
                lsplit(ok ? v[e] : 0.0f, bscale, a, c);
                hi[e] = a;
                lo[e] = c;
            }
            *reinterpret_cast<lh4 *>(&s_b[buf][0][st_off + u * 16 * 32]) = hi;  // (c + 16: 16 rows of 32 bytes further)
            *reinterpret_cast<lh4 *>(&s_b[buf][1][st_off + u * 16 * 32]) = lo;
        }
    };
    auto request_a = [&](int s, uint4 (&ah)[RT], uint4 (&al)[RT]) {
        const int sc = min(s, ksteps - 1);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int rt = min(wv * RT + r, (rows + 15) / 16 - 1);
            ah[r] = fragA[(((size_t)rt * ksteps + sc) * 2 + 0) * 64 + lane];
            al[r] = fragA[(((size_t)rt * ksteps + sc) * 2 + 1) * 64 + lane];
        }
    };
    float4 b0, b1;
    uint4 ah[RT], al[RT];
    request_b(0, b0, b1);
    request_a(0, ah, al);
    stage_b(0, 0, b0, b1);
    request_b(1, b0, b1);
    __syncthreads();
    for (int s = 0; s < ksteps; ++s) {
        const int buf = s & 1;
        uint4 nh[RT], nl[RT];
        request_a(s + 1, nh, nl);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const char *blk = &s_b[buf][0][ct * 1024 + lane * 8];
            const lh8 bh = ltr_pair(blk, blk + 512), bl = ltr_pair(blk + 4096, blk + 4096 + 512);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const lh8 xh = *reinterpret_cast<const lh8 *>(&ah[r]), xl = *reinterpret_cast<const lh8 *>(&al[r]);
                acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bh, acc[r][ct], 0, 0, 0);
                acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, bh, acc[r][ct], 0, 0, 0);
                acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bl, acc[r][ct], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        stage_b(buf ^ 1, s + 1, b0, b1);
        request_b(s + 2, b0, b1);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            ah[r] = nh[r];
            al[r] = nl[r];
        }
        __syncthreads();
    }
    const float ib = 1.0f / bscale;
    double sq = 0.0;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int rt = wv * RT + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = rt * 16 + 4 * q + e;
            if (row < rows) {
                const float sc = inv_a[row] * ib;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const int k = k0 + ct * 16 + i;
                    const float v = acc[r][ct][e] * sc;
                    if (k < K) {
                        Cout[(size_t)row * K + k] = v;
                        sq += (double)v * (double)v;
                    }
                }
            }
        }
    }
    // The gradient-norm clip needs sum g^2 over ALL parameters before the first one is updated; for the 13.8 M weights of fc_grid
    // (94 % of them) that sum is taken here, while the values are in registers, instead of by a 55 MB pass of its own behind the
    // backward (k_grad_sqnorm: 17 us on the update's critical path).  Fixed order: lanes by xor-shuffle, waves 0..3, workgroups by
    // the consumer (gnbv_clip_adam_step_ex).
    if (sq_partial != nullptr) {
        __shared__ double s_sq[4];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) sq += __shfl_xor(sq, d, 64);
        if (lane == 0) s_sq[wv] = sq;
        __syncthreads();
        if (threadIdx.x == 0) sq_partial[blockIdx.x] = (s_sq[0] + s_sq[1]) + (s_sq[2] + s_sq[3]);
    }
}

inline int pick_chunks(int K)
{
    const int nstep16 = (K + 15) / 16;
    int c = 128;  // 2 workgroups of 4 waves per CU at N = 256
    // keep >= 4 steps per chunk (round 6: 8 before -- fc_grid at the reference's 20^3, K = 1 024, was 8 chunks = 32 workgroups; 16 chunks:
    // -1.35 us per minibatch, 32: +1.5; profiles/r06_ab_train_g20_lin_min_steps.json.  K = 54 000 has 26 steps per chunk either way)
    while (c > 1 && nstep16 / c < 4) c >>= 1;
    return c;
}

}  // namespace

GNBV_API size_t gnbv_linear_workspace_bytes(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (size_t)pick_chunks(K) * M * N * sizeof(float) + 256;
}

namespace {
inline bool split_kernels_on()
{
    const char *e = getenv("GENNBV_CONV_SPLIT");  // "0": the fp32-MFMA kernels everywhere (csrc/encoder.hip conv_split_path)
    return !(e && e[0] == '0');
}
}  // namespace

GNBV_API int gnbv_linear_fold_ok(int M, int N, int K, int P)
{
    if (M <= 0 || N <= 0 || K < 64 || K % 8 != 0 || P < 512 || K % P != 0 || !split_kernels_on()) return 0;
    const int nchunks = pick_chunks(K), nstep32 = (K + 31) / 32;
    return ((nstep32 + nchunks - 1) / nchunks + 1) * 32 <= P ? 1 : 0;  // (a chunk meets at most one channel boundary)
}

GNBV_API int gnbv_linear_forward_fold(const float *y, const float *scale, const float *shift, int P, int *range_flag, const float *w, const float *bias,
                                      int M, int N, int K, int relu, float *out, void *workspace, size_t workspace_bytes, void *stream)
{
    GNBV_CHECK_ARG(y && scale && shift && w && bias && out && workspace);
    GNBV_CHECK_ARG(N > 0 && N % kTileN == 0 && gnbv_linear_fold_ok(M, N, K, P) && !(relu & 2));
    GNBV_CHECK_ARG((((uintptr_t)y | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)out | (uintptr_t)workspace) & 15) == 0);
    GNBV_CHECK_ARG(workspace_bytes >= gnbv_linear_workspace_bytes(M, N, K));
    hipStream_t st = gnbv_stream(stream);
    const int nchunks = pick_chunks(K), ntn = N / kTileN;
    const int blocks = ((nchunks + 7) / 8) * 8 * ntn;
    if (M > 128)  // (the rollout: 256 rows behind one staged W tile)
        hipLaunchKernelGGL((k_linear_splitk_split<true, 8>), dim3(blocks * ((M + 255) / 256)), dim3(512), 0, st, y, w, M, N, K, nchunks, (float *)workspace, scale,
                           shift, P, range_flag);
    else
        hipLaunchKernelGGL((k_linear_splitk_split<true, 4>), dim3(blocks * ((M + 127) / 128)), dim3(kLinThreads), 0, st, y, w, M, N, K, nchunks, (float *)workspace, scale,
                           shift, P, range_flag);
    int err;
    if ((err = gnbv_launch_status())) return err;
    const int64_t total4 = (int64_t)M * N / 4;
    hipLaunchKernelGGL(k_linear_reduce, dim3((unsigned)((total4 + kRedQ - 1) / kRedQ)), dim3(kRedSplit * kRedQ), 0, st, (const float *)workspace, bias, M, N, nchunks,
                       relu & 1, out);
    return gnbv_launch_status();
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
    const bool fp32_arith = (relu & 2) != 0;  // (flag word: include/gennbv_hip.h)
    relu &= 1;
    if (split_kernels_on() && !fp32_arith && K % 8 == 0 && K >= 64 && M > 128)
        hipLaunchKernelGGL((k_linear_splitk_split<false, 8>), dim3(blocks * ((M + 255) / 256)), dim3(512), 0, st, x, w, M, N, K, nchunks, (float *)workspace);
    else if (split_kernels_on() && !fp32_arith && K % 8 == 0 && K >= 64)
        hipLaunchKernelGGL((k_linear_splitk_split<false, 4>), dim3(blocks * ((M + 127) / 128)), dim3(kLinThreads), 0, st, x, w, M, N, K, nchunks, (float *)workspace);
    else
        hipLaunchKernelGGL(k_linear_splitk, dim3(blocks, (M + 127) / 128), dim3(kLinThreads), 0, st, x, w, M, N, K, nchunks, (float *)workspace);
    int err;
    if ((err = gnbv_launch_status())) return err;
    const int64_t total4 = (int64_t)M * N / 4;
    hipLaunchKernelGGL(k_linear_reduce, dim3((unsigned)((total4 + kRedQ - 1) / kRedQ)), dim3(kRedSplit * kRedQ), 0, st, (const float *)workspace, bias, M, N, nchunks,
                       relu, out);
    return gnbv_launch_status();
}


GNBV_API size_t gnbv_linear_bwd_workspace_bytes(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const size_t mp = (size_t)((M + 15) / 16) * 16, np = (size_t)((N + 15) / 16) * 16, cn = (size_t)((N + 31) / 32) * 32, cm = (size_t)((M + 31) / 32) * 32;
    return (mp * cn + np * cm) * 2 * sizeof(_Float16) + (mp + np) * sizeof(float) + 1024;
}

namespace {
struct FcBwdWs {
    _Float16 *frag_dx, *frag_dw;
    float *inv_dx, *inv_dw;
};
inline FcBwdWs fc_bwd_carve(void *ws, int M, int N)
{
    const size_t mp = (size_t)((M + 15) / 16) * 16, np = (size_t)((N + 15) / 16) * 16, cn = (size_t)((N + 31) / 32) * 32, cm = (size_t)((M + 31) / 32) * 32;
    FcBwdWs w;
    w.frag_dx = (_Float16 *)ws;
    w.frag_dw = w.frag_dx + mp * cn * 2;
    w.inv_dx = (float *)(((uintptr_t)(w.frag_dw + np * cm * 2) + 255) & ~(uintptr_t)255);
    w.inv_dw = w.inv_dx + mp;
    return w;
}
}  // namespace

GNBV_API int gnbv_linear_bwd_prep(const float *d_out, const float *out, int M, int N, float *db, void *workspace, size_t workspace_bytes, void *stream)
{
    GNBV_CHECK_ARG(d_out && out && workspace && M > 0 && N > 0 && M % 16 == 0 && N % 16 == 0 && M <= 256 && N <= 256);
    GNBV_CHECK_ARG(workspace_bytes >= gnbv_linear_bwd_workspace_bytes(M, N, 4) && ((uintptr_t)workspace & 255) == 0);
    const FcBwdWs w = fc_bwd_carve(workspace, M, N);
    hipLaunchKernelGGL(k_fc_bwd_prep, dim3((M + N + 3) / 4), dim3(256), 0, gnbv_stream(stream), d_out, out, M, N, w.frag_dx, w.inv_dx, w.frag_dw, w.inv_dw, db);
    return gnbv_launch_status();
}

GNBV_API int gnbv_linear_bwd_dx(const void *workspace, const float *w, int M, int N, int K, float *dx, void *stream)
{
    GNBV_CHECK_ARG(workspace && w && dx && M > 0 && M % 16 == 0 && M <= 128 && N % 16 == 0 && N <= 256 && K >= 64 && K % 4 == 0);
    GNBV_CHECK_ARG((((uintptr_t)w | (uintptr_t)dx) & 15) == 0);
    const FcBwdWs ws = fc_bwd_carve(const_cast<void *>(workspace), M, N);
    hipLaunchKernelGGL(k_skinny_gemm_split<2>, dim3((K + 63) / 64), dim3(256), 0, gnbv_stream(stream), (const uint4 *)ws.frag_dx, (const float *)ws.inv_dx, w, M, N, K,
                       kLinWScale, dx);
    return gnbv_launch_status();
}

static int linear_bwd_dw(const void *workspace, const float *x, int M, int N, int K, float *dw, double *sq_partial, void *stream,
                         const float *scale = nullptr, const float *shift = nullptr, int P = 0)
{
    GNBV_CHECK_ARG(workspace && x && dw && M > 0 && M % 16 == 0 && M <= 256 && N % 16 == 0 && N <= 256 && K >= 64 && K % 4 == 0);
    GNBV_CHECK_ARG((((uintptr_t)x | (uintptr_t)dw) & 15) == 0);
    const FcBwdWs ws = fc_bwd_carve(const_cast<void *>(workspace), M, N);
    if (scale != nullptr) {
        GNBV_CHECK_ARG(shift != nullptr && P > 0 && K % P == 0);
        hipLaunchKernelGGL((k_skinny_gemm_split<4, true>), dim3((K + 63) / 64), dim3(256), 0, gnbv_stream(stream), (const uint4 *)ws.frag_dw,
                           (const float *)ws.inv_dw, x, N, M, K, kLinXScale, dw, sq_partial, scale, shift, P);
    } else {
        hipLaunchKernelGGL(k_skinny_gemm_split<4>, dim3((K + 63) / 64), dim3(256), 0, gnbv_stream(stream), (const uint4 *)ws.frag_dw, (const float *)ws.inv_dw, x, N, M, K,
                           kLinXScale, dw, sq_partial);
    }
    return gnbv_launch_status();
}

GNBV_API int gnbv_linear_bwd_dw_fold(const void *workspace, const float *y, const float *scale, const float *shift, int P, int M, int N, int K, float *dw,
                                     double *sq_partial /*NULL: none*/, void *stream)
{
    GNBV_CHECK_ARG(scale && shift);
    return linear_bwd_dw(workspace, y, M, N, K, dw, sq_partial, stream, scale, shift, P);
}

GNBV_API int gnbv_linear_bwd_dw(const void *workspace, const float *x, int M, int N, int K, float *dw, void *stream)
{
    return linear_bwd_dw(workspace, x, M, N, K, dw, nullptr, stream);
}

GNBV_API int gnbv_linear_bwd_dw_sq_parts(int K) { return K > 0 ? (K + 63) / 64 : 0; }

GNBV_API int gnbv_linear_bwd_dw_sq(const void *workspace, const float *x, int M, int N, int K, float *dw, double *sq_partial, void *stream)
{
    GNBV_CHECK_ARG(sq_partial && (((uintptr_t)sq_partial) & 7) == 0);
    return linear_bwd_dw(workspace, x, M, N, K, dw, sq_partial, stream);
}


// ---------------------------------------------------------------------------
// Pose-history input of the encoder (gennbv/network/hybrid_encoder.py:63-74,78-80): gather of the state columns + positional
// encoding, one launch instead of five (gather, scale, sin, cos, cat):
//   out[b][24 p + i] = sin(x[b][6 p + i / 2] * (1 + i % 2)),  out[b][24 p + 12 + i] = cos(same),   i < 12, p < n_pose
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pose_encode(const float *__restrict__ base, const int64_t *__restrict__ rows, int64_t row_stride, int batch, int n_pose,
                                                     float *__restrict__ out)
{
    const int total = batch * n_pose * 12;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int i = t % 12, bp = t / 12, p = bp % n_pose, b = bp / n_pose;
        const float x = base[(rows ? rows[b] : (int64_t)b) * row_stride + 6 * p + (i >> 1)];
        const float v = x * ((i & 1) ? 2.0f : 1.0f);
        float *o = out + ((size_t)b * n_pose + p) * 24 + i;
        o[0] = sinf(v);
        o[12] = cosf(v);
    }
}

GNBV_API int gnbv_pose_encode(const float *base, const int64_t *rows, int64_t row_stride, int batch, int n_pose, float *out, void *stream)
{
    GNBV_CHECK_ARG(base && out && batch > 0 && n_pose > 0 && row_stride >= 6 * (int64_t)n_pose);
    const int total = batch * n_pose * 12;
    hipLaunchKernelGGL(k_pose_encode, dim3((total + 255) / 256), dim3(256), 0, gnbv_stream(stream), base, rows, row_stride, batch, n_pose, out);
    return gnbv_launch_status();
}
