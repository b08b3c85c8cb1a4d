// encoder.hip -- the 3-D-CNN occupancy encoder of gennbv/network/hybrid_encoder.py:38-45
//   Conv3d(1,16,k3,s2) -> BatchNorm3d(16) -> ReLU -> Conv3d(16,16,k3,s2) -> BatchNorm3d(16) -> ReLU
// forward AND backward, hand-written for gfx950 (MI355X).
//
// Why not the library path: MIOpen runs these convolutions as a per-sample im2col + GEMM
// (1664 tiny GEMM launches per minibatch step, profiles/r01_torch_minibatch.txt): 18.4 ms per
// PPO minibatch at B=128, G=64.  The contraction is tiny (K = 27 and K = 432) while the
// activations are large (conv1 output at G=64: 244 MB fp32 per minibatch), so the design is
// bandwidth-first:
//   * activations are CHANNELS-LAST ([B, D, H, W, 16]): the 16 channels of a voxel are one
//     64-byte line, which is exactly one MFMA operand row -- no im2col buffer ever exists;
//   * the contraction runs on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32,
//     bit-equal to an fmaf chain) with operands loaded straight from L1/L2;
//   * BatchNorm statistics are accumulated in the producing kernel's epilogue (per-wave
//     partials, reduced in a fixed order -> deterministic), BN + ReLU of layer 1 is applied
//     in the CONSUMER's operand load, BN backward of layer 1 is fused into conv1's
//     weight-gradient kernel: the conv1 activation is written once and read three times,
//     never rewritten;
//   * conv1 reads its input rows straight out of the rollout buffer through a row-index
//     table (the minibatch gather of buffers.py:753-762 is never materialised).
//
// MFMA 16x16x4 fp32 fragment layout (cdna_hip_programming.md section 3), lane l:
//   A[i = l & 15][k = l >> 4]   B[k = l >> 4][j = l & 15]   D[i = 4*(l >> 4) + r][j = l & 15], r = 0..3
#include <cstdlib>

#include "common.h"
#include "../../include/gennbv_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kC = 16;          // channels of both conv layers
constexpr int kTaps = 27;
constexpr int kEncThreads = 256;
constexpr int kEncWaves = kEncThreads / kWave;

// Layer-1 activations (y1, dz1') are channels-last AND split by x-parity:
//   voxel (b, z, y, x) -> (((b*O1 + z)*O1 + y)*2 + (x & 1))*XH + (x >> 1)      XH = ceil(O1 / 2)
// A stride-2 convolution reads, for a fixed tap, input x = 2*ox + dx: with the split layout the 16
// consecutive outputs of an MFMA tile read 16 CONSECUTIVE voxels (1 KiB per wave-load, fully
// coalesced) instead of every other 64-byte half line; the transposed conv (dgrad) tiles are per
// x-parity already and become contiguous too.
// 32-bit index arithmetic: the host checks that the buffer has < 2^31 elements (64-bit integer
// multiplies are quarter-rate VALU ops and sat on the critical path of every operand load).
__device__ __forceinline__ uint32_t vox1(int b, int z, int y, int x, int O1)
{
    const uint32_t XH = (uint32_t)(O1 + 1) >> 1;
    return ((((uint32_t)b * O1 + z) * O1 + y) * 2 + (x & 1)) * XH + ((uint32_t)x >> 1);
}
static inline size_t y1_elems(int batch, int O1) { return (size_t)batch * O1 * O1 * 2 * ((O1 + 1) / 2) * kC; }

// Storage type of the layer-1 activations (y1, dz1'): fp32.  (The kernels keep the storage policy as a template parameter; the bf16
// storage mode of rounds 1-3 was removed in round 4: slower than fp32 on the split kernels and 1e-3-class loss deltas)
struct ActF32 {
    typedef float T;
    static __device__ __forceinline__ float4 ld4(const T *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void st4(T *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
    static __device__ __forceinline__ float ld1(const T *p) { return *p; }
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// sum over the four k-groups (lanes l, l^16, l^32, l^48): afterwards every lane holds the total
__device__ __forceinline__ float kgroup_sum(float v)
{
    v += __shfl_xor(v, 16, kWave);
    v += __shfl_xor(v, 32, kWave);
    return v;
}

// Workgroup -> (sample b, plane o) with all planes of sample b on XCD b % 8 (block i runs on XCD
// i % 8, observed; speed only): neighbouring output planes share an input plane, which then hits
// the XCD's L2 instead of being fetched twice from HBM.  Returns false for padding blocks.
__device__ __forceinline__ bool sample_plane(int B, int O, int &b, int &o)
{
    const int i = blockIdx.x, xcd = i & 7, slot = i >> 3;
    o = slot % O;
    b = (slot / O) * 8 + xcd;
    return b < B;
}
static inline int sample_plane_grid(int B, int O) { return ((B + 7) / 8) * 8 * O; }

// BN partial sums: one row of 32 floats per WORKGROUP, part[block][0][c] = sum, [1][c] = sum of squares
// / second sum.  The waves' 32-vectors meet in LDS and are added in wave order (deterministic).
// Every thread of the workgroup must call (contains a barrier).
// lanes 0..15 own channel c = lane after the k-group sum
__device__ __forceinline__ void write_partials(float *partials, int nwaves, int wv, float s, float q)
{
    __shared__ float wsum[16][2 * kC];
    s = kgroup_sum(s);
    q = kgroup_sum(q);
    const int lane = threadIdx.x & (kWave - 1);
    if (lane < kC) {
        wsum[wv][lane] = s;
        wsum[wv][kC + lane] = q;
    }
    __syncthreads();
    if (partials != nullptr && threadIdx.x < 2 * kC) {
        float t = wsum[0][threadIdx.x];
        for (int w = 1; w < nwaves; ++w) t += wsum[w][threadIdx.x];
        partials[(size_t)blockIdx.x * 2 * kC + threadIdx.x] = t;
    }
}

// same, for kernels whose lanes own channels 4*kq .. 4*kq+3 of position (lane & 15):
// reduce over the 16 positions (lanes with equal kq), lanes with m == 0 hold 4 channels each
__device__ __forceinline__ void write_partials_cl(float *partials, int nwaves, int wv, float (&s)[4], float (&q)[4])
{
    __shared__ float wsum[16][2 * kC];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            s[r] += __shfl_xor(s[r], d, kWave);
            q[r] += __shfl_xor(q[r], d, kWave);
        }
    }
    const int lane = threadIdx.x & (kWave - 1);
    if ((lane & 15) == 0) {
        const int kq = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            wsum[wv][4 * kq + r] = s[r];
            wsum[wv][kC + 4 * kq + r] = q[r];
        }
    }
    __syncthreads();
    if (partials != nullptr && threadIdx.x < 2 * kC) {
        float t = wsum[0][threadIdx.x];
        for (int w = 1; w < nwaves; ++w) t += wsum[w][threadIdx.x];
        partials[(size_t)blockIdx.x * 2 * kC + threadIdx.x] = t;
    }
}

// W2 [co][ci][27] -> the two LDS images the conv2 kernels use, written once per call so that the
// workgroups fill their LDS with coalesced 16-byte loads instead of 6912 scattered 4-byte reads:
//   fwd  image [tap][lane = 16kq + n][s] = W2[co = n][ci = 4kq+s][tap]
//   dgrad image [tap][lane = 16kq + n][s] = W2[co = 4kq+s][ci = n][tap]
// i.e. the four k-steps (s) of a tap are ONE 16-byte LDS read per lane (ds_read_b128), which the kernels issue one tap
// ahead of the MFMAs that use it (w2_tap below).
__device__ __forceinline__ void prep_w2_element(int i, const float *__restrict__ W2, float *__restrict__ img_fwd, float *__restrict__ img_dgrad)
{
    const int s = i & 3, n = (i >> 2) & 15, kq = (i >> 6) & 3, tap = i >> 8;
    img_fwd[i] = W2[((size_t)n * kC + 4 * kq + s) * kTaps + tap];
    img_dgrad[i] = W2[((size_t)(4 * kq + s) * kC + n) * kTaps + tap];
}
__device__ __forceinline__ float4 w2_tap(const float *img_lds, int tap, int lane) { return reinterpret_cast<const float4 *>(img_lds)[tap * kWave + lane]; }

__global__ void k_prep_w2(const float *__restrict__ W2, float *__restrict__ img_fwd, float *__restrict__ img_dgrad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < kTaps * 256) prep_w2_element(i, W2, img_fwd, img_dgrad);
}

// (the images only depend on W2: the conv1 forward kernel, which runs before every conv2 forward, writes them in
// passing -- a few hundred extra stores in an HBM-bound launch instead of a dependent 5 us launch)
__device__ void prep_w2_split_in_passing(const float *__restrict__ W2, float *__restrict__ w2img);  // conv_split.h: the f16 images of the split kernels
template <bool WIDE>
__device__ __forceinline__ void prep_w2_split_items(const float *__restrict__ W2, float *__restrict__ w2img, int first, int stride);
// `split_images` = 0: the fp32 images only.  The f16 images belong to the split kernels (G = 64, 128); their last 128 items are sums over
// |W2| whose loads the narrow form chains -- at the reference's 20^3, where no split kernel runs, that chain WAS the conv1 launch: 20 us
// with it, for 6 us of convolution (round 6).
__device__ __forceinline__ void prep_w2_in_passing(const float *__restrict__ W2, float *__restrict__ w2img, int split_images = 1)
{
    if (W2 != nullptr) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kTaps * 256; i += gridDim.x * blockDim.x)
            prep_w2_element(i, W2, w2img, w2img + kTaps * 256);
        if (split_images) prep_w2_split_in_passing(W2, w2img);
    }
}

// ---------------------------------------------------------------------------
// conv1 forward: in [B rows of the obs buffer, G^3 fp32] -> y1 [B,O1,O1,O1,16] (pre-BN, + bias)
// workgroup = (sample b, output plane oz); wave = output rows oy; tile = 16 outputs along x
// ---------------------------------------------------------------------------
template <typename A, typename IN = float>
__global__ __launch_bounds__(kEncThreads) void k_conv1_fwd(
    const IN *__restrict__ obs_base, const int64_t *__restrict__ rows, int64_t row_stride, int B, int G, int O1,
    const float *__restrict__ W1 /*[16][27]*/, const float *__restrict__ b1, typename A::T *__restrict__ y1,
    float *__restrict__ partials, const float *__restrict__ W2 = nullptr, float *__restrict__ w2img = nullptr, int split_images = 1)
{
    prep_w2_in_passing(W2, w2img, split_images);
    int b, oz;
    const bool live = sample_plane(B, O1, b, oz);
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform -> SALU index math
    const int m = lane & 15, kq = lane >> 4;
    float s_sum[4] = {0.f, 0.f, 0.f, 0.f}, s_sq[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const IN *in = obs_base + (rows ? rows[b] : (int64_t)b) * row_stride;
        // MFMA roles: A[i = co][k = tap] = W1, B[k = tap][j = output position] = input voxel
        //   -> D[i = co = 4*kq + r][j = position m]: a lane owns 4 consecutive channels of one
        //      voxel = one 16-byte channels-last store.
        float wf[7];
        int off[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const int t = 4 * s + kq;
            const bool ok = t < kTaps;
            wf[s] = ok ? W1[m * kTaps + t] : 0.0f;  // A[i = co = m][k = tap]
            const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
            off[s] = ok ? (dz * G + dy) * G + dx : 0;
        }
        // The weights are SrcA of every MFMA below.  A register last written by a global load is a slow MFMA source when waves
        // share the SIMD (tools/ubench/README.md: 50-58 instead of 34 cycles per MFMA); one VALU move each puts them on the fast path.
#pragma unroll
        for (int s = 0; s < 7; ++s) asm volatile("v_mov_b32 %0, %0" : "+v"(wf[s]));
        const float4 bias = *reinterpret_cast<const float4 *>(b1 + 4 * kq);
        // trips = (output row, 32-output x range); the 14 operand requests of trip k+1 are issued
        // before the MFMAs / stores of trip k (register double buffer, pinned by scheduling barriers)
        const int nx = (O1 + 31) / 32, nrow = (O1 - wv + kEncWaves - 1) / kEncWaves, ntrip = nrow > 0 ? nrow * nx : 0;
        auto request = [&](int k, float (&v)[2][7]) {
            const int oy = wv + kEncWaves * (k / nx), ox0 = 32 * (k % nx);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ox = min(ox0 + 16 * t + m, O1 - 1);
                const IN *p = in + ((size_t)(2 * oz) * G + 2 * oy) * G + 2 * ox;
#pragma unroll
                for (int s = 0; s < 7; ++s) v[t][s] = (float)p[off[s]];
            }
        };
        auto consume = [&](int k, const float (&v)[2][7]) {
            const int oy = wv + kEncWaves * (k / nx), ox0 = 32 * (k % nx);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 7; ++s) acc = mfma4(wf[s], v[t][s], acc);
                const int oxm = ox0 + 16 * t + m;
                if (oxm < O1) {
                    float4 y = make_float4(acc[0] + bias.x, acc[1] + bias.y, acc[2] + bias.z, acc[3] + bias.w);
                    A::st4(y1 + vox1(b, oz, oy, oxm, O1) * kC + 4 * kq, y);
                    s_sum[0] += y.x; s_sq[0] += y.x * y.x;
                    s_sum[1] += y.y; s_sq[1] += y.y * y.y;
                    s_sum[2] += y.z; s_sq[2] += y.z * y.z;
                    s_sum[3] += y.w; s_sq[3] += y.w * y.w;
                }
            }
        };
        // (requests unconditional: see k_conv2_fwd)
        float va[2][7], vb[2][7];
        if (ntrip > 0) request(0, va);
        for (int k = 0; k < ntrip; k += 2) {
            request(min(k + 1, ntrip - 1), vb);
            __builtin_amdgcn_sched_barrier(0);
            consume(k, va);
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 >= ntrip) break;
            request(min(k + 2, ntrip - 1), va);
            __builtin_amdgcn_sched_barrier(0);
            consume(k + 1, vb);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    write_partials_cl(partials, kEncWaves, wv, s_sum, s_sq);
}

// LDS-staged conv1 forward (used when G % 4 == 0 and the slab fits): the workgroup's input slab -- the
// first 2*O1+1 rows of planes 2oz, 2oz+1, 2oz+2, each a contiguous run of the observation row -- is
// fetched ONCE with full-width 16-byte requests (the direct kernel gathers it with 4-byte stride-2
// requests, every input dword 3.4 times), the MFMA operands are then 4-byte LDS reads.
// IN = float: the grid slice of the fp32 observation rows; IN = int8_t: the compact side copy of the tri-class
// grid the state-encoding kernel writes next to it (values -1/0/1: the conversion is exact, the slab is
// fetched with a quarter of the bytes and widened while it is written to LDS).
// LDS8 (int8 input only): the slab stays int8 in LDS (a quarter of the bytes: G = 128 fits, 48 KiB) and is widened
// when the MFMA operand is read; no faster than the fp32 slab where that fits (measured at G = 64), so only used beyond.
template <typename A, typename IN, bool LDS8 = false>
__global__ __launch_bounds__(kEncThreads) void k_conv1_fwd_lds(
    const IN *__restrict__ obs_base, const int64_t *__restrict__ rows, int64_t row_stride, int B, int G, int O1,
    const float *__restrict__ W1 /*[16][27]*/, const float *__restrict__ b1, typename A::T *__restrict__ y1,
    float *__restrict__ partials, const float *__restrict__ W2 = nullptr, float *__restrict__ w2img = nullptr, int split_images = 1)
{
    prep_w2_in_passing(W2, w2img, split_images);
    extern __shared__ __attribute__((aligned(16))) float s_in[];  // [3][NR][G]
    int b, oz;
    const bool live = sample_plane(B, O1, b, oz);
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int m = lane & 15, kq = lane >> 4;
    const int NR = 2 * O1 + 1, plane4 = NR * G / 4, total4 = 3 * plane4;
    float s_sum[4] = {0.f, 0.f, 0.f, 0.f}, s_sq[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const IN *in = obs_base + (rows ? rows[b] : (int64_t)b) * row_stride + (size_t)(2 * oz) * G * G;
        if constexpr (sizeof(IN) == 4) {
            // ---- stage the slab: 4 requests per thread in flight ----
            for (int base = 0; base < total4; base += 4 * kEncThreads) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = min(base + u * kEncThreads + (int)threadIdx.x, total4 - 1);
                    const int p = idx / plane4, r = idx - p * plane4;
                    v[u] = ActF32::ld4(reinterpret_cast<const float *>(in) + (size_t)p * G * G + 4 * r);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + u * kEncThreads + (int)threadIdx.x;
                    if (idx < total4) reinterpret_cast<float4 *>(s_in)[idx] = v[u];
                }
            }
        } else {
            // int8 grid: 16 voxels per 16-byte request (G % 16 == 0), widened on the way into LDS
            const int plane16 = plane4 / 4, total16 = 3 * plane16;
            for (int base = 0; base < total16; base += 2 * kEncThreads) {
                uint4 v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int idx = min(base + u * kEncThreads + (int)threadIdx.x, total16 - 1);
                    const int p = idx / plane16, r = idx - p * plane16;
                    v[u] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const int8_t *>(in) + (size_t)p * G * G + 16 * r);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int idx = base + u * kEncThreads + (int)threadIdx.x;
                    if (LDS8) {
                        if (idx < total16) reinterpret_cast<uint4 *>(s_in)[idx] = v[u];
                    } else if (idx < total16) {
                        const uint32_t wd[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            reinterpret_cast<float4 *>(s_in)[4 * idx + q] =
                                make_float4((float)(int8_t)(wd[q] & 255u), (float)(int8_t)((wd[q] >> 8) & 255u),
                                            (float)(int8_t)((wd[q] >> 16) & 255u), (float)(int8_t)(wd[q] >> 24));
                    }
                }
            }
        }
    }
    __syncthreads();
    if (live) {
        // MFMA roles as in k_conv1_fwd: A[i = co][k = tap] = W1, B[k = tap][j = position] = input voxel
        float wf[7];
        int off[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const int t = 4 * s + kq;
            const bool ok = t < kTaps;
            wf[s] = ok ? W1[m * kTaps + t] : 0.0f;
            const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
            off[s] = ok ? (dz * NR + dy) * G + dx : 0;
        }
        // (SrcA registers re-written by the VALU after their global load: see k_conv1_fwd)
#pragma unroll
        for (int s = 0; s < 7; ++s) asm volatile("v_mov_b32 %0, %0" : "+v"(wf[s]));
        const float4 bias = *reinterpret_cast<const float4 *>(b1 + 4 * kq);
        for (int oy = wv; oy < O1; oy += kEncWaves) {
            for (int ox0 = 0; ox0 < O1; ox0 += 16) {
                const int ox = min(ox0 + m, O1 - 1);
                float v[7];
                if (LDS8) {
                    const int8_t *p = reinterpret_cast<const int8_t *>(s_in) + 2 * oy * G + 2 * ox;
#pragma unroll
                    for (int s = 0; s < 7; ++s) v[s] = (float)p[off[s]];
                } else {
                    const float *p = s_in + 2 * oy * G + 2 * ox;
#pragma unroll
                    for (int s = 0; s < 7; ++s) v[s] = p[off[s]];
                }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 7; ++s) acc = mfma4(wf[s], v[s], acc);
                const int oxm = ox0 + m;
                if (oxm < O1) {
                    float4 y = make_float4(acc[0] + bias.x, acc[1] + bias.y, acc[2] + bias.z, acc[3] + bias.w);
                    if (partials != nullptr) {
                        s_sum[0] += y.x; s_sq[0] += y.x * y.x;
                        s_sum[1] += y.y; s_sq[1] += y.y * y.y;
                        s_sum[2] += y.z; s_sq[2] += y.z * y.z;
                        s_sum[3] += y.w; s_sq[3] += y.w * y.w;
                    }
                    A::st4(y1 + vox1(b, oz, oy, oxm, O1) * kC + 4 * kq, y);
                }
            }
        }
    }
    if (partials != nullptr) write_partials_cl(partials, kEncWaves, wv, s_sum, s_sq);
}

__device__ __forceinline__ void fill_lds_image(float *lds, const float *__restrict__ img)
{
    for (int i = threadIdx.x; i < kTaps * 64; i += blockDim.x)
        reinterpret_cast<float4 *>(lds)[i] = reinterpret_cast<const float4 *>(img)[i];
    __syncthreads();
}

// Workgroup -> (sample b, group of PZ planes) with all groups of sample b on XCD b % 8.
__device__ __forceinline__ bool sample_plane_group(int B, int O, int PZ, int &b, int &o0, int &o1, int block = -1)
{
    const int ng = (O + PZ - 1) / PZ;
    const int i = block < 0 ? (int)blockIdx.x : block, xcd = i & 7, slot = i >> 3;
    const int gidx = slot % ng;
    b = (slot / ng) * 8 + xcd;
    o0 = gidx * PZ;
    o1 = min(O, o0 + PZ);
    return b < B;
}
static inline int sample_plane_group_grid(int B, int O, int PZ) { return ((B + 7) / 8) * 8 * ((O + PZ - 1) / PZ); }
constexpr int kPlanesPerGroup = 4;
// Planes per workgroup of the fp32 conv2 forward kernel: the smallest of 1, 2, 4 whose grid is at most 512 workgroups (two
// per CU: one round), four beyond that.  At the reference's 20^3 (O2 = 4, 128 samples) four planes were ONE workgroup per sample whose
// waves walked two tiles one after the other in a launch that is nothing but latency:
// forward 128 -> 512 workgroups: 14.3 -> 11.7 us.  (The data gradient does NOT gain: see gnbv_encoder_grid_backward.)
static inline int planes_per_group(int B, int O)
{
    for (int pz = 1; pz < kPlanesPerGroup; pz <<= 1)
        if (sample_plane_group_grid(B, O, pz) <= 512) return pz;
    return kPlanesPerGroup;
}
constexpr int kBigThreads = 1024;  // conv2 fwd / dgrad: 16 waves share one 27 KiB weight image -> 32 waves per CU
constexpr int kBigWaves = kBigThreads / kWave;

// ---------------------------------------------------------------------------
// conv2 forward: z1 = relu(scale1*y1 + shift1) applied on load; y2 [B,16,O2^3] (NCDHW, pre-BN)
// ---------------------------------------------------------------------------
// Workgroup = 12 waves (three per SIMD, <= 168 VGPRs each): room for the third slab buffer; 4 planes x 15 rows = 60 tiles = 5 per wave.
constexpr int kFwdThreads = 768, kFwdWaves = kFwdThreads / kWave;
template <typename A>
__global__ __launch_bounds__(kFwdThreads) void k_conv2_fwd(
    const typename A::T *__restrict__ y1, const float *__restrict__ scale1, const float *__restrict__ shift1, int B, int O1, int O2,
    const float *__restrict__ W2img /*k_prep_w2 fwd image*/, const float *__restrict__ b2, float *__restrict__ y2,
    float *__restrict__ partials, int PZ /*output planes per workgroup: planes_per_group()*/)
{
    __shared__ __attribute__((aligned(16))) float w2s[kTaps * 4 * 4 * kC];  // [tap][lane = 16kq + n][s] = W2[n][4kq+s][tap]
    fill_lds_image(w2s, W2img);
    int b, oz0, oz1;
    const bool live = sample_plane_group(B, O2, PZ, b, oz0, oz1);
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform -> SALU index math
    const int m = lane & 15, kq = lane >> 4;
    if (!live) { write_partials(partials, kFwdWaves, wv, 0.f, 0.f); return; }
    float sc[4], sh[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        sc[s] = scale1[4 * kq + s];
        sh[s] = shift1[4 * kq + s];
    }
    const float bias = b2[m];
    const int P2 = O2 * O2 * O2;
    float s_sum = 0.0f, s_sq = 0.0f;
    const int ntile_x = (O2 + 15) / 16, nwork = (oz1 - oz0) * O2 * ntile_x;
    const uint32_t XHC = ((uint32_t)(O1 + 1) >> 1) * kC, rowC = 2 * XHC;
    // Register ring over dz slabs (9 taps = 9 KiB per wave each), one buffer per dz: the requests of slab s+2
    // (possibly a slab of the wave's next tile) are issued BEFORE the 36 MFMAs of slab s.
    // Left to itself the compiler sinks every load next to its use (load, s_waitcnt vmcnt(0),
    // 4 MFMA): one 1 KiB request in flight per wave, latency-bound at ~3.5 TB/s.  The scheduling
    // barriers pin the request groups where they are written.
    auto request = [&](int wk, int dz, float4 (&v)[9]) {
        const int oz = oz0 + wk / (O2 * ntile_x), rr = wk % (O2 * ntile_x), oy = rr / ntile_x;
        const int ox = min((rr % ntile_x) * 16 + m, O2 - 1);
        const uint32_t base = vox1(b, 2 * oz + dz, 2 * oy, 2 * ox, O1) * kC + 4 * kq;  // even-parity voxel ox
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3, dx = t % 3;
            // dx = 0, 2: even plane, voxels ox, ox + 1; dx = 1: odd plane, voxel ox
            v[t] = A::ld4(y1 + base + (dy * rowC + (dx == 1 ? XHC : 0) + (dx == 2 ? kC : 0)));
        }
    };
    auto consume = [&](int dz, float4 (&v)[9], f32x4 &acc) {
        // BN + ReLU of the whole slab IN PLACE first, then the 36 MFMAs back to back: left to the compiler every MFMA is
        // preceded by its own v_fma, v_max, s_nop (46 cycles per MFMA in tools/ubench/mfma_valu_groups.hip against 37-38
        // when the vector work is grouped in front of >= 8 MFMAs: what costs is the MFMA -> VALU -> MFMA turn-around, not
        // the vector instructions themselves).
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            v[t].x = fmaxf(fmaf(sc[0], v[t].x, sh[0]), 0.f);
            v[t].y = fmaxf(fmaf(sc[1], v[t].y, sh[1]), 0.f);
            v[t].z = fmaxf(fmaf(sc[2], v[t].z, sh[2]), 0.f);
            v[t].w = fmaxf(fmaf(sc[3], v[t].w, sh[3]), 0.f);
        }
        // weights: one 16-byte LDS read per tap, issued ONE TAP AHEAD of its MFMAs and pinned there (left alone the
        // compiler puts every LDS read right in front of its use: ds_read, s_waitcnt lgkmcnt(0), 2 MFMAs -- the LDS
        // latency exposed 54 times per tile)
        float4 wq = w2_tap(w2s, dz * 9, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 wn = w2_tap(w2s, dz * 9 + (t < 8 ? t + 1 : t), lane);
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma4(v[t].x, wq.x, acc);
            acc = mfma4(v[t].y, wq.y, acc);
            acc = mfma4(v[t].z, wq.z, acc);
            acc = mfma4(v[t].w, wq.w, acc);
            __builtin_amdgcn_sched_barrier(0);
            wq = wn;
        }
    };
    auto finish = [&](int wk, const f32x4 &acc) {
        const int oz = oz0 + wk / (O2 * ntile_x), rr = wk % (O2 * ntile_x), oy = rr / ntile_x, ox0 = (rr % ntile_x) * 16;
        float *out = y2 + ((size_t)b * kC + m) * P2 + ((size_t)oz * O2 + oy) * O2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oxi = ox0 + 4 * kq + r;
            if (oxi < O2) {
                const float y = acc[r] + bias;
                out[oxi] = y;
                s_sum += y;
                s_sq += y * y;
            }
        }
    };
    // Requests are unconditional (past the last tile they re-read it): a branch around a request group
    // makes the s_waitcnt insertion assume the worst case at the join, i.e. wait for the prefetch too.
    // Three slab buffers, one per dz: slab (tile, dz) is requested TWO slabs (72 MFMAs) before it is consumed.
    float4 v0[9], v1[9], v2[9];
    if (wv < nwork) {
        request(wv, 0, v0);
        request(wv, 1, v1);
    }
    for (int wk = wv; wk < nwork; wk += kFwdWaves) {
        const int nxt = min(wk + kFwdWaves, nwork - 1);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        request(wk, 2, v2);
        __builtin_amdgcn_sched_barrier(0);
        consume(0, v0, acc);
        __builtin_amdgcn_sched_barrier(0);
        request(nxt, 0, v0);
        __builtin_amdgcn_sched_barrier(0);
        consume(1, v1, acc);
        __builtin_amdgcn_sched_barrier(0);
        request(nxt, 1, v1);
        __builtin_amdgcn_sched_barrier(0);
        consume(2, v2, acc);
        finish(wk, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    write_partials(partials, kFwdWaves, wv, s_sum, s_sq);
}

// ---------------------------------------------------------------------------
// BatchNorm bookkeeping.  sums [2][16] come from k_reduce_partials.
//   train: mean / biased var from (sum, sumsq); running stats updated with the UNBIASED var
//          (torch.nn.BatchNorm3d, momentum 0.1) unless *skip_flag != 0
//   eval : statistics = running stats
//   out  : scale = gamma*rstd, shift = beta - mean*scale, mean, rstd
// ---------------------------------------------------------------------------
__device__ __forceinline__ void bn_finalize_channel(int c, double sum, double sumsq, double count, const float *__restrict__ gamma,
                                                    const float *__restrict__ beta, float eps, float momentum, int training,
                                                    float *__restrict__ running_mean, float *__restrict__ running_var,
                                                    int64_t *__restrict__ num_batches_tracked, const int *__restrict__ skip_flag,
                                                    float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean_out,
                                                    float *__restrict__ rstd_out)
{
    float mean, var;
    if (training) {
        const double mu = sum / count;
        double v = sumsq / count - mu * mu;
        v = v < 0.0 ? 0.0 : v;
        mean = (float)mu;
        var = (float)v;
        const bool skip = skip_flag != nullptr && *skip_flag != 0;
        if (!skip && running_mean != nullptr) {
            const float unbiased = (float)(v * (count / (count > 1.0 ? count - 1.0 : 1.0)));
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
            if (c == 0 && num_batches_tracked != nullptr) num_batches_tracked[0] += 1;
        }
    } else {
        mean = running_mean[c];
        var = running_var[c];
    }
    const float rstd = 1.0f / sqrtf(var + eps);
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    mean_out[c] = mean;
    rstd_out[c] = rstd;
}

// Range guard of the split-f16 kernels (conv_split.h): z1 = relu(scale * y1 + shift) is stored as f16 x 2^8 and CLAMPED at 253.9.
// conv1's input is a tri-class grid (|x| <= 1), so |y1[c] - b1[c]| <= sum_t |W1[c][t]| and
//     z1[c] <= |scale| * sum_t |W1[c][t]| + |scale * b1[c] + shift|
// -- a bound from the parameters alone, evaluated by the one thread that just wrote scale[c] / shift[c].  A channel whose bound
// reaches the clamp sets bit 1 of *range_flag (GnbvEncoderParams.range_flag); the host mirror reads the word once per train() /
// rollout and raises (or selects the fp32-MFMA kernels, GnbvEncoderParams.force_fp32).  Default-initialised and normally trained
// layers sit at 5 - 40.
__device__ __forceinline__ void l1_range_guard(int c, const float *__restrict__ W1, const float *__restrict__ b1, const float *scale,
                                               const float *shift, int *__restrict__ range_flag)
{
    if (range_flag == nullptr || W1 == nullptr) return;
    float s = 0.0f;
    for (int t = 0; t < kTaps; ++t) s += fabsf(W1[c * kTaps + t]);
    const float bound = fabsf(scale[c]) * s + fabsf(fmaf(scale[c], b1[c], shift[c]));
    if (!(bound <= 253.0f)) atomicOr(range_flag, 2);
}

// eval mode: statistics = running stats (no reduction)
__global__ void k_bn_finalize(double count, const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float momentum,
                              float *__restrict__ running_mean, float *__restrict__ running_var, float *__restrict__ scale,
                              float *__restrict__ shift, float *__restrict__ mean_out, float *__restrict__ rstd_out,
                              const float *__restrict__ rg_W1 = nullptr, const float *__restrict__ rg_b1 = nullptr, int *__restrict__ range_flag = nullptr)
{
    const int c = threadIdx.x;
    if (c < kC) {
        bn_finalize_channel(c, 0.0, 0.0, count, gamma, beta, eps, momentum, 0, running_mean, running_var, nullptr, nullptr, scale, shift,
                            mean_out, rstd_out);
        l1_range_guard(c, rg_W1, rg_b1, scale, shift, range_flag);  // (layer 1 only: the callers pass W1 for it)
    }
}

// data-parallel: BatchNorm finalize (train mode) from batch sums that were summed over the replicas
__global__ void k_bn_finalize_sums(const double *__restrict__ sums /*[2][16]: sum, sum of squares*/, double count, const float *__restrict__ gamma,
                                   const float *__restrict__ beta, float eps, float momentum, float *__restrict__ running_mean,
                                   float *__restrict__ running_var, int64_t *__restrict__ num_batches_tracked, const int *__restrict__ skip_flag,
                                   float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean_out, float *__restrict__ rstd_out)
{
    const int c = threadIdx.x;
    if (c < kC)
        bn_finalize_channel(c, sums[c], sums[kC + c], count, gamma, beta, eps, momentum, 1, running_mean, running_var, num_batches_tracked,
                            skip_flag, scale, shift, mean_out, rstd_out);
}

// data-parallel: out[32] = sum over the slices of the fused backward's BN1-backward sums (tmp[sl][512 ..  512 + 32))
__global__ void k_gather_bn1_sums(const double *__restrict__ tmp, int slices, double *__restrict__ out)
{
    const int i = threadIdx.x;
    if (i >= 2 * kC) return;
    double a = 0.0;
    for (int sl = 0; sl < slices; ++sl) a += tmp[(size_t)sl * (512 + 2 * kC) /*kE1F*/ + 512 + i];
    out[i] = a;
}

// One workgroup sums the [P][32] per-workgroup partial rows in fp64 and in a fixed order (128 slices of
// P, then the slices in order), writes the 32 sums (out_d, optional) and, when gamma != nullptr,
// finalizes the BatchNorm layer in the same launch (train mode).
__global__ __launch_bounds__(1024) void k_stats_reduce(const float *__restrict__ partial, int P, double *__restrict__ out_d, double count,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                       float momentum, float *__restrict__ running_mean, float *__restrict__ running_var,
                                                       int64_t *__restrict__ num_batches_tracked, const int *__restrict__ skip_flag,
                                                       float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean_out,
                                                       float *__restrict__ rstd_out, const float *__restrict__ W2, float *__restrict__ w2img,
                                                       const float *__restrict__ rg_W1 = nullptr, const float *__restrict__ rg_b1 = nullptr,
                                                       int *__restrict__ range_flag = nullptr)
{
    // (same launch, independent work) the two LDS weight images of the conv2 kernels: saves a dependent launch
    if (W2 != nullptr)
        for (int i = threadIdx.x; i < kTaps * 256; i += 1024) prep_w2_element(i, W2, w2img, w2img + kTaps * 256);
    // thread = (slice of P: 128) x (4 consecutive sums: 8); 8 independent 16-byte requests in flight
    __shared__ double sh[128][2 * kC];
    __shared__ double tot[2 * kC];
    const int e4 = (threadIdx.x & 7) * 4, sl = threadIdx.x >> 3;
    const int per = (P + 127) / 128, p0 = sl * per, p1 = min(P, p0 + per);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int p = p0; p < p1; p += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ActF32::ld4(partial + (size_t)min(p + u, p1 - 1) * 2 * kC + e4);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (p + u < p1) {
                a0 += (double)v[u].x; a1 += (double)v[u].y; a2 += (double)v[u].z; a3 += (double)v[u].w;
            }
    }
    sh[sl][e4] = a0; sh[sl][e4 + 1] = a1; sh[sl][e4 + 2] = a2; sh[sl][e4 + 3] = a3;
    __syncthreads();
    if (threadIdx.x < 2 * kC) {
        double t = 0.0;
        for (int i = 0; i < 128; ++i) t += sh[i][threadIdx.x];
        tot[threadIdx.x] = t;
        if (out_d != nullptr) out_d[threadIdx.x] = t;
    }
    __syncthreads();
    if (gamma != nullptr && threadIdx.x < kC) {
        bn_finalize_channel(threadIdx.x, tot[threadIdx.x], tot[kC + threadIdx.x], count, gamma, beta, eps, momentum, 1, running_mean,
                            running_var, num_batches_tracked, skip_flag, scale, shift, mean_out, rstd_out);
        l1_range_guard(threadIdx.x, rg_W1, rg_b1, scale, shift, range_flag);
    }
}

// out[e] = sum_p partial[p][e] in fp64 and in a fixed order (deterministic), two stages:
//   stage 1: grid (ceil(E/64), slices): every workgroup sums a contiguous slice of P (4 waves
//            stride through it, combined through LDS) -> tmp[slice][E] (fp64)
//   stage 2: the same kernel over tmp with one slice.
template <typename T>
__global__ __launch_bounds__(256) void k_reduce_partials(const T *__restrict__ partial, int P, int E, int per_slice,
                                                         double *__restrict__ out_d, float *__restrict__ out_f)
{
    __shared__ double s[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int p0 = blockIdx.y * per_slice, p1 = min(P, p0 + per_slice);
    double acc = 0.0;
    if (e < E)
        for (int p = p0 + wv; p < p1; p += 4) acc += (double)partial[(size_t)p * E + e];
    s[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && e < E) {
        const double t = ((s[0][lane] + s[1][lane]) + s[2][lane]) + s[3][lane];
        if (out_d) out_d[(size_t)blockIdx.y * E + e] = t;
        if (out_f) out_f[(size_t)blockIdx.y * E + e] = (float)t;
    }
}

// Same sums in the same order (wave w of a slice adds rows p0 + w, p0 + w + 4, ... in fp64, then ((s0 + s1) + s2) + s3), four
// columns per lane with 16-byte loads, eight rows requested before the first is added: the scalar form above ran the 14 MB of
// the conv2 weight-gradient partials at 0.85 TB/s (a chain of dependent 4-byte loads per lane) on the update's critical path.
__device__ __forceinline__ void reduce_partials4_body(const float *__restrict__ partial, int P, int E /* % 4 == 0 */, int per_slice,
                                                      double *__restrict__ out_d, int bx, int by)
{
    __shared__ double s[4][64][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int e = (bx * 64 + lane) * 4;
    const int p0 = by * per_slice, p1 = min(P, p0 + per_slice);
    const int ec = min(e, E - 4);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    constexpr int kU = 8;
    for (int p = p0 + wv; p < p1; p += 4 * kU) {
        float4 v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) v[u] = *reinterpret_cast<const float4 *>(partial + (size_t)min(p + 4 * u, p1 - 1) * E + ec);  // (clamped: unconditional)
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (p + 4 * u < p1) {
                a0 += (double)v[u].x;
                a1 += (double)v[u].y;
                a2 += (double)v[u].z;
                a3 += (double)v[u].w;
            }
    }
    s[wv][lane][0] = a0;
    s[wv][lane][1] = a1;
    s[wv][lane][2] = a2;
    s[wv][lane][3] = a3;
    __syncthreads();
    if (e < E) {  // wave w finishes column e + w
        const double t = ((s[0][lane][wv] + s[1][lane][wv]) + s[2][lane][wv]) + s[3][lane][wv];
        out_d[(size_t)by * E + e + wv] = t;
    }
}

__global__ __launch_bounds__(256) void k_reduce_partials4(const float *__restrict__ partial, int P, int E /* % 4 == 0 */, int per_slice,
                                                          double *__restrict__ out_d)
{
    reduce_partials4_body(partial, P, E, per_slice, out_d, blockIdx.x, blockIdx.y);
}

// z2 = relu(scale2*y2 + shift2), y2 NCDHW [B,16,P2] -> flat features [B, 16*P2]
// (range guard, bit 2 of *range_flag: these features are fc_grid's input, which the split-f16 linear kernels clamp at 1015,
// csrc/linear.hip -- the largest value is tracked here, where every element passes through a register anyway)
__global__ void k_bn_relu_apply(const float *__restrict__ y2, const float *__restrict__ scale, const float *__restrict__ shift,
                                int64_t total, int P2, float *__restrict__ z2, int *__restrict__ range_flag)
{
    float zmax = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / P2) % kC);
        const float z = fmaxf(fmaf(scale[c], y2[i], shift[c]), 0.0f);
        z2[i] = z;
        zmax = fmaxf(zmax, z);  // (fmaxf drops a NaN: NaNs are not this guard's business)
    }
    if (range_flag != nullptr && __any(zmax > 1000.0f) && (threadIdx.x & (kWave - 1)) == 0) atomicOr(range_flag, 4);
}

// ---------------------------------------------------------------------------
// BN2 + ReLU backward.  pass 1: per-(b, c) partial sums S1 = sum dz', S2 = sum dz'*xhat
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bn2_bwd_reduce(const float *__restrict__ dz2, const float *__restrict__ y2,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        const float *__restrict__ mean, const float *__restrict__ rstd, int P2,
                                                        float *__restrict__ partials /*[B*16][2]*/)
{
    // Workgroup = one (sample, channel) row of P2 values.  kU x 2 independent requests per lane are issued before any is
    // consumed (clamped addresses): with one request in flight per lane the 2048 short workgroups ran at 1.3 TB/s (42 us
    // for 55 MB), the loop was a chain of ~14 dependent round trips.
    constexpr int kU = 8;
    const int bc = blockIdx.x, c = bc % kC;
    const float sc = scale[c], sh = shift[c], mu = mean[c], rs = rstd[c];
    const float *yr = y2 + (size_t)bc * P2, *gr = dz2 + (size_t)bc * P2;
    float s1 = 0.f, s2 = 0.f;
    for (int i0 = threadIdx.x; i0 < P2; i0 += 256 * kU) {
        float yv[kU], gv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = min(i0 + u * 256, P2 - 1);
            yv[u] = yr[i];
            gv[u] = gr[i];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const bool live = i0 + u * 256 < P2;
            const float y = yv[u];
            const float g = (live && fmaf(sc, y, sh) > 0.0f) ? gv[u] : 0.0f;
            s1 += g;
            s2 += g * ((y - mu) * rs);
        }
    }
    __shared__ float r1[4], r2[4];
    s1 = wave_reduce_sum(s1);
    s2 = wave_reduce_sum(s2);
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[(size_t)bc * 2] = (r1[0] + r1[1]) + (r1[2] + r1[3]);
        partials[(size_t)bc * 2 + 1] = (r2[0] + r2[1]) + (r2[2] + r2[3]);
    }
}

// sums over b of the [B][16][2] partials -> S[2][16] (fp64 accumulate)
__global__ __launch_bounds__(256) void k_bn2_bwd_finalize(const float *__restrict__ partials, int B, double *__restrict__ S, unsigned *__restrict__ absmax)
{
    if (threadIdx.x < 64) absmax[threadIdx.x * 32] = 0u;
    // 32 outputs (which, c) x 8 slices of b; fixed summation order
    __shared__ double sh[8][32];
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int which = o / kC, c = o % kC;
    double acc = 0.0;
    for (int b = sl; b < B; b += 128) {  // (sixteen requests in flight -- ONE round trip at B = 128: this single workgroup runs beside
                                         // fc_grid's dW GEMM, whose stream triples the latency of each --, clamped duplicates: same order of additions)
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = partials[((size_t)min(b + 8 * u, B - 1) * kC + c) * 2 + which];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (b + 8 * u < B) acc += (double)v[u];
    }
    sh[sl][o] = acc;
    __syncthreads();
    if (sl == 0) {
        double t = 0.0;
        for (int i = 0; i < 8; ++i) t += sh[i][o];
        S[which * kC + c] = t;
    }
}

// pass 2: dy2 (channels-last [B,P2,16]) = scale*(dz' - S1/M - xhat*S2/M)
// Workgroup = (sample b, tile of 64 positions): the 16 channel rows of y2 / dz2 (NCDHW) are read with coalesced
// requests (a wave = 64 consecutive positions of one channel), transposed through LDS and written as contiguous
// channels-last vectors.  (The first version read both inputs with a 16-way channel stride: 26 us for 83 MB.)
__global__ __launch_bounds__(256) void k_bn2_bwd_apply(const float *__restrict__ dz2, const float *__restrict__ y2, const float *__restrict__ scale,
                                                      const float *__restrict__ shift, const float *__restrict__ mean,
                                                      const float *__restrict__ rstd, const double *__restrict__ S, double count, int P2,
                                                      float *__restrict__ dy2, unsigned *__restrict__ absmax /*max |dy2| as bit patterns in 64 slots of 32 uints (zeroed by k_bn2_bwd_finalize)*/)
{
    __shared__ float tile[64][kC + 1];
    const int tiles = (P2 + 63) / 64, b = blockIdx.x / tiles, pos0 = (blockIdx.x - b * tiles) * 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pos = min(pos0 + lane, P2 - 1);
    float yv[4], gv[4], amax = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = ((size_t)b * kC + (wv + 4 * k)) * P2 + pos;
        yv[k] = y2[i];
        gv[k] = dz2[i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = wv + 4 * k;
        const float y = yv[k];
        const float g = fmaf(scale[c], y, shift[c]) > 0.0f ? gv[k] : 0.0f;
        const float xhat = (y - mean[c]) * rstd[c];
        const float m1 = (float)(S[c] / count), m2 = (float)(S[kC + c] / count);
        tile[lane][c] = scale[c] * (g - m1 - xhat * m2);
        amax = fmaxf(amax, pos0 + lane < P2 ? fabsf(tile[lane][c]) : 0.0f);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) amax = fmaxf(amax, __shfl_xor(amax, d, 64));
    __shared__ float wmax[4];
    if (lane == 0) wmax[wv] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
        // one atomic per workgroup, and only when it would raise the maximum: 27 000 workgroups on one address otherwise
        // serialise (measured: this 18 us kernel took 312 us with an unconditional atomic per wave)
        const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        const unsigned bits = __float_as_uint(m);  // (non-negative floats order like their bit patterns; NaN never passes m == m)
        unsigned *slot = absmax + (blockIdx.x & 63) * 32;  // 64 slots, one cache line each
        if (m == m && bits > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, bits);
    }
    const int p = threadIdx.x >> 2, c4 = (threadIdx.x & 3) * 4;
    if (pos0 + p < P2)
        *reinterpret_cast<float4 *>(dy2 + ((size_t)b * P2 + pos0 + p) * kC + c4) =
            make_float4(tile[p][c4], tile[p][c4 + 1], tile[p][c4 + 2], tile[p][c4 + 3]);
}

// Weight-gradient kernels walk output rows (b, oz, oy).  Every wave owns a CONTIGUOUS range of rows
// (neighbouring rows share input lines -> L1/L2 hits) and the ranges of the 8 XCDs are contiguous
// too (block i runs on XCD i % 8), so a sample's planes stay within one XCD's L2.
__device__ __forceinline__ void wave_row_range(int nrows, int &r0, int &r1)
{
    const int nblk = gridDim.x, per_xcd = (nblk + 7) / 8;
    const int chunk = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);  // position of this block in row order
    const int wave = chunk * kEncWaves + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), nwaves = per_xcd * 8 * kEncWaves;
    const int per = (nrows + nwaves - 1) / nwaves;
    r0 = min(nrows, wave * per);
    r1 = min(nrows, r0 + per);
}

// Weight-gradient kernels end with a workgroup-level reduction of the per-wave accumulators through
// LDS: wave 0 stores, waves 1.. add in order (each lane owns the same slots in every wave), then the
// workgroup copies the sums to ITS row of the partial buffer -- 4x fewer partial rows to write / reduce.
// ---------------------------------------------------------------------------
// conv2 weight gradient: dW2[(tap, ci), co] = sum_pos z1[inpos(pos, tap), ci] * dy2[pos, co]
// MFMA: i = ci, j = co, k = 4 consecutive output positions along x.  27 accumulators / wave.
// partial[w][tap][ci][co] (+ 16 bias sums), reduced by k_reduce_partials.
// ---------------------------------------------------------------------------
// The epilogue of the conv2 weight-gradient kernels (see k_conv2_wgrad): `red` = two buffers of kTaps * 256 + kC floats in LDS.
__device__ __forceinline__ void wgrad_reduce_store(f32x4 (&acc)[kTaps], float bsum, float (*red)[kTaps * 256 + kC], float *__restrict__ partial, int lane,
                                                   int wv, int n, int kq)
{
    // Workgroup-level sum in LDS, fixed order (w0 + w2) + (w1 + w3), then one partial row per workgroup: [tap][ci][co] +
    // 16 bias sums.  Exchanges use one 16-byte LDS access per accumulator ([tap][lane] layout); only the final result is
    // laid out for the row.  (The first version chained four read-modify-write passes of 108 dwords each: ~12 us.)
    static_assert(kEncWaves == 4, "pairwise reduction below assumes four waves");
    bsum = kgroup_sum(bsum);  // every lane: total of channel n
    if (wv >= 2) {
        f32x4 *dst = reinterpret_cast<f32x4 *>(red[wv - 2]);
#pragma unroll
        for (int tap = 0; tap < kTaps; ++tap) dst[tap * kWave + lane] = acc[tap];
        if (lane < kC) red[wv - 2][kTaps * 256 + lane] = bsum;
    }
    __syncthreads();
    if (wv < 2) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(red[wv]);
#pragma unroll
        for (int tap = 0; tap < kTaps; ++tap) acc[tap] += src[tap * kWave + lane];
        bsum += red[wv][kTaps * 256 + n];
    }
    __syncthreads();
    if (wv == 1) {
        f32x4 *dst = reinterpret_cast<f32x4 *>(red[0]);
#pragma unroll
        for (int tap = 0; tap < kTaps; ++tap) dst[tap * kWave + lane] = acc[tap];
        if (lane < kC) red[0][kTaps * 256 + lane] = bsum;
    }
    __syncthreads();
    if (wv == 0) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(red[0]);
#pragma unroll
        for (int tap = 0; tap < kTaps; ++tap) {
            const f32x4 t = acc[tap] + src[tap * kWave + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) red[1][tap * 256 + (4 * kq + r) * kC + n] = t[r];
        }
        if (lane < kC) red[1][kTaps * 256 + lane] = bsum + red[0][kTaps * 256 + lane];
    }
    __syncthreads();
    float *out = partial + (size_t)blockIdx.x * (kTaps * 256 + kC);
    for (int o = threadIdx.x; o < kTaps * 256 + kC; o += kEncThreads) out[o] = red[1][o];
}

template <typename A>
__global__ __launch_bounds__(kEncThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_conv2_wgrad(
    const typename A::T *__restrict__ y1, const float *__restrict__ scale1, const float *__restrict__ shift1,
    const float *__restrict__ dy2 /*[B,O2,O2,O2,16]*/, int B, int O1, int O2, float *__restrict__ partial)
{
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform -> SALU index math
    const int n = lane & 15, kq = lane >> 4;
    const int wave_global = blockIdx.x * kEncWaves + wv, nwaves = gridDim.x * kEncWaves;
    const float sc = scale1[n], sh = shift1[n];
    f32x4 acc[kTaps];
#pragma unroll
    for (int t = 0; t < kTaps; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    const int nrows = B * O2 * O2;
    int row0, row1;
    wave_row_range(nrows, row0, row1);
    (void)nwaves;
    // Work items = (row, group of 4 output positions along x), flattened over the wave's rows.  Register
    // ping-pong: the 28 operand requests of item i+1 (possibly the first group of the NEXT row) are in
    // flight while the 27 MFMAs of item i run; requests are unconditional (clamped), no copies.
    const int ng = (O2 + 3) / 4, nitems = (row1 - row0) * ng;
    // Index math: the items are walked in order, so (sample, plane, row, x-group) are carried as wave-uniform counters and
    // advanced incrementally -- the divisions of the first version cost ~75 SALU instructions per item (27 MFMAs), and
    // every instruction, scalar ones included, takes an issue slot of the SIMD the MFMAs need.  The 27 operand addresses
    // are 9 wave-uniform (dz, dy) bases + one 32-bit lane offset (dx = 2: +64 B immediate, dx = 1: a second lane offset).
    const uint32_t XHC = ((uint32_t)(O1 + 1) >> 1) * kC, rowC = 2 * XHC, planeC = rowC * O1;
    const typename A::T *tapp[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tapp[t] = y1 + (size_t)(t / 3) * planeC + (size_t)(t % 3) * rowC;
    struct Cursor {
        int xg, oy, oz, b;
        uint32_t ybase, dbase;  // element offsets of (b, 2oz, 2oy, x = 0) in y1 and of (b, oz, oy, 0) in dy2
    };
    auto cursor_at = [&](int row) {
        Cursor c;
        c.b = row / (O2 * O2);
        const int rem = row - c.b * O2 * O2;
        c.oz = rem / O2;
        c.oy = rem - c.oz * O2;
        c.xg = 0;
        c.ybase = vox1(c.b, 2 * c.oz, 2 * c.oy, 0, O1) * kC;
        c.dbase = (uint32_t)row * O2 * kC;
        return c;
    };
    auto advance = [&](Cursor &c) {
        if (++c.xg < ng) return;
        c.xg = 0;
        c.dbase += O2 * kC;
        if (++c.oy == O2) {
            c.oy = 0;
            if (++c.oz == O2) { c.oz = 0; ++c.b; }
        }
        c.ybase = vox1(c.b, 2 * c.oz, 2 * c.oy, 0, O1) * kC;
    };
    Cursor rq = cursor_at(row0);  // the next item to request
    int requested = 0, cxg = 0;    // items requested so far; x-group of the next item to consume
    auto request = [&](float &bv, float (&av)[kTaps]) {
        const int xc = min(4 * rq.xg + kq, O2 - 1);
        bv = dy2[rq.dbase + xc * kC + n];
        // voxel 2xc of the even plane / voxel xc of the odd plane
        const uint32_t off = rq.ybase + (uint32_t)xc * kC + n, off1 = off + XHC;
#pragma unroll
        for (int tap = 0; tap < kTaps; ++tap) {
            const int dx = tap % 3;
            av[tap] = A::ld1(tapp[tap / 3] + (dx == 1 ? off1 : off) + (dx == 2 ? kC : 0));
        }
        if (++requested < nitems) advance(rq);  // (past the last item the same one is requested again: unconditional requests)
    };
    auto consume = [&](int it, float bv, const float (&av)[kTaps]) {
        const bool ok = 4 * cxg + kq < O2 && it < nitems;  // (odd item count: one padded, all-zero item)
        if (++cxg == ng) cxg = 0;
        const float bb = ok ? bv : 0.0f;  // positions past the row end contribute nothing
        bsum += bb;
        // (BN + ReLU per tap right in front of its MFMA: grouping nine taps' VALU work in front of nine back-to-back MFMAs, as
        // k_conv2_fwd does, measured no faster alone and 5 % slower beside the data-gradient kernel in the PPO minibatch -- this
        // kernel is bound by its 28 four-byte requests per 27 MFMAs, profiles/r01_notes.md)
#pragma unroll
        for (int tap = 0; tap < kTaps; ++tap) {
            const float a = fmaxf(fmaf(sc, av[tap], sh), 0.0f);
            acc[tap] = mfma4(a, bb, acc[tap]);
        }
    };
    float b0, a0[kTaps], b1, a1[kTaps];
    if (nitems > 0) request(b0, a0);
    for (int it = 0; it < nitems; it += 2) {
        request(b1, a1);
        __builtin_amdgcn_sched_barrier(0);
        consume(it, b0, a0);
        __builtin_amdgcn_sched_barrier(0);
        request(b0, a0);
        __builtin_amdgcn_sched_barrier(0);
        consume(it + 1, b1, a1);
        __builtin_amdgcn_sched_barrier(0);
    }
    __shared__ __attribute__((aligned(16))) float red[2][kTaps * 256 + kC];
    wgrad_reduce_store(acc, bsum, red, partial, lane, wv, n, kq);
    (void)wave_global;
}


// partial-sum layout [tap][ci][co] (+16) -> torch layout dW2 [co][ci][27], db2 [16]
// (+ the conv2 weight images for the data-gradient kernel that follows: saves a dependent launch)
__device__ __forceinline__ void conv2_wgrad_finish_body(const double *__restrict__ tmp /*[slices][E]*/, int slices, float *__restrict__ dW2,
                                                        float *__restrict__ db2, const float *__restrict__ W2, float *__restrict__ w2img, int i)
{
    const int E = kTaps * 256 + kC;
    if (W2 != nullptr && i < kTaps * 256) prep_w2_element(i, W2, w2img, w2img + kTaps * 256);
    if (i >= E) return;
    double t = 0.0;
    for (int sl0 = 0; sl0 < slices; sl0 += 16) {  // (sixteen requests in flight: the usual 16 slices are one round trip; same order of additions)
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = tmp[(size_t)min(sl0 + u, slices - 1) * E + i];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (sl0 + u < slices) t += v[u];
    }
    if (i < kTaps * 256) {
        const int tap = i >> 8, ci = (i >> 4) & 15, co = i & 15;
        dW2[((size_t)co * kC + ci) * kTaps + tap] = (float)t;
    } else {
        db2[i - kTaps * 256] = (float)t;
    }
}
__global__ void k_conv2_wgrad_finish(const double *__restrict__ tmp /*[slices][E]*/, int slices, float *__restrict__ dW2,
                                     float *__restrict__ db2, const float *__restrict__ W2, float *__restrict__ w2img)
{
    conv2_wgrad_finish_body(tmp, slices, dW2, db2, W2, w2img, blockIdx.x * blockDim.x + threadIdx.x);
}

// The transposed conv as 8 sub-convolutions.  A "super-tile" (a, c, j0) is the 2 x 2 x 32 block of
// layer-1 voxels iz in {2a, 2a+1}, iy in {2c, 2c+1}, ix = 2j + ex (j = j0 .. j0+15, ex in {0, 1}).
// Along one axis an even index 2a receives taps d = 0 (from output a) and d = 2 (from a-1), an odd
// index 2a+1 receives d = 1 (from a): the whole block depends on the 2 x 2 x 17 neighbourhood
// dy2[a-zo][c-yo][j-xo] (zo, yo, xo in {0, 1}) -- EIGHT 16-byte operand requests per lane feed all 27
// taps (108 MFMAs); per-voxel tiles would request each of them 3.4 times.  The 8 dy2 vectors and the
// 8 y1 vectors of the epilogues are requested back to back before the first MFMA (16 KiB in flight
// per wave); out-of-grid neighbours are zeros (select, no branch), so the block is straight-line.
// Accumulation order per voxel: (zo, yo, xo) lexicographic = taps (dz, dy, dx) ascending.
template <typename A, int EZ, int EY, int EX>
__device__ __forceinline__ void dgrad_subtile(
    const float4 (&L)[8], const float4 &y, const float *w2d, typename A::T *__restrict__ dz1p, uint32_t idx, bool lane_ok,
    const float4 &sc, const float4 &sh, const float4 &mu, const float4 &rs, float (&s1)[4], float (&s2)[4])
{
    const int lane = threadIdx.x & (kWave - 1);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int zo = 0; zo < (EZ ? 1 : 2); ++zo)
#pragma unroll
        for (int yo = 0; yo < (EY ? 1 : 2); ++yo)
#pragma unroll
            for (int xo = 0; xo < (EX ? 1 : 2); ++xo) {
                const int dz = EZ ? 1 : 2 * zo, dy = EY ? 1 : 2 * yo, dx = EX ? 1 : 2 * xo;
                const float4 &t = L[(zo * 2 + yo) * 2 + xo];
                const float4 wq = w2_tap(w2d, (dz * 3 + dy) * 3 + dx, lane);  // [tap][lane (kq, ci = m)][s]
                acc = mfma4(wq.x, t.x, acc);
                acc = mfma4(wq.y, t.y, acc);
                acc = mfma4(wq.z, t.z, acc);
                acc = mfma4(wq.w, t.w, acc);
            }
    if (lane_ok) {
        float4 g;
        g.x = fmaf(sc.x, y.x, sh.x) > 0.0f ? acc[0] : 0.0f;
        g.y = fmaf(sc.y, y.y, sh.y) > 0.0f ? acc[1] : 0.0f;
        g.z = fmaf(sc.z, y.z, sh.z) > 0.0f ? acc[2] : 0.0f;
        g.w = fmaf(sc.w, y.w, sh.w) > 0.0f ? acc[3] : 0.0f;
        A::st4(dz1p + idx, g);
        s1[0] += g.x; s2[0] += g.x * ((y.x - mu.x) * rs.x);
        s1[1] += g.y; s2[1] += g.y * ((y.y - mu.y) * rs.y);
        s1[2] += g.z; s2[2] += g.z * ((y.z - mu.z) * rs.z);
        s1[3] += g.w; s2[3] += g.w * ((y.w - mu.w) * rs.w);
    }
}

// ---------------------------------------------------------------------------
// conv2 data gradient (transposed conv, stride 2) + ReLU mask of layer 1 + BN1-backward sums.
//   dz1'[v, ci] = [pre1 > 0] * sum_{tap, co} dy2[(v - tap)/2, co] * W2[co][ci][tap]
// workgroup = (sample b, kPlanesPerGroup plane pairs a); wave = super-tiles (a, c)
// ---------------------------------------------------------------------------
template <typename A>
__global__ __launch_bounds__(kBigThreads) void k_conv2_dgrad(
    const float *__restrict__ dy2, const float *__restrict__ W2, const typename A::T *__restrict__ y1, const float *__restrict__ scale1,
    const float *__restrict__ shift1, const float *__restrict__ mean1, const float *__restrict__ rstd1, int B, int O1, int O2,
    typename A::T *__restrict__ dz1p, float *__restrict__ partials, int PZ /*plane pairs per workgroup: planes_per_group()*/)
{
    __shared__ __attribute__((aligned(16))) float w2d[kTaps * 4 * 4 * kC];  // [tap][lane = 16kq + n][s] = W2[co = 4kq+s][ci = n][tap]
    fill_lds_image(w2d, W2 /* k_prep_w2 dgrad image */);
    const int NA = (O1 + 1) >> 1;  // plane pairs / row pairs / voxels per x-parity
    int b, a0, a1;
    const bool live = sample_plane_group(B, NA, PZ, b, a0, a1);
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform -> SALU index math
    const int m = lane & 15, kq = lane >> 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (!live) { write_partials_cl(partials, kBigWaves, wv, s1, s2); return; }
    // MFMA roles: A[i = ci][k = co] = W2, B[k = co][j = input voxel] = dy2 -> D[i = ci = 4*kq+r][j = voxel m]:
    // a lane owns 4 consecutive channels of one voxel (16-byte load of y1, 16-byte store of dz1')
    const float4 sc = *reinterpret_cast<const float4 *>(scale1 + 4 * kq), sh = *reinterpret_cast<const float4 *>(shift1 + 4 * kq);
    const float4 mu = *reinterpret_cast<const float4 *>(mean1 + 4 * kq), rs = *reinterpret_cast<const float4 *>(rstd1 + 4 * kq);
    const int ntx = (NA + 15) / 16, nwork = (a1 - a0) * NA * ntx;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int wk = wv; wk < nwork; wk += kBigWaves) {
        const int a = a0 + wk / (NA * ntx), rr = wk % (NA * ntx), c = rr / ntx, j = (rr % ntx) * 16 + m;
        // ---- requests: 8 dy2 neighbours, 8 y1 vectors ----
        float4 L[8], Y[8];
        bool okL[8];
#pragma unroll
        for (int zo = 0; zo < 2; ++zo)
#pragma unroll
            for (int yo = 0; yo < 2; ++yo)
#pragma unroll
                for (int xo = 0; xo < 2; ++xo) {
                    const int oz = a - zo, oy = c - yo, ox = j - xo;
                    okL[(zo * 2 + yo) * 2 + xo] = oz >= 0 && oz < O2 && oy >= 0 && oy < O2 && ox >= 0 && ox < O2;
                    const int ozc = min(max(oz, 0), O2 - 1), oyc = min(max(oy, 0), O2 - 1), oxc = min(max(ox, 0), O2 - 1);
                    L[(zo * 2 + yo) * 2 + xo] = *reinterpret_cast<const float4 *>(
                        dy2 + ((((uint32_t)b * O2 + ozc) * O2 + oyc) * O2 + oxc) * kC + 4 * kq);
                }
        uint32_t idx[8];
        bool okV[8];
#pragma unroll
        for (int ez = 0; ez < 2; ++ez)
#pragma unroll
            for (int ey = 0; ey < 2; ++ey)
#pragma unroll
                for (int ex = 0; ex < 2; ++ex) {
                    const int iz = 2 * a + ez, iy = 2 * c + ey, ix = 2 * j + ex, e = (ez * 2 + ey) * 2 + ex;
                    okV[e] = iz < O1 && iy < O1 && ix < O1;
                    idx[e] = vox1(b, min(iz, O1 - 1), min(iy, O1 - 1), min(ix, O1 - 1 - ((O1 - 1 - ex) & 1)), O1) * kC + 4 * kq;
                    Y[e] = A::ld4(y1 + idx[e]);
                }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (!okL[t]) L[t] = zero4;
        dgrad_subtile<A, 0, 0, 0>(L, Y[0], w2d, dz1p, idx[0], okV[0], sc, sh, mu, rs, s1, s2);
        dgrad_subtile<A, 0, 0, 1>(L, Y[1], w2d, dz1p, idx[1], okV[1], sc, sh, mu, rs, s1, s2);
        dgrad_subtile<A, 0, 1, 0>(L, Y[2], w2d, dz1p, idx[2], okV[2], sc, sh, mu, rs, s1, s2);
        dgrad_subtile<A, 0, 1, 1>(L, Y[3], w2d, dz1p, idx[3], okV[3], sc, sh, mu, rs, s1, s2);
        dgrad_subtile<A, 1, 0, 0>(L, Y[4], w2d, dz1p, idx[4], okV[4], sc, sh, mu, rs, s1, s2);
        dgrad_subtile<A, 1, 0, 1>(L, Y[5], w2d, dz1p, idx[5], okV[5], sc, sh, mu, rs, s1, s2);
        dgrad_subtile<A, 1, 1, 0>(L, Y[6], w2d, dz1p, idx[6], okV[6], sc, sh, mu, rs, s1, s2);
        dgrad_subtile<A, 1, 1, 1>(L, Y[7], w2d, dz1p, idx[7], okV[7], sc, sh, mu, rs, s1, s2);
    }
    write_partials_cl(partials, kBigWaves, wv, s1, s2);
}

// ---------------------------------------------------------------------------
// Fused conv2 data gradient + conv1 weight gradient (fp32 activations, int8 grid rows, G % 16 == 0): dz1' is never
// written.  BN1 backward is linear in the two batch sums m1 = S1/M, m2 = S2/M that are only known after the whole
// data gradient exists, so the conv1 weight gradient is split into sums that do not need them:
//   dy1 = scale1 * (g - m1 - xhat*m2)          g = dz1' (ReLU-masked), xhat = (y1 - mean1) * rstd1
//   dW1[co][tap] = sum_pos x[pos, tap] * dy1[pos, co] = scale1[co] * (T1[tap][co] - m1[co]*T2[tap] - m2[co]*T3[tap][co])
//   T1 = sum x*g,  T2 = sum x,  T3 = sum x*xhat,  S1 = sum g,  S2 = sum g*xhat
// (db1 = sum dy1 = scale1 * (S1 - M m1 - m2 sum xhat) is identically zero: a bias in front of BatchNorm has no gradient)
// T1 is one more MFMA contraction (i = tap, j = co, k = voxel) on the data-gradient tile while it is still in
// registers.  T2 and T3 do not depend on the gradient at all: y1 = W1 * x + b1 is linear in the input, so
//   T3[tap][co] = rstd1[co] * sum_tap' W1[co][tap'] * (R[tap][tap'] - T2[tap] T2[tap'] / M),   R = sum_pos x[pos,tap] x[pos,tap']
// with R the 27 x 27 autocorrelation of the minibatch's input patches -- exact integers from k_input_autocorr (int8
// MFMA, runs beside the other backward kernels on a second stream).  k_c1w_fused_finish combines everything in fp64.
// Saves the 244 MB write + 488 MB of reads of dz1'/y1 that the separate conv1 weight-gradient kernel costs.
//
// The data-gradient MFMA runs with its operands swapped (A = dy2, B = W2): D[i = voxel 4kq+r][j = ci = lane & 15],
// i.e. the tile arrives transposed, in exactly the B-operand layout of the second contraction (B[k = voxel][j = co]);
// y1 is read with 4-byte requests in the same layout (mask + xhat).  The input operand x[voxel][tap] comes from a
// per-wave int8 slab in LDS: the 5 x 5 x 65 input voxels under one 2 x 2 x 32 super-tile.
// ---------------------------------------------------------------------------
constexpr int kSlabRow = 80, kSlabBytes = 25 * kSlabRow;  // 5 planes x 5 rows x (65 -> 80) bytes
constexpr int kE1F = 512 + 2 * kC;                        // T1 [32 taps][16], S1, S2 [16]

template <int EZ, int EY, int EX>
__device__ __forceinline__ void dgrad_c1w_subtile(
    const float4 (&L)[8], const float (&Y)[4], const float *w2d, const int8_t *slab0, const int8_t *slab1, const bool (&ok0)[4],
    const bool (&ok1)[4], float sc, float sh, float mu, float rs, float &s1, float &s2, f32x4 (&T1)[2])
{
    const int lane = threadIdx.x & (kWave - 1);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int zo = 0; zo < (EZ ? 1 : 2); ++zo)
#pragma unroll
        for (int yo = 0; yo < (EY ? 1 : 2); ++yo)
#pragma unroll
            for (int xo = 0; xo < (EX ? 1 : 2); ++xo) {
                const int dz = EZ ? 1 : 2 * zo, dy = EY ? 1 : 2 * yo, dx = EX ? 1 : 2 * xo;
                const float4 &t = L[(zo * 2 + yo) * 2 + xo];
                const float4 wq = w2_tap(w2d, (dz * 3 + dy) * 3 + dx, lane);  // [tap][lane (kq, ci = m)][s]
                acc = mfma4(t.x, wq.x, acc);  // A[i = voxel][k = co], B[k = co][j = ci]
                acc = mfma4(t.y, wq.y, acc);
                acc = mfma4(t.z, wq.z, acc);
                acc = mfma4(t.w, wq.w, acc);
            }
    constexpr int kOff = ((2 * EZ) * 5 + 2 * EY) * kSlabRow + 2 * EX;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // Out-of-grid voxels (x only: out-of-grid planes / rows skip the sub-tile) are masked in the A operand, so
        // their g never reaches T1; g is masked as well for the channel sums.
        const float y = ok0[r] ? Y[r] : 0.0f;  // (out-of-grid slots of y1 are never written: may hold Inf / NaN)
        const float g = (ok0[r] && fmaf(sc, y, sh) > 0.0f) ? acc[r] : 0.0f;
        s1 += g;
        s2 = fmaf(g, (y - mu) * rs, s2);
        const float a0v = (float)slab0[kOff + 4 * r], a1v = (float)slab1[kOff + 4 * r];  // unconditional reads, then selects
        T1[0] = mfma4(ok0[r] ? a0v : 0.0f, g, T1[0]);  // A[i = tap][k = voxel]
        T1[1] = mfma4(ok1[r] ? a1v : 0.0f, g, T1[1]);
    }
}

__global__ __launch_bounds__(kBigThreads) void k_conv2_dgrad_c1w(
    const float *__restrict__ dy2, const float *__restrict__ W2 /*dgrad image*/, const float *__restrict__ y1, const float *__restrict__ scale1,
    const float *__restrict__ shift1, const float *__restrict__ mean1, const float *__restrict__ rstd1, const int8_t *__restrict__ grid_i8,
    const int64_t *__restrict__ rows, int64_t grid_row_stride, int B, int G, int O1, int O2, float *__restrict__ partial /*[blocks][kE1F]*/)
{
    __shared__ __attribute__((aligned(16))) float w2d[kTaps * 4 * 4 * kC];
    __shared__ __attribute__((aligned(16))) int8_t slabs[kBigWaves][kSlabBytes];
    fill_lds_image(w2d, W2);
    const int NA = (O1 + 1) >> 1;  // == XH: plane pairs / row pairs / voxels per x-parity (a multiple of 4 when G % 16 == 0)
    int b, a0, a1;
    const bool live = sample_plane_group(B, NA, kPlanesPerGroup, b, a0, a1);
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int m = lane & 15, kq = lane >> 4;
    float s1 = 0.f, s2 = 0.f;
    f32x4 T1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (live) {
        const float sc = scale1[m], sh = shift1[m], mu = mean1[m], rs = rstd1[m];
        // lane's taps m and 16 + m: byte offset of the tap inside the slab, plus this lane's k-slot (4 voxels = 16 bytes apart)
        const bool tok1 = 16 + m < kTaps;
        const int t1 = tok1 ? 16 + m : 0;
        const int8_t *slab = slabs[wv];
        const int8_t *slab0 = slab + ((m / 9) * 5 + (m / 3) % 3) * kSlabRow + m % 3 + 16 * kq;
        const int8_t *slab1 = slab + ((t1 / 9) * 5 + (t1 / 3) % 3) * kSlabRow + t1 % 3 + 16 * kq;
        const int8_t *in = grid_i8 + (rows ? rows[b] : (int64_t)b) * grid_row_stride;
        const int g3m16 = G * G * G - 16;
        const int ntx = (NA + 15) / 16, nwork = (a1 - a0) * NA * ntx;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int wk = wv; wk < nwork; wk += kBigWaves) {
            const int a = a0 + wk / (NA * ntx), rr = wk % (NA * ntx), c = rr / ntx, j0 = (rr % ntx) * 16, j = j0 + m;
            // ---- requests: 8 dy2 neighbours (lane = voxel j, channels 4kq..), 2 slab vectors, y1 values ----
            float4 L[8];
            bool okL[8];
#pragma unroll
            for (int zo = 0; zo < 2; ++zo)
#pragma unroll
                for (int yo = 0; yo < 2; ++yo)
#pragma unroll
                    for (int xo = 0; xo < 2; ++xo) {
                        const int oz = a - zo, oy = c - yo, ox = j - xo;
                        okL[(zo * 2 + yo) * 2 + xo] = oz >= 0 && oz < O2 && oy >= 0 && oy < O2 && ox >= 0 && ox < O2;
                        const int ozc = min(max(oz, 0), O2 - 1), oyc = min(max(oy, 0), O2 - 1), oxc = min(max(ox, 0), O2 - 1);
                        L[(zo * 2 + yo) * 2 + xo] = *reinterpret_cast<const float4 *>(
                            dy2 + ((((uint32_t)b * O2 + ozc) * O2 + oyc) * O2 + oxc) * kC + 4 * kq);
                    }
            uint4 sv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = min(lane + 64 * u, 124), row = idx / 5, seg = idx - 5 * row, zr = row / 5, yr = row - 5 * zr;
                const int off = (min(4 * a + zr, G - 1) * G + min(4 * c + yr, G - 1)) * G + 4 * j0 + 16 * seg;
                sv[u] = *reinterpret_cast<const uint4 *>(in + min(off, g3m16));
            }
            // x validity of this lane's voxels j0 + 4kq + r (x = 2j + ex); ok1 also folds in the validity of tap 16 + m
            bool ok0[2][4], ok1[2][4];
#pragma unroll
            for (int ex = 0; ex < 2; ++ex)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ok0[ex][r] = 2 * (j0 + 4 * kq + r) + ex < O1;
                    ok1[ex][r] = ok0[ex][r] && tok1;
                }
            // y1 values of sub-tile e (this lane: channel m of voxels jb + r, four 64-byte lines apart: one base address,
            // immediate offsets; jb is clamped so that the four stay inside the row -- lanes it moves are out of grid).
            // Requested in three groups so that at most 16 + 8 of the 32 are live beside the 32 dy2 registers.
            const int jb = min(j0 + 4 * kq, NA - 4);
            float Y[8][4];
            auto request_y = [&](int e, float (&dst)[4]) {
                const int ez = e >> 2, ey = (e >> 1) & 1, ex = e & 1;
                const int iz = min(2 * a + ez, O1 - 1), iy = min(2 * c + ey, O1 - 1);
                const uint32_t blk = ((((uint32_t)b * O1 + iz) * O1 + iy) * 2 + ex) * NA * kC;  // the (row, parity) block
                const float *p = y1 + blk + (uint32_t)jb * kC + m;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r] = p[r * kC];
            };
            request_y(0, Y[0]);
            request_y(1, Y[1]);
            request_y(2, Y[2]);
            request_y(3, Y[3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (!okL[t]) L[t] = zero4;
            // the wave's private slab (the previous super-tile's reads are done: program order inside a wave)
            reinterpret_cast<uint4 *>(slabs[wv])[lane] = sv[0];
            if (lane + 64 < 125) reinterpret_cast<uint4 *>(slabs[wv])[lane + 64] = sv[1];
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
            // sub-tiles on out-of-grid planes / rows are skipped (wave-uniform)
            const bool z0 = 2 * a < O1, z1 = 2 * a + 1 < O1, y0 = 2 * c < O1, y1ok = 2 * c + 1 < O1;
            if (z0 && y0) dgrad_c1w_subtile<0, 0, 0>(L, Y[0], w2d, slab0, slab1, ok0[0], ok1[0], sc, sh, mu, rs, s1, s2, T1);
            if (z0 && y0) dgrad_c1w_subtile<0, 0, 1>(L, Y[1], w2d, slab0, slab1, ok0[1], ok1[1], sc, sh, mu, rs, s1, s2, T1);
            __builtin_amdgcn_sched_barrier(0);
            request_y(4, Y[4]);
            request_y(5, Y[5]);
            __builtin_amdgcn_sched_barrier(0);
            if (z0 && y1ok) dgrad_c1w_subtile<0, 1, 0>(L, Y[2], w2d, slab0, slab1, ok0[0], ok1[0], sc, sh, mu, rs, s1, s2, T1);
            if (z0 && y1ok) dgrad_c1w_subtile<0, 1, 1>(L, Y[3], w2d, slab0, slab1, ok0[1], ok1[1], sc, sh, mu, rs, s1, s2, T1);
            __builtin_amdgcn_sched_barrier(0);
            request_y(6, Y[6]);
            request_y(7, Y[7]);
            __builtin_amdgcn_sched_barrier(0);
            if (z1 && y0) dgrad_c1w_subtile<1, 0, 0>(L, Y[4], w2d, slab0, slab1, ok0[0], ok1[0], sc, sh, mu, rs, s1, s2, T1);
            if (z1 && y0) dgrad_c1w_subtile<1, 0, 1>(L, Y[5], w2d, slab0, slab1, ok0[1], ok1[1], sc, sh, mu, rs, s1, s2, T1);
            if (z1 && y1ok) dgrad_c1w_subtile<1, 1, 0>(L, Y[6], w2d, slab0, slab1, ok0[0], ok1[0], sc, sh, mu, rs, s1, s2, T1);
            if (z1 && y1ok) dgrad_c1w_subtile<1, 1, 1>(L, Y[7], w2d, slab0, slab1, ok0[1], ok1[1], sc, sh, mu, rs, s1, s2, T1);
            __builtin_amdgcn_wave_barrier();  // the next super-tile overwrites the slab
        }
    }
    // ---- workgroup-level sums: binary tree over the 16 waves (fixed order -> deterministic), one partial row per
    // workgroup.  Slot w of the buffer = two 16-byte vectors per lane (T1) + 32 floats (S1, S2 of channel lane & 15).
    s1 = kgroup_sum(s1);
    s2 = kgroup_sum(s2);
    __syncthreads();  // every wave is done with the weight image (reused as the reduction buffer)
    constexpr int kSlot = 2 * kWave * 4 + 2 * kC;  // floats per wave slot
    static_assert(8 * kSlot <= kTaps * 4 * 4 * kC && kBigWaves == 16, "tree reduction buffer");
    float *red = w2d;
#pragma unroll
    for (int half = kBigWaves / 2; half >= 1; half >>= 1) {
        if (wv >= half && wv < 2 * half) {
            float *slot = red + (wv - half) * kSlot;
            reinterpret_cast<f32x4 *>(slot)[lane] = T1[0];
            reinterpret_cast<f32x4 *>(slot)[kWave + lane] = T1[1];
            if (lane < kC) {
                slot[2 * kWave * 4 + lane] = s1;
                slot[2 * kWave * 4 + kC + lane] = s2;
            }
        }
        __syncthreads();
        if (wv < half) {
            const float *slot = red + wv * kSlot;
            T1[0] += reinterpret_cast<const f32x4 *>(slot)[lane];
            T1[1] += reinterpret_cast<const f32x4 *>(slot)[kWave + lane];
            s1 += slot[2 * kWave * 4 + m];
            s2 += slot[2 * kWave * 4 + kC + m];
        }
        __syncthreads();
    }
    if (wv == 0) {  // final layout: [tap][co], S1, S2
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(16 * tt + 4 * kq + r) * kC + m] = T1[tt][r];
        if (lane < kC) {
            red[512 + lane] = s1;
            red[512 + kC + lane] = s2;
        }
    }
    __syncthreads();
    float *out = partial + (size_t)blockIdx.x * kE1F;
    for (int o = threadIdx.x; o < kE1F; o += kBigThreads) out[o] = red[o];
}

// Input autocorrelation: R[t][t'] = sum over output positions of x[pos, t] * x[pos, t'], t, t' in 0..26 the conv1
// taps, t = 27 a pseudo-tap that is 1 on every valid position (so R[t][27] = T2[t] = sum x and R[27][27] = #positions).
// x in {-1, 0, 1}: int8 MFMA (v_mfma_i32_16x16x64_i8), exact.  A and B operands are the SAME registers (R = X^T X:
// lane (tap, k-group) holds 16 positions of its tap), so the k <-> position mapping of the instruction does not
// matter.  Workgroup = (sample, group of P output planes): its 2P+1 input planes are fetched once into LDS as bytes
// (one barrier); a work item = (plane, k-step of 4 units of 16 positions along x).  Results are ADDED to
// out + sample * out_row_stride with integer atomics (order-independent, deterministic; the caller zeroes):
// one row per sample (gnbv_input_autocorr: a property of the observation, computed once when the env produces the
// grid) or, with out_row_stride == 0, the sum over the minibatch.
// Row layout: 3 tiles of 16 x 16 ints = R[0..15][0..15], R[0..15][16..31], R[16..31][16..31].
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int kAcThreads = 512, kAcWaves = kAcThreads / kWave, kAcRow = 3 * 256;
static inline int autocorr_planes(int grid)
{
    const int p = (64 * 1024 / (grid * grid) - 1) / 2;
    return p < 1 ? 1 : (p > 4 ? 4 : p);
}

__global__ __launch_bounds__(kAcThreads) void k_input_autocorr(const int8_t *__restrict__ grid_i8, const int64_t *__restrict__ rows,
                                                             int64_t row_stride, int B, int G, int O1, int P, int *__restrict__ out,
                                                             int64_t out_row_stride)
{
    // LDS rows are de-interleaved by x parity ([even bytes | odd bytes], G/2 each): the 16 positions of a unit read
    // input x = 2(16 xg + j) + dx, i.e. 16 CONSECUTIVE bytes of the even half (dx = 0), the odd half (dx = 1) or the even
    // half one byte on (dx = 2): two 8-byte reads + one dword, aligned, and a funnel shift -- not 16 byte gathers.
    extern __shared__ __attribute__((aligned(16))) int8_t s_x[];  // (2P+1) planes x G rows x G
    __shared__ int red[kAcRow];
    int b, z0, z1;
    const bool live = sample_plane_group(B, O1, P, b, z0, z1);
    if (!live) return;
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int m = lane & 15, kq = lane >> 4, H = G >> 1;
    const int8_t *in = grid_i8 + (rows ? rows[b] : (int64_t)b) * row_stride + (size_t)(2 * z0) * G * G;
    const int nload = (2 * (z1 - z0) + 1) * G * G / 16, per_row = G / 16;
    for (int i = threadIdx.x; i < nload; i += kAcThreads) {
        const uint4 v = reinterpret_cast<const uint4 *>(in)[i];
        const int row = i / per_row, c = i - row * per_row;
        uint2 ev, od;
        ev.x = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u);
        ev.y = __builtin_amdgcn_perm(v.w, v.z, 0x06040200u);
        od.x = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);
        od.y = __builtin_amdgcn_perm(v.w, v.z, 0x07050301u);
        *reinterpret_cast<uint2 *>(s_x + row * G + 8 * c) = ev;
        *reinterpret_cast<uint2 *>(s_x + row * G + H + 8 * c) = od;
    }
    for (int i = threadIdx.x; i < kAcRow; i += kAcThreads) red[i] = 0;
    // this lane's two taps: byte offset of the tap's (plane, row, parity half) inside a 3-plane window and its funnel
    // shift (dx = 2: one byte); tap 27 = ones, taps 28..31 = zeros
    const int t1 = 16 + m;
    const bool real1 = t1 < kTaps;
    auto tap_off = [&](int t) { return ((t / 9) * G + (t / 3) % 3) * G + (t % 3 == 1 ? H : 0); };
    const int off0 = tap_off(m), off1 = real1 ? tap_off(t1) : 0;
    const uint32_t sh0 = (m % 3 == 2) ? 1u : 0u, sh1 = (real1 && t1 % 3 == 2) ? 1u : 0u;
    const uint32_t ones1 = t1 == kTaps ? 0x01010101u : 0u;
    i32x4 r00 = {0, 0, 0, 0}, r01 = {0, 0, 0, 0}, r11 = {0, 0, 0, 0};
    // units = (output row y, group of 16 positions along x); a k-step takes 4 consecutive units (k-group kq each)
    const int nxg = (O1 + 15) / 16, nunit = O1 * nxg, nstep = (nunit + 3) / 4, nitem = (z1 - z0) * nstep;
    __syncthreads();
    for (int it = wv; it < nitem; it += kAcWaves) {
        const int zl = it / nstep, st = it - zl * nstep;
        const int u = min(4 * st + kq, nunit - 1), y = u / nxg, xg = u - y * nxg;
        const bool uok = 4 * st + kq < nunit;
        const int8_t *p = s_x + (2 * zl * G + 2 * y) * G + 16 * xg;
        const uint2 a = *reinterpret_cast<const uint2 *>(p + off0), c = *reinterpret_cast<const uint2 *>(p + off0 + 8);
        const uint32_t e = *reinterpret_cast<const uint32_t *>(p + off0 + 16);
        const uint2 a2 = *reinterpret_cast<const uint2 *>(p + off1), c2 = *reinterpret_cast<const uint2 *>(p + off1 + 8);
        const uint32_t e2 = *reinterpret_cast<const uint32_t *>(p + off1 + 16);
        const uint32_t d0[5] = {a.x, a.y, c.x, c.y, e}, d1[5] = {a2.x, a2.y, c2.x, c2.y, e2};
        i32x4 a0, a1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // bytes of positions x = 16 xg + 4q + k >= O1 (and of units past the plane) are zero in every tap
            const int nv = uok ? min(max(O1 - 16 * xg - 4 * q, 0), 4) : 0;
            const uint32_t msk = nv >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nv)) - 1u);
            a0[q] = (int)(__builtin_amdgcn_alignbyte(d0[q + 1], d0[q], sh0) & msk);
            a1[q] = (int)((real1 ? __builtin_amdgcn_alignbyte(d1[q + 1], d1[q], sh1) : ones1) & msk);
        }
        r00 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, a0, r00, 0, 0, 0);
        r01 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, a1, r01, 0, 0, 0);
        r11 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, a1, r11, 0, 0, 0);
    }
    // workgroup sum in LDS (integers), then one atomic per non-zero element
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = (4 * kq + r) * 16 + m;  // D[i = 4kq + r][j = m]
        atomicAdd(&red[o], r00[r]);
        atomicAdd(&red[256 + o], r01[r]);
        atomicAdd(&red[512 + o], r11[r]);
    }
    __syncthreads();
    int *dst = out + (size_t)b * out_row_stride;
    for (int i = threadIdx.x; i < kAcRow; i += kAcThreads)
        if (red[i] != 0) atomicAdd(&dst[i], red[i]);
}

// sum over the minibatch's autocorrelation rows (wave w sums rows w, w + 16, ...: 16-byte requests -- the vector-memory
// pipe of the one CU these single-workgroup kernels run on takes ~16 cycles per wave request whatever its width -- row
// index wave-uniform; integers, so the order does not matter) or, with nrows == 0, the minibatch total at ac.
// 1024 threads; result in Ri[kAcRow] after the trailing barrier.
__device__ __forceinline__ void gather_autocorr(const int *__restrict__ ac, int64_t ac_row_stride, const int64_t *__restrict__ rows, int nrows,
                                                int *Ri)
{
    for (int i = threadIdx.x; i < kAcRow; i += 1024) Ri[i] = nrows == 0 ? ac[i] : 0;
    __syncthreads();
    if (nrows > 0) {
        const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
        int4 t[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        // a wave's rows four at a time: the four row numbers, then the twelve 16-byte requests, are in flight together (one row per
        // iteration was a chain of nrows / 16 x 2 dependent round trips: 13 us of k_bn1_analytic on the update's critical path at
        // 128 rows; integer sums: the order of the additions does not matter)
        for (int b0 = wv; b0 < nrows; b0 += 64) {
            int64_t rr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = min(b0 + 16 * u, nrows - 1);  // (clamped duplicate, masked below)
                rr[u] = rows ? rows[b] : (int64_t)b;
            }
            int4 v[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 3; ++k) v[u][k] = reinterpret_cast<const int4 *>(ac + rr[u] * ac_row_stride)[lane + 64 * k];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (b0 + 16 * u < nrows) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        t[k].x += v[u][k].x; t[k].y += v[u][k].y; t[k].z += v[u][k].z; t[k].w += v[u][k].w;
                    }
                }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = 4 * (lane + 64 * k);
            atomicAdd(&Ri[c], t[k].x);
            atomicAdd(&Ri[c + 1], t[k].y);
            atomicAdd(&Ri[c + 2], t[k].z);
            atomicAdd(&Ri[c + 3], t[k].w);
        }
        __syncthreads();
    }
}
__device__ __forceinline__ int ac_index(int t, int u)  // symmetric R[32][32]; stored tiles: (0,0), (0,1), (1,1)
{
    if ((t >> 4) > (u >> 4)) { const int v = t; t = u; u = v; }
    return ((t >> 4) + (u >> 4)) * 256 + (t & 15) * 16 + (u & 15);
}

// BatchNorm-1 batch statistics WITHOUT the activations: y1 = W1 x + b1 is linear in the input patches, so
//   sum_pos y1[c]   = W1[c] . T2 + M b1[c]
//   sum_pos y1[c]^2 = W1[c]^T R W1[c] + 2 b1[c] W1[c] . T2 + M b1[c]^2
// with R / T2 / M from the input autocorrelation (exact integers) -- known BEFORE conv1 runs, so conv1 can apply
// BN + ReLU in its epilogue.  fp64; same finalize (running statistics, scale / shift / mean / rstd) as the measured path.
__global__ __launch_bounds__(1024) void k_bn1_analytic(const int *__restrict__ ac, int64_t ac_row_stride, const int64_t *__restrict__ rows, int nrows,
                                                      const float *__restrict__ W1 /*[16][27]*/, const float *__restrict__ b1,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float momentum,
                                                      float *__restrict__ running_mean, float *__restrict__ running_var,
                                                      int64_t *__restrict__ num_batches_tracked, const int *__restrict__ skip_flag,
                                                      float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean_out,
                                                      float *__restrict__ rstd_out, int *__restrict__ total_out /*[kAcRow]: saved for backward*/,
                                                      const int *__restrict__ ac_global /*NULL, or [kAcRow]: total over ALL replicas (statistics)*/,
                                                      const float *__restrict__ W2 = nullptr, float *__restrict__ w2img = nullptr,
                                                      int *__restrict__ range_flag = nullptr)
{
    if (blockIdx.x > 0) {  // extra workgroups: the conv2 kernels' f16 weight images (when no conv1 kernel follows to write them in passing)
        prep_w2_split_items<true>(W2, w2img, (blockIdx.x - 1) * blockDim.x + threadIdx.x, (gridDim.x - 1) * blockDim.x);
        return;
    }
    __shared__ int Ri[kAcRow];
    __shared__ double q[kC][kTaps];
    __shared__ float w1s[kC * kTaps];  // (staged: the loops below would otherwise chain 27 global loads per thread)
    if (threadIdx.x < kC * kTaps) w1s[threadIdx.x] = W1[threadIdx.x];
    gather_autocorr(ac, ac_row_stride, rows, nrows, Ri);
    if (total_out != nullptr)
        for (int i = threadIdx.x; i < kAcRow; i += 1024) total_out[i] = Ri[i];  // (this replica's rows: the backward's local sums)
    if (ac_global != nullptr) {  // data-parallel: the batch statistics are those of the global minibatch
        __syncthreads();
        for (int i = threadIdx.x; i < kAcRow; i += 1024) Ri[i] = ac_global[i];
        __syncthreads();
    }
    if (threadIdx.x < kC * kTaps) {  // q[c][t] = W1[c][t] * sum_u W1[c][u] R[t][u]
        const int c = threadIdx.x / kTaps, t = threadIdx.x - c * kTaps;
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < kTaps; ++u) a += (double)w1s[c * kTaps + u] * (double)Ri[ac_index(t, u)];
        q[c][t] = (double)w1s[c * kTaps + t] * a;
    }
    __syncthreads();
    if (threadIdx.x < kC) {
        const int c = threadIdx.x;
        const double count = (double)Ri[ac_index(kTaps, kTaps)], bb = (double)b1[c];
        double wt = 0.0, quad = 0.0;
#pragma unroll
        for (int t = 0; t < kTaps; ++t) {
            wt += (double)w1s[c * kTaps + t] * (double)Ri[ac_index(t, kTaps)];
            quad += q[c][t];
        }
        bn_finalize_channel(c, wt + count * bb, quad + 2.0 * bb * wt + count * bb * bb, count, gamma, beta, eps, momentum, 1, running_mean,
                            running_var, num_batches_tracked, skip_flag, scale, shift, mean_out, rstd_out);
        l1_range_guard(c, W1, b1, scale, shift, range_flag);
    }
}

// dW1, db1 and the BN affine gradients from the fused kernel's sums and the input autocorrelation (fp64).
// One workgroup of 1024 threads.  ac: per-sample autocorrelation rows (gather_autocorr) or, with nrows == 0, the
// minibatch total.
__device__ __forceinline__ void c1w_fused_finish_body(const double *__restrict__ tmp /*[slices][kE1F]*/, int slices, const int *__restrict__ ac,
                                                          int64_t ac_row_stride, const int64_t *__restrict__ rows, int nrows,
                                                          const float *__restrict__ W1 /*[16][27]*/, const float *__restrict__ scale1,
                                                          const float *__restrict__ rstd1,
                                                          const float *__restrict__ beta1, float *__restrict__ dW1, float *__restrict__ db1,
                                                          const double *__restrict__ S2 /*BN2 sums*/, float *g1w, float *g1b, float *g2w,
                                                          float *g2b, const double *__restrict__ S1g /*NULL, or [2][16]: BN1-backward sums over ALL replicas*/,
                                                          const int *__restrict__ ac_global /*NULL, or the global autocorrelation total*/)
{
    __shared__ int Ri[kAcRow];
    __shared__ double R[kAcRow];
    gather_autocorr(ac, ac_row_stride, rows, nrows, Ri);
    for (int i = threadIdx.x; i < kAcRow; i += 1024) R[i] = (double)Ri[i];
    __syncthreads();
    // the <= 64 slices of the kE1F sums, added in slice order by one thread per sum with eight requests in flight (every thread
    // walking its three sums slice by slice was a chain of `slices` dependent round trips: 17 us on the update's critical path)
    __shared__ double ssum[kE1F];
    __shared__ float w1s[kC * kTaps];
    __shared__ double t2g[kTaps];
    if (threadIdx.x < kC * kTaps) w1s[threadIdx.x] = W1[threadIdx.x];
    if (threadIdx.x >= 992 && threadIdx.x < 992 + kTaps) {
        const int u = threadIdx.x - 992;
        t2g[u] = ac_global ? (double)ac_global[ac_index(u, kTaps)] : R[ac_index(u, kTaps)];
    }
    if (threadIdx.x < kE1F) {
        double a = 0.0;
        for (int sl0 = 0; sl0 < slices; sl0 += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = tmp[(size_t)min(sl0 + u, slices - 1) * kE1F + threadIdx.x];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (sl0 + u < slices) a += v[u];
        }
        ssum[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x >= 512) return;
    const int i = threadIdx.x;
    const int tap = i >> 4, co = i & 15;
    double T1 = ssum[i], s1 = ssum[512 + co], s2 = ssum[512 + kC + co];
    auto Rs = [&](int t, int u) { return R[ac_index(t, u)]; };
    // Data-parallel replicas: BatchNorm-1's mean and the two backward means are those of the GLOBAL minibatch (count Mg, column
    // sums T2g, sums S1g), while T1, R and T2 stay this replica's: the gradient all-reduce adds the replicas' dW1.
    const double count = ac_global ? (double)ac_global[ac_index(kTaps, kTaps)] : Rs(kTaps, kTaps), inv_count = 1.0 / count;
    const double s1m = (S1g ? S1g[co] : s1) * inv_count, s2m = (S1g ? S1g[kC + co] : s2) * inv_count;
    if (tap < kTaps) {
        const double T2 = Rs(tap, kTaps);
        double cw = 0.0;  // sum_tap' W1[co][tap'] * (R[tap][tap'] - T2[tap] T2g[tap'] / Mg)
        // (W1 and the global column sums come from LDS and the loop is unrolled by three only: fully unrolled, the 27 hoisted global
        // loads and index computations spilled 144 bytes per thread of this 1024-thread workgroup)
#pragma unroll 3
        for (int u = 0; u < kTaps; ++u) cw += (double)w1s[co * kTaps + u] * (Rs(tap, u) - T2 * t2g[u] * inv_count);
        const double T3 = (double)rstd1[co] * cw;
        dW1[co * kTaps + tap] = (float)((double)scale1[co] * (T1 - s1m * T2 - s2m * T3));
    }
    if (tap == 0) {
        db1[co] = 0.0f;
        g1b[co] = (float)s1;
        g1w[co] = (float)s2;
        g2b[co] = (float)S2[co];
        g2w[co] = (float)S2[kC + co];
    }
}

__global__ __launch_bounds__(1024) void k_c1w_fused_finish(const double *__restrict__ tmp /*[slices][kE1F]*/, int slices, const int *__restrict__ ac,
                                                          int64_t ac_row_stride, const int64_t *__restrict__ rows, int nrows,
                                                          const float *__restrict__ W1 /*[16][27]*/, const float *__restrict__ scale1,
                                                          const float *__restrict__ rstd1,
                                                          const float *__restrict__ beta1, float *__restrict__ dW1, float *__restrict__ db1,
                                                          const double *__restrict__ S2 /*BN2 sums*/, float *g1w, float *g1b, float *g2w,
                                                          float *g2b, const double *__restrict__ S1g /*NULL, or [2][16]: BN1-backward sums over ALL replicas*/,
                                                          const int *__restrict__ ac_global /*NULL, or the global autocorrelation total*/)
{
    c1w_fused_finish_body(tmp, slices, ac, ac_row_stride, rows, nrows, W1, scale1, rstd1, beta1, dW1, db1, S2, g1w, g1b, g2w, g2b, S1g, ac_global);
}

// ---------------------------------------------------------------------------
// conv1 weight gradient with BN1 backward fused into the operand load:
//   dy1 = scale1 * (dz1' - S1/M - xhat * S2/M);  dW1[co][tap] = sum_pos in[inpos(pos,tap)] * dy1[pos, co]
// MFMA: i = tap (two 16-row tiles), j = co, k = 4 consecutive output positions along x.
// ---------------------------------------------------------------------------
template <typename A, typename IN = float>
__global__ __launch_bounds__(kEncThreads) void k_conv1_wgrad(
    const IN *__restrict__ obs_base, const int64_t *__restrict__ rows, int64_t row_stride, const typename A::T *__restrict__ dz1p,
    const typename A::T *__restrict__ y1, const float *__restrict__ scale1, const float *__restrict__ mean1,
    const float *__restrict__ rstd1, const double *__restrict__ S /*[2][16]*/, double count, int B, int G, int O1,
    float *__restrict__ partial /*[nwaves][2*256 + 16]*/)
{
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform -> SALU index math
    const int n = lane & 15, kq = lane >> 4;
    const int wave_global = blockIdx.x * kEncWaves + wv, nwaves = gridDim.x * kEncWaves;
    const float sc = scale1[n], mu = mean1[n], rs = rstd1[n];
    const float m1 = (float)(S[n] / count), m2 = (float)(S[kC + n] / count);
    int off[2];
    bool tok[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int t = 16 * tt + n;  // A row i = tap (lane & 15)
        tok[tt] = t < kTaps;
        const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
        off[tt] = tok[tt] ? (dz * G + dy) * G + dx : 0;
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    const int nrows = B * O1 * O1;
    int row0, row1;
    wave_row_range(nrows, row0, row1);
    (void)nwaves;
    for (int row = row0; row < row1; ++row) {
        const int b = row / (O1 * O1), rem = row - b * O1 * O1, oz = rem / O1, oy = rem - oz * O1;
        const IN *in = obs_base + (rows ? rows[b] : (int64_t)b) * row_stride + ((size_t)(2 * oz) * G + 2 * oy) * G;
        for (int x0 = 0; x0 < O1; x0 += 32) {
            // eight x-groups per trip: all 32 loads are issued before the first MFMA
            float g4[8], y4[8], a04[8], a14[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int x = x0 + 4 * u + kq;
                const int xc = x < O1 ? x : O1 - 1;
                const size_t idx = vox1(b, oz, oy, xc, O1) * kC + n;
                g4[u] = A::ld1(dz1p + idx);
                y4[u] = A::ld1(y1 + idx);
                a04[u] = tok[0] ? (float)in[2 * xc + off[0]] : 0.0f;  // A[i = tap][k = pos]
                a14[u] = tok[1] ? (float)in[2 * xc + off[1]] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = x0 + 4 * u + kq < O1;
                float dy = sc * (g4[u] - m1 - ((y4[u] - mu) * rs) * m2);  // B[k = pos][j = co = n]
                dy = ok ? dy : 0.0f;
                bsum += dy;
                acc0 = mfma4(a04[u], dy, acc0);
                acc1 = mfma4(a14[u], dy, acc1);
            }
        }
    }
    __shared__ float red[2 * 256 + kC];
    bsum = kgroup_sum(bsum);
    for (int w = 0; w < kEncWaves; ++w) {
        if (wv == w) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o0 = (4 * kq + r) * kC + n, o1 = 256 + o0;   // [tap 0..15][co], [tap 16..31][co]
                red[o0] = (w == 0 ? 0.0f : red[o0]) + acc0[r];
                red[o1] = (w == 0 ? 0.0f : red[o1]) + acc1[r];
            }
            if (lane < kC) red[512 + lane] = (w == 0 ? 0.0f : red[512 + lane]) + bsum;
        }
        __syncthreads();
    }
    float *out = partial + (size_t)blockIdx.x * (2 * 256 + kC);
    for (int o = threadIdx.x; o < 2 * 256 + kC; o += kEncThreads) out[o] = red[o];
    (void)wave_global;
}

// LDS-staged variant (used when G % 4 == 0).  The direct kernel issues four 4-byte-per-lane loads per
// MFMA pair (256 B per wave-load): the vector-memory pipe processes a wave-load in ~16 cycles whatever
// its width, so those narrow loads -- not HBM bytes, not MFMA -- set the kernel's time.  Here every
// output row is staged with full-width loads (16 B/lane, 1 KiB per wave-load): the dz1'/y1 row (BN1
// backward applied on the way in) and the 9 input rows of its receptive field; the MFMA operands
// are then 4-byte LDS reads.  7 wide loads per row instead of 32 narrow ones.  Each wave owns a
// private LDS region (no workgroup barrier: rows per wave differ at the tail).
template <typename A, int NR, int NI, typename IN>
__global__ __launch_bounds__(kEncThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_conv1_wgrad_lds(
    const IN *__restrict__ obs_base, const int64_t *__restrict__ rows, int64_t row_stride, const typename A::T *__restrict__ dz1p,
    const typename A::T *__restrict__ y1, const float *__restrict__ scale1, const float *__restrict__ mean1,
    const float *__restrict__ rstd1, const double *__restrict__ S /*[2][16]*/, double count, int B, int G, int O1,
    float *__restrict__ partial /*[nwaves][2*256 + 16]*/)
{
    // NR / NI = 16-byte requests per lane that cover one y1 row (rowlen floats) / the 9 input rows
    // (9 G floats).  All 2 NR + NI requests of row r+1 are issued (clamped addresses, unconditional)
    // before the MFMA phase of row r and land in LDS after it: one exposed latency per wave, not one
    // per request as in the first version of this kernel (2 + 1 + 1 + 1 dependent round trips a row).
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform -> SALU index math
    const int n = lane & 15, kq = lane >> 4;
    const int wave_global = blockIdx.x * kEncWaves + wv;
    const int XH = (O1 + 1) >> 1, rowlen = 2 * XH * kC;  // floats of one (b, z, y) row of y1 / dz1'
    float *s_dy = lds + (size_t)wv * (rowlen + 9 * G), *s_in = s_dy + rowlen;
    // staging role of this lane: channels c4 .. c4+3 of the voxels it copies
    const int c4 = 4 * (lane & 3);
    float sc4[4], mu4[4], rs4[4], m14[4], m24[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc4[k] = scale1[c4 + k]; mu4[k] = mean1[c4 + k]; rs4[k] = rstd1[c4 + k];
        m14[k] = (float)(S[c4 + k] / count); m24[k] = (float)(S[kC + c4 + k] / count);
    }
    int offl[2];
    bool tok[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int t = 16 * tt + n;  // A row i = tap
        tok[tt] = t < kTaps;
        const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
        offl[tt] = tok[tt] ? (dz * 3 + dy) * G + dx : 0;
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    const int nrows = B * O1 * O1;
    int row0, row1;
    wave_row_range(nrows, row0, row1);
    // int8 grid: the 9 input rows are 9 G bytes = one 16-byte request for the first 9 G / 16 lanes (G % 16 == 0)
    constexpr bool kI8 = sizeof(IN) == 1;
    float4 rg[NR], ry[NR], ri[NI];
    uint4 ri8 = make_uint4(0, 0, 0, 0);
    auto request = [&](int row, float4 (&rg)[NR], float4 (&ry)[NR], float4 (&ri)[NI]) {
        const int b = row / (O1 * O1), rem = row - b * O1 * O1, oz = rem / O1, oy = rem - oz * O1;
        const uint32_t rb = vox1(b, oz, oy, 0, O1) * kC;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = min(lane * 4 + u * kWave * 4, rowlen - 4);
            rg[u] = A::ld4(dz1p + rb + i);
            ry[u] = A::ld4(y1 + rb + i);
        }
        const IN *in = obs_base + (rows ? rows[b] : (int64_t)b) * row_stride;
        if constexpr (kI8) {
            const int i = min(lane * 16, 9 * G - 16);
            const int dz = i / (3 * G), r = i - dz * 3 * G;  // rows dy = 0..2 of one dz are contiguous
            ri8 = *reinterpret_cast<const uint4 *>(reinterpret_cast<const int8_t *>(in) + ((size_t)(2 * oz + dz) * G + 2 * oy) * G + r);
        } else {
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int i = min(lane * 4 + u * kWave * 4, 9 * G - 4);
                const int dz = i / (3 * G), r = i - dz * 3 * G;  // rows dy = 0..2 of one dz are contiguous
                ri[u] = ActF32::ld4(reinterpret_cast<const float *>(in) + ((size_t)(2 * oz + dz) * G + 2 * oy) * G + r);
            }
        }
    };
    if (row0 < row1) request(row0, rg, ry, ri);
    for (int row = row0; row < row1; ++row) {
        // ---- stage dy1 = BN1-backward(dz1', y1) of the row and its 9 input rows ----
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = lane * 4 + u * kWave * 4;
            if (i < rowlen) {
                const float4 g = rg[u], y = ry[u];
                const int v = i >> 4, plane = v >= XH ? 1 : 0, x = 2 * (v - plane * XH) + plane;
                float4 d;
                d.x = sc4[0] * (g.x - m14[0] - ((y.x - mu4[0]) * rs4[0]) * m24[0]);
                d.y = sc4[1] * (g.y - m14[1] - ((y.y - mu4[1]) * rs4[1]) * m24[1]);
                d.z = sc4[2] * (g.z - m14[2] - ((y.z - mu4[2]) * rs4[2]) * m24[2]);
                d.w = sc4[3] * (g.w - m14[3] - ((y.w - mu4[3]) * rs4[3]) * m24[3]);
                if (x >= O1) d = make_float4(0.f, 0.f, 0.f, 0.f);  // padding slot of the odd plane
                bs[0] += d.x; bs[1] += d.y; bs[2] += d.z; bs[3] += d.w;
                *reinterpret_cast<float4 *>(s_dy + i) = d;
            }
        }
        if constexpr (kI8) {
            if (lane * 16 < 9 * G) {
                const uint32_t wd[4] = {ri8.x, ri8.y, ri8.z, ri8.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(s_in + lane * 16 + 4 * q) =
                        make_float4((float)(int8_t)(wd[q] & 255u), (float)(int8_t)((wd[q] >> 8) & 255u),
                                    (float)(int8_t)((wd[q] >> 16) & 255u), (float)(int8_t)(wd[q] >> 24));
            }
        } else {
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int i = lane * 4 + u * kWave * 4;
                if (i < 9 * G) *reinterpret_cast<float4 *>(s_in + i) = ri[u];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (row + 1 < row1) request(row + 1, rg, ry, ri);
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA: i = tap, j = co, k = 4 consecutive output positions ----
        for (int x0 = 0; x0 < O1; x0 += 4) {
            const int pos = x0 + kq;
            const bool ok = pos < O1;
            const int pc = ok ? pos : 0;
            const float bv = ok ? s_dy[((pc & 1) * XH + (pc >> 1)) * kC + n] : 0.0f;  // B[k = pos][j = co]
            const float a0 = tok[0] ? s_in[offl[0] + 2 * pc] : 0.0f;                  // A[i = tap][k = pos]
            const float a1 = tok[1] ? s_in[offl[1] + 2 * pc] : 0.0f;
            acc0 = mfma4(a0, bv, acc0);
            acc1 = mfma4(a1, bv, acc1);
        }
        __builtin_amdgcn_wave_barrier();  // next row overwrites the staging area
    }
    // bias sums: lanes with equal (lane & 3) hold the same channel group
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int d = 4; d < kWave; d <<= 1) bs[k] += __shfl_xor(bs[k], d, kWave);
    // workgroup-level sum (wave order) in the staging LDS, one partial row per workgroup
    __syncthreads();  // every wave is done with its staging area
    float *red = lds;
    for (int w = 0; w < kEncWaves; ++w) {
        if (wv == w) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o0 = (4 * kq + r) * kC + n, o1 = 256 + o0;   // [tap 0..15][co], [tap 16..31][co]
                red[o0] = (w == 0 ? 0.0f : red[o0]) + acc0[r];
                red[o1] = (w == 0 ? 0.0f : red[o1]) + acc1[r];
            }
            if (lane < 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) red[512 + 4 * lane + k] = (w == 0 ? 0.0f : red[512 + 4 * lane + k]) + bs[k];
            }
        }
        __syncthreads();
    }
    float *out = partial + (size_t)blockIdx.x * (2 * 256 + kC);
    for (int o = threadIdx.x; o < 2 * 256 + kC; o += kEncThreads) out[o] = red[o];
    (void)wave_global;
}

// (+ the BN affine gradients: d beta = S[0], d gamma = S[1] of each layer's backward sums)
__global__ void k_conv1_wgrad_finish(const double *__restrict__ tmp /*[slices][E]*/, int slices, float *__restrict__ dW1,
                                     float *__restrict__ db1, const double *__restrict__ S1, const double *__restrict__ S2, float *g1w,
                                     float *g1b, float *g2w, float *g2b)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, E = 512 + kC;
    if (i < kC) {
        g1b[i] = (float)S1[i];
        g1w[i] = (float)S1[kC + i];
        g2b[i] = (float)S2[i];
        g2w[i] = (float)S2[kC + i];
    }
    if (i >= E) return;
    double t = 0.0;
#pragma unroll 8
    for (int sl = 0; sl < slices; ++sl) t += tmp[(size_t)sl * E + i];
    if (i < 512) {
        const int tap = i >> 4, co = i & 15;
        if (tap < kTaps) dW1[co * kTaps + tap] = (float)t;
    } else {
        db1[i - 512] = (float)t;
    }
}

// ===========================================================================
// C-ABI
// ===========================================================================
#include "conv_split.h"
#include "conv_splitx.h"

static inline int out_size(int g) { return (g - 3) / 2 + 1; }

GNBV_API size_t gnbv_encoder_y1_elems(int batch, int grid)
{
    if (batch <= 0 || grid < 7) return 0;
    return y1_elems(batch, (grid - 3) / 2 + 1);
}

GNBV_API size_t gnbv_encoder_workspace_bytes(int batch, int grid)
{
    if (batch <= 0 || grid < 7) return 0;
    const int o1 = out_size(grid);
    // BN partials of the largest producer (conv1 fwd / conv2 dgrad: B*O1 workgroups x 4 waves x 32 floats),
    // weight-gradient partials (kWgradWaves x 6928 floats), fp64 reduction scratch
    const size_t bn = (size_t)(batch + 8) * o1 * kEncWaves * 2 * kC * sizeof(float);
    const size_t wg = (size_t)2048 * (kTaps * 256 + kC) * sizeof(float);
    return bn + wg + (8192 + (size_t)64 * (kTaps * 256 + kC)) * sizeof(double) + 2 * kTaps * 256 * sizeof(float) +
           2 * 14 * 2 * 64 * 16 /*split f16 weight images*/ + 1024 /*their |W2| bounds*/ + 8192 + 4096;
}

struct EncWs {
    float *bn_part, *wg_part;
    double *red, *tmp;
    float *w2img;  // 2 x 6912 floats
    uint4 *w2split;  // 2 x split::kW2ImgU4 (conv_split.h)
};
static inline EncWs enc_carve(void *ws, int batch, int grid)
{
    EncWs w;
    const int o1 = out_size(grid);
    w.bn_part = (float *)ws;
    w.wg_part = w.bn_part + (size_t)(batch + 8) * o1 * kEncWaves * 2 * kC;
    w.red = (double *)(((uintptr_t)(w.wg_part + (size_t)2048 * (kTaps * 256 + kC)) + 255) & ~(uintptr_t)255);
    w.tmp = w.red + 8192;
    w.w2img = (float *)(w.tmp + (size_t)64 * (kTaps * 256 + kC));
    w.w2split = (uint4 *)(w.w2img + 2 * kTaps * 256);
    return w;
}

constexpr int kReduceSlices = 64;

// stage 1 of the deterministic fp64 reduction: partial [P][E] -> tmp [slices][E]; the consumer (a
// *_finish kernel) adds the <= 64 slices in order.  Returns the number of slices.
static inline int reduce_slices(int P)
{
    const int slices = P / 32;  // ~32 partial rows per workgroup
    return slices < 1 ? 1 : (slices > kReduceSlices ? kReduceSlices : slices);
}
static inline int reduce_stage1(const float *partial, int P, int E, double *tmp, hipStream_t st)
{
    const int slices = reduce_slices(P);
    const int per = (P + slices - 1) / slices;
    if (E % 4 == 0 && (((uintptr_t)partial) & 15) == 0)
        hipLaunchKernelGGL(k_reduce_partials4, dim3((E / 4 + 63) / 64, slices), dim3(256), 0, st, partial, P, E, per, tmp);
    else
        hipLaunchKernelGGL(k_reduce_partials<float>, dim3((E + 63) / 64, slices), dim3(256), 0, st, partial, P, E, per, tmp,
                           (float *)nullptr);
    return slices;
}
// Kernel-path predicates shared by forward and backward (they must agree on what the layer-1 buffer holds).
//   fused_path: conv2 data gradient fused with the conv1 weight gradient (needs the grid as aligned int8 rows)
static inline bool env_off(const char *name)
{
    const char *e = getenv(name);
    return e && e[0] == '0';
}
// conv2 kernels on the f16 matrix pipe with split (hi + lo) operands, staged through LDS (conv_split.h): fp32 y1 in the default
// x-parity layout, 16 voxel slots per half row (G = 64)
static inline bool conv_split_path(const GnbvEncoderParams *p, int grid)
{
    const int O1 = out_size(grid), O2 = out_size(O1);
    return !env_off("GENNBV_CONV_SPLIT") && !p->force_fp32 && (O1 + 1) / 2 == 16 && O2 <= 15;
}
// the same arithmetic for rows wider than 16 voxel slots per x parity (conv_splitx.h: x tiles of 16 outputs, 17-voxel ring rows): the
// G = 128 class (O1 = 63, O2 = 31); GENNBV_SPLITX=1 also routes the G = 64 class through these kernels (tests)
static inline bool conv_splitx_path(const GnbvEncoderParams *p, int grid)
{
    const int O1 = out_size(grid), O2 = out_size(O1), XH = (O1 + 1) / 2;
    if (env_off("GENNBV_CONV_SPLIT") || p->force_fp32 || O2 < 1) return false;
    const char *e = getenv("GENNBV_SPLITX");
    if (e && e[0] == '0') return false;
    return (XH > 16 && XH <= 32 && O2 <= 32) || (XH == 16 && O2 <= 15 && e && e[0] == '1');
}
// most workgroups of the x-tiled backward kernels (each walks items block, block + grid, ...); GENNBV_SPLITX_MAXWG lowers it so that small
// test shapes exercise the several-items-per-workgroup path
static inline int splitx_max_wg()
{
    const char *e = getenv("GENNBV_SPLITX_MAXWG");
    const int v = e ? atoi(e) : 0;
    return v >= 8 && v <= 512 ? (v & ~7) : 512;
}
// most workgroups of the G = 64 split kernels that walk several items (round 6: k_conv2_wgrad_split_dma): 256 = one per CU, i.e. two items
// per workgroup at the bench's minibatch instead of two workgroups per CU one after the other; GENNBV_CONV_MAXWG=512 restores one item per
// workgroup (bit-identity tests against the register-staged kernel, A/B runs)
static inline int conv_max_wg()
{
    const char *e = getenv("GENNBV_CONV_MAXWG");
    const int v = e ? atoi(e) : 0;
    return v >= 8 && v <= 512 ? (v & ~7) : 256;
}
// conv2 weight gradient with the LDS-DMA transport (k_conv2_wgrad_split_dma, round 6): the default at G = 64 -- bit-identical to the
// register-staged k_conv2_wgrad_split, which GENNBV_WGRAD_DMA=0 keeps selectable (A/B runs, tests/test_encoder_gpu.py)
static inline bool wgrad_dma_path()
{
    const char *e = getenv("GENNBV_WGRAD_DMA");
    return !(e && e[0] == '0');
}
static inline bool fused_path(const GnbvEncoderParams *p, int grid)
{
    return !env_off("GENNBV_FUSED_BWD") && p->grid_i8 != nullptr && grid % 16 == 0 && 3 * grid * grid <= 64 * 1024 &&
           p->grid_i8_row_stride % 16 == 0 && (((uintptr_t)p->grid_i8 & 15) == 0);
}
static inline bool conv1_i8_staged(const GnbvEncoderParams *p, int grid)
{
    const int O1 = out_size(grid);
    return p->grid_i8 != nullptr && grid % 16 == 0 && p->grid_i8_row_stride % 16 == 0 && (((uintptr_t)p->grid_i8 & 15) == 0) &&
           (size_t)3 * (2 * O1 + 1) * grid * sizeof(float) <= 64 * 1024 && 2 * O1 + 1 <= grid;
}

// bn_state = [2][4][16] floats (scale, shift, mean, rstd per layer) + kAcRow ints: the minibatch total of the input
// autocorrelation, written by the forward whenever it computed BN1's statistics from it (analytic_bn1) and read
// back by the backward instead of gathering the rows again
constexpr int kBnStateFloats = 2 * 4 * kC;
static inline bool analytic_bn1(const GnbvEncoderParams *p, int grid)
{
    return p->autocorr != nullptr && conv1_i8_staged(p, grid) && 3 * grid * grid <= 64 * 1024 && !env_off("GENNBV_ANALYTIC_BN1");
}

// minibatch total of the input autocorrelation into `Rac` (when the caller has no per-row results)
static int launch_autocorr_total(const GnbvEncoderParams *p, const int64_t *rows, int batch, int grid, int *Rac, hipStream_t st)
{
    if (hipMemsetAsync(Rac, 0, kAcRow * sizeof(int), st) != hipSuccess) return (int)hipGetLastError();
    const int P = autocorr_planes(grid), O1 = out_size(grid);
    hipLaunchKernelGGL(k_input_autocorr, dim3(sample_plane_group_grid(batch, O1, P)), dim3(kAcThreads), (size_t)(2 * P + 1) * grid * grid, st,
                       p->grid_i8, rows, p->grid_i8_row_stride, batch, grid, O1, P, Rac, (int64_t)0);
    return gnbv_launch_status();
}

// The launches of an inference forward that depend on the parameters only (GnbvEncoderParams.eval_prepared): BatchNorm-1's
// (scale, shift, mean, rstd) from the running statistics + the bound check of its channels, the conv2 weight images, and -- bn2 --
// BatchNorm-2's.  One-launch conv1 + conv2 inference path (fused_eval below) only.
static int eval_prepare_launches(const GnbvEncoderParams *p, int batch, int O1, int P2, float *bn_state, const EncWs &w, hipStream_t st, bool bn2_too)
{
    float *bn1 = bn_state, *bn2 = bn_state + 4 * kC;
    hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(64), 0, st, (double)batch * O1 * O1 * O1, p->bn1_w, p->bn1_b, p->eps, p->momentum, p->bn1_rm,
                       p->bn1_rv, bn1, bn1 + kC, bn1 + 2 * kC, bn1 + 3 * kC, p->w1, p->b1, p->range_flag);
    hipLaunchKernelGGL(k_prep_w2_split_only, dim3(12), dim3(256), 0, st, p->w2, w.w2img);
    if (bn2_too)
        hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(64), 0, st, (double)batch * P2, p->bn2_w, p->bn2_b, p->eps, p->momentum, p->bn2_rm,
                           p->bn2_rv, bn2, bn2 + kC, bn2 + 2 * kC, bn2 + 3 * kC);
    return gnbv_launch_status();
}

GNBV_API int gnbv_encoder_eval_prepare(int batch, int grid, const GnbvEncoderParams *p, float *bn_state, void *workspace, size_t workspace_bytes,
                                       void *stream)
{
    GNBV_CHECK_ARG(p && bn_state && workspace && batch > 0 && grid >= 7);
    GNBV_CHECK_ARG(workspace_bytes >= gnbv_encoder_workspace_bytes(batch, grid) && ((uintptr_t)workspace & 255) == 0);
    GNBV_CHECK_ARG(p->w1 && p->b1 && p->bn1_w && p->bn1_b && p->bn1_rm && p->bn1_rv && p->w2 && p->b2 && p->bn2_w && p->bn2_b &&
                   p->bn2_rm && p->bn2_rv);
    if (!(conv_split_path(p, grid) && conv1_i8_staged(p, grid) && grid == 64 && !env_off("GENNBV_FUSED_EVAL"))) return GNBV_ERR_NOT_APPLICABLE;
    const int O1 = out_size(grid), O2 = out_size(O1);
    return eval_prepare_launches(p, batch, O1, O2 * O2 * O2, bn_state, enc_carve(workspace, batch, grid), gnbv_stream(stream), /*bn2=*/true);
}

GNBV_API int gnbv_encoder_grid_forward(const float *obs_grid, const int64_t *rows, int64_t row_stride, int batch, int grid,
                                       const GnbvEncoderParams *p, int training, const int *skip_flag, void *y1, float *y2,
                                       float *bn_state /*[2][4][16]: scale, shift, mean, rstd per layer*/, float *features,
                                       void *workspace, size_t workspace_bytes, void *stream)
{
    // obs_grid == NULL: compact observations, the grid exists only as the int8 rows p->grid_i8 (fp32 activations only)
    GNBV_CHECK_ARG(p && (obs_grid || (p->grid_i8)) && y1 && y2 && bn_state && workspace && batch > 0 && grid >= 7);
    GNBV_CHECK_ARG(p->grid_i8 == nullptr || p->grid_i8_row_stride >= (int64_t)grid * grid * grid);
    GNBV_CHECK_ARG(workspace_bytes >= gnbv_encoder_workspace_bytes(batch, grid) && ((uintptr_t)workspace & 255) == 0);
    GNBV_CHECK_ARG(p->w1 && p->b1 && p->bn1_w && p->bn1_b && p->bn1_rm && p->bn1_rv && p->w2 && p->b2 && p->bn2_w && p->bn2_b &&
                   p->bn2_rm && p->bn2_rv);
    hipStream_t st = gnbv_stream(stream);
    const int O1 = out_size(grid), O2 = out_size(O1);
    GNBV_CHECK_ARG(O2 >= 1 && y1_elems(batch, O1) < ((size_t)1 << 31));
    const int P2 = O2 * O2 * O2;
    EncWs w = enc_carve(workspace, batch, grid);
    float *bn1 = bn_state, *bn2 = bn_state + 4 * kC;
    int err;
    GNBV_CHECK_ARG(p->autocorr == nullptr || (p->autocorr_row_stride >= kAcRow && p->autocorr_row_stride % 4 == 0 && ((uintptr_t)p->autocorr & 15) == 0));
    const bool dp = training && p->world > 1 && p->sync_sum != nullptr && p->sync_buf != nullptr;  // BatchNorm over the replicas' global minibatch
    // Inference with the grid as int8 rows at G = 64: conv1 + BN1 + ReLU + conv2 in one kernel, no layer-1 buffer at all
    // (conv_split.h).  y1 is left untouched (no backward follows an eval-mode forward).
    const bool fused_eval = !training && conv_split_path(p, grid) && conv1_i8_staged(p, grid) && grid == 64 && !env_off("GENNBV_FUSED_EVAL");
    bool fused_train = false;
    if (fused_eval) {
        if (!p->eval_prepared && (err = eval_prepare_launches(p, batch, O1, P2, bn_state, w, st, /*bn2=*/false))) return err;
        static bool attr_fe = false;
        if (!attr_fe) {
            const hipError_t e = hipFuncSetAttribute((const void *)k_conv12_fwd_split<false>, hipFuncAttributeMaxDynamicSharedMemorySize, fsplit::kLdsBytes);
            if (e != hipSuccess) return (int)e;
            attr_fe = true;
        }
        hipLaunchKernelGGL(k_conv12_fwd_split<false>, dim3(sample_plane_group_grid(batch, O2, split::kNP)), dim3(split::kThreads), fsplit::kLdsBytes, st,
                           p->grid_i8, rows, p->grid_i8_row_stride, p->w1, p->b1, (const float *)bn1, (const float *)(bn1 + kC), batch, grid, O1, O2,
                           (const uint4 *)w.w2split, p->b2, y2, (float *)nullptr, (float *)nullptr);
        if ((err = gnbv_launch_status())) return err;
    } else {
    // BN1 batch statistics analytically from the stored input autocorrelation rows (k_bn1_analytic: 6 us, independent of
    // conv1) instead of partial sums in conv1 + a 12 us reduction behind it; y1 is stored as before
    const bool analytic = training && analytic_bn1(p, grid);
    if (dp && !(analytic && fused_path(p, grid) && p->autocorr_global != nullptr)) return (int)hipErrorInvalidValue;  // (see GnbvEncoderParams.world)
    // Training with BN1's statistics known beforehand: conv1 + BN1 + ReLU + conv2 as ONE launch that also stores y1 for the backward
    // (conv_split.h) -- conv2 never reads the 244 MB it would otherwise fetch right after conv1 wrote them.
    fused_train = analytic && conv_split_path(p, grid) && grid == 64 && !env_off("GENNBV_CONV1_SPLIT") && !env_off("GENNBV_FUSED_TRAIN");
    if (analytic) {
        // (the caller may hand over the minibatch's autocorrelation total: no gather then -- GnbvEncoderParams.autocorr_total)
        const bool have_total = p->autocorr_total != nullptr && !dp;
        hipLaunchKernelGGL(k_bn1_analytic, dim3(fused_train ? 9 : 1), dim3(1024), 0, st, have_total ? (const int *)p->autocorr_total : (const int *)p->autocorr,
                           p->autocorr_row_stride, rows, have_total ? 0 : batch, p->w1,
                           p->b1, p->bn1_w, p->bn1_b, p->eps, p->momentum, p->bn1_rm, p->bn1_rv, p->bn1_nbt, skip_flag, bn1, bn1 + kC, bn1 + 2 * kC,
                           bn1 + 3 * kC, (int *)(bn_state + kBnStateFloats), dp ? (const int *)p->autocorr_global : (const int *)nullptr,
                           p->w2, w.w2img, p->range_flag);
        if ((err = gnbv_launch_status())) return err;
    }
    float *c1_part = (training && !analytic) ? w.bn_part : nullptr;
    const int split_img = (conv_split_path(p, grid) || conv_splitx_path(p, grid)) ? 1 : 0;  // (the f16 weight images: only where a split kernel reads them)
    // conv1 (+ BN1 statistics)
    if (fused_train) {
        static bool attr_ft = false;
        if (!attr_ft) {
            const hipError_t e = hipFuncSetAttribute((const void *)k_conv12_fwd_split<true>, hipFuncAttributeMaxDynamicSharedMemorySize, fsplit::kLdsBytes);
            if (e != hipSuccess) return (int)e;
            attr_ft = true;
        }
        hipLaunchKernelGGL(k_conv12_fwd_split<true>, dim3(sample_plane_group_grid(batch, O2, split::kNP)), dim3(split::kThreads), fsplit::kLdsBytes, st,
                           p->grid_i8, rows, p->grid_i8_row_stride, p->w1, p->b1, (const float *)bn1, (const float *)(bn1 + kC), batch, grid, O1, O2,
                           (const uint4 *)w.w2split, p->b2, y2, (float *)y1, w.bn_part);
    } else {
        const size_t c1_lds = (size_t)3 * (2 * O1 + 1) * grid * sizeof(float);
        const bool c1_staged = obs_grid != nullptr && (grid % 4 == 0) && (row_stride % 4 == 0) && (((uintptr_t)obs_grid & 15) == 0) && c1_lds <= 64 * 1024 &&
                               2 * O1 + 1 <= grid;
        if (conv1_i8_staged(p, grid) && c1_part == nullptr && grid == 64 && conv_split_path(p, grid) && !env_off("GENNBV_CONV1_SPLIT"))  // (no partial sums: BN1 is analytic or in eval mode)
            hipLaunchKernelGGL(k_conv1_fwd_split, dim3(sample_plane_grid(batch, O1)), dim3(kEncThreads), 0, st, p->grid_i8, rows, p->grid_i8_row_stride, batch, grid,
                               O1, p->w1, p->b1, (float *)y1, p->w2, w.w2img);
        else if (conv1_i8_staged(p, grid))  // compact int8 copy of the tri-class grid: a quarter of the input bytes
            hipLaunchKernelGGL((k_conv1_fwd_lds<ActF32, int8_t>), dim3(sample_plane_grid(batch, O1)), dim3(kEncThreads), c1_lds, st, p->grid_i8, rows,
                               p->grid_i8_row_stride, batch, grid, O1, p->w1, p->b1, (float *)y1, c1_part, p->w2, w.w2img, split_img);
        else if (c1_staged)
            hipLaunchKernelGGL((k_conv1_fwd_lds<ActF32, float>), dim3(sample_plane_grid(batch, O1)), dim3(kEncThreads), c1_lds, st, obs_grid, rows,
                               row_stride, batch, grid, O1, p->w1, p->b1, (float *)y1, c1_part, p->w2, w.w2img, split_img);
        else if (p->grid_i8 != nullptr && grid % 16 == 0 && p->grid_i8_row_stride % 16 == 0 && (((uintptr_t)p->grid_i8 & 15) == 0) &&
                 c1_lds / 4 <= 64 * 1024 && 2 * O1 + 1 <= grid)  // int8 rows, fp32 slab too large (G = 128): int8 slab
            hipLaunchKernelGGL((k_conv1_fwd_lds<ActF32, int8_t, true>), dim3(sample_plane_grid(batch, O1)), dim3(kEncThreads), c1_lds / 4, st, p->grid_i8,
                               rows, p->grid_i8_row_stride, batch, grid, O1, p->w1, p->b1, (float *)y1, c1_part, p->w2, w.w2img, split_img);
        else if (obs_grid == nullptr)  // compact rows at a size the staged kernels do not take
            hipLaunchKernelGGL((k_conv1_fwd<ActF32, int8_t>), dim3(sample_plane_grid(batch, O1)), dim3(kEncThreads), 0, st, p->grid_i8, rows,
                               p->grid_i8_row_stride, batch, grid, O1, p->w1, p->b1, (float *)y1, c1_part, p->w2, w.w2img, split_img);
        else
            hipLaunchKernelGGL(k_conv1_fwd<ActF32>, dim3(sample_plane_grid(batch, O1)), dim3(kEncThreads), 0, st, obs_grid, rows, row_stride,
                               batch, grid, O1, p->w1, p->b1, (float *)y1, c1_part, p->w2, w.w2img, split_img);
    }
    if ((err = gnbv_launch_status())) return err;
    if (analytic) {
    } else if (training)
        hipLaunchKernelGGL(k_stats_reduce, dim3(1), dim3(1024), 0, st, w.bn_part, sample_plane_grid(batch, O1), (double *)nullptr,
                           (double)batch * O1 * O1 * O1, p->bn1_w, p->bn1_b, p->eps, p->momentum, p->bn1_rm, p->bn1_rv, p->bn1_nbt, skip_flag,
                           bn1, bn1 + kC, bn1 + 2 * kC, bn1 + 3 * kC, (const float *)nullptr, (float *)nullptr, p->w1, p->b1, p->range_flag);
    else
        hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(64), 0, st, (double)batch * O1 * O1 * O1, p->bn1_w, p->bn1_b, p->eps, p->momentum,
                           p->bn1_rm, p->bn1_rv, bn1, bn1 + kC, bn1 + 2 * kC, bn1 + 3 * kC, p->w1, p->b1, p->range_flag);
    if ((err = gnbv_launch_status())) return err;
    }
    // conv2 (BN1 + ReLU on load; + BN2 statistics).  Its LDS weight images were written by the conv1 kernel in passing.
    const int pz2 = planes_per_group(batch, O2);
    int g2 = sample_plane_group_grid(batch, O2, pz2);
    if (fused_eval || fused_train) {
        // (y2 was written by k_conv12_fwd_split above)
        g2 = sample_plane_group_grid(batch, O2, split::kNP);
    } else if (conv_splitx_path(p, grid)) {
        const int XT = (O2 + 15) / 16;
        g2 = splitx::items(batch, O2, XT);
        static bool attr_sx = false;
        if (!attr_sx) {
            const hipError_t e = hipFuncSetAttribute((const void *)k_conv2_fwd_splitx, hipFuncAttributeMaxDynamicSharedMemorySize, splitx::kLdsBytes);
            if (e != hipSuccess) return (int)e;
            attr_sx = true;
        }
        hipLaunchKernelGGL(k_conv2_fwd_splitx, dim3(g2), dim3(split::kThreads), splitx::kLdsBytes, st, (const float *)y1, bn1, bn1 + kC, batch, O1, O2, XT,
                           (const uint4 *)w.w2split, p->b2, y2, training ? w.bn_part : nullptr);
    } else if (conv_split_path(p, grid)) {
        // (its weight images, and the data-gradient kernel's, were written by the conv1 kernel in passing)
        g2 = sample_plane_group_grid(batch, O2, split::kNP);
        static bool attr_split = false;
        if (!attr_split) {
            const hipError_t e = hipFuncSetAttribute((const void *)k_conv2_fwd_split, hipFuncAttributeMaxDynamicSharedMemorySize, split::kLdsBytes);
            if (e != hipSuccess) return (int)e;
            attr_split = true;
        }
        hipLaunchKernelGGL(k_conv2_fwd_split, dim3(g2), dim3(split::kThreads), split::kLdsBytes, st, (const float *)y1, bn1, bn1 + kC, batch, O1, O2,
                           (const uint4 *)w.w2split, p->b2, y2, training ? w.bn_part : nullptr);
    } else {
        hipLaunchKernelGGL(k_conv2_fwd<ActF32>, dim3(g2), dim3(kFwdThreads), 0, st, (const float *)y1, bn1, bn1 + kC, batch, O1, O2, w.w2img, p->b2, y2,
                       training ? w.bn_part : nullptr, pz2);
    }
    if ((err = gnbv_launch_status())) return err;
    if (training && dp) {
        // BatchNorm-2 over the global minibatch: this replica's (sum, sum of squares) -> sum over the replicas -> finalize
        hipLaunchKernelGGL(k_stats_reduce, dim3(1), dim3(1024), 0, st, w.bn_part, g2, p->sync_buf, 0.0, (const float *)nullptr, (const float *)nullptr,
                           0.0f, 0.0f, (float *)nullptr, (float *)nullptr, (int64_t *)nullptr, (const int *)nullptr, (float *)nullptr, (float *)nullptr,
                           (float *)nullptr, (float *)nullptr, (const float *)nullptr, (float *)nullptr);
        if ((err = gnbv_launch_status())) return err;
        if ((err = p->sync_sum(p->sync_ctx, 0, 2 * kC, stream))) return err;
        hipLaunchKernelGGL(k_bn_finalize_sums, dim3(1), dim3(64), 0, st, (const double *)p->sync_buf, (double)batch * P2 * p->world, p->bn2_w, p->bn2_b,
                           p->eps, p->momentum, p->bn2_rm, p->bn2_rv, p->bn2_nbt, skip_flag, bn2, bn2 + kC, bn2 + 2 * kC, bn2 + 3 * kC);
    } else if (training)
        hipLaunchKernelGGL(k_stats_reduce, dim3(1), dim3(1024), 0, st, w.bn_part, g2, (double *)nullptr, (double)batch * P2, p->bn2_w, p->bn2_b,
                           p->eps, p->momentum, p->bn2_rm, p->bn2_rv, p->bn2_nbt, skip_flag, bn2, bn2 + kC, bn2 + 2 * kC, bn2 + 3 * kC,
                           (const float *)nullptr, (float *)nullptr);
    else if (!(fused_eval && p->eval_prepared))
        hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(64), 0, st, (double)batch * P2, p->bn2_w, p->bn2_b, p->eps, p->momentum, p->bn2_rm,
                           p->bn2_rv, bn2, bn2 + kC, bn2 + 2 * kC, bn2 + 3 * kC);
    if ((err = gnbv_launch_status())) return err;
    // BN2 + ReLU -> flat features [B, 16*P2] (C-major, the reference's .reshape(num_env, -1))
    const int64_t total = (int64_t)batch * kC * P2;
    int grid_x = (int)((total + 255) / 256);
    grid_x = grid_x > 4096 ? 4096 : grid_x;
    if (features == nullptr) return gnbv_launch_status();  // (the consumer folds BN2 + ReLU into its operand load: gnbv_linear_forward_fold)
    hipLaunchKernelGGL(k_bn_relu_apply, dim3(grid_x), dim3(256), 0, st, y2, bn2, bn2 + kC, total, P2, features, p->range_flag);
    return gnbv_launch_status();
}

GNBV_API int gnbv_encoder_grid_backward(const float *obs_grid, const int64_t *rows, int64_t row_stride, int batch, int grid,
                                        const GnbvEncoderParams *p, const void *y1, const float *y2, const float *bn_state,
                                        const float *d_features, float *dy2_scratch, void *dz1_scratch,
                                        const GnbvEncoderGrads *g, void *workspace, size_t workspace_bytes, void *stream)
{
    GNBV_CHECK_ARG(p && (obs_grid || (p->grid_i8)) && y1 && y2 && bn_state && d_features && dy2_scratch && dz1_scratch && g && workspace);
    GNBV_CHECK_ARG(p->grid_i8 == nullptr || p->grid_i8_row_stride >= (int64_t)grid * grid * grid);
    GNBV_CHECK_ARG(batch > 0 && grid >= 7 && workspace_bytes >= gnbv_encoder_workspace_bytes(batch, grid));
    GNBV_CHECK_ARG(g->w1 && g->b1 && g->bn1_w && g->bn1_b && g->w2 && g->b2 && g->bn2_w && g->bn2_b);
    hipStream_t st = gnbv_stream(stream);
    const int O1 = out_size(grid), O2 = out_size(O1), P2 = O2 * O2 * O2;
    EncWs w = enc_carve(workspace, batch, grid);
    const float *bn1 = bn_state, *bn2 = bn_state + 4 * kC;
    int err;
    // conv2 data gradient fused with the conv1 weight gradient (dz1' never stored) when the grid exists as aligned
    // int8 rows; the input autocorrelation it needs runs on a second stream beside the kernels below
    const bool fused = fused_path(p, grid);  // (GENNBV_FUSED_BWD=0: A/B runs, bit-equality tests)
    GNBV_CHECK_ARG(p->autocorr == nullptr || (p->autocorr_row_stride >= kAcRow && p->autocorr_row_stride % 4 == 0 && ((uintptr_t)p->autocorr & 15) == 0));
    // the input autocorrelation: per-sample rows computed when the observation was produced (p->autocorr), or
    // the minibatch total computed here
    int *Rac = (int *)(w.red + 1024);  // [kAcRow]
    // (a training forward that computed BN1's statistics from the autocorrelation left the minibatch total in bn_state)
    const bool saved_total = fused && analytic_bn1(p, grid);
    if (fused && !saved_total && p->autocorr == nullptr && (err = launch_autocorr_total(p, rows, batch, grid, Rac, st))) return err;
    // ---- BN2 + ReLU backward ----
    hipLaunchKernelGGL(k_bn2_bwd_reduce, dim3(batch * kC), dim3(256), 0, st, d_features, y2, bn2, bn2 + kC, bn2 + 2 * kC,
                       bn2 + 3 * kC, P2, w.bn_part);
    if ((err = gnbv_launch_status())) return err;
    double *S2 = w.red + 128;
    unsigned *dy2_absmax = (unsigned *)(w.red + 2048);  // max |dy2| (bit patterns, 64 slots x 128 bytes): the gradient scale of the split kernels
    hipLaunchKernelGGL(k_bn2_bwd_finalize, dim3(1), dim3(256), 0, st, w.bn_part, batch, S2, dy2_absmax);
    if ((err = gnbv_launch_status())) return err;
    const bool dp = p->world > 1 && p->sync_sum != nullptr && p->sync_buf != nullptr;
    if (dp && !(fused && p->autocorr_global != nullptr && saved_total)) return (int)hipErrorInvalidValue;  // (see GnbvEncoderParams.world)
    const double *S2m = S2;  // the sums the elementwise BN2 backward uses: this replica's, or the global minibatch's
    if (dp) {
        if (hipMemcpyAsync(p->sync_buf + 2 * kC, S2, 2 * kC * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return (int)hipGetLastError();
        if ((err = p->sync_sum(p->sync_ctx, 2 * kC, 2 * kC, stream))) return err;
        S2m = p->sync_buf + 2 * kC;
    }
    hipLaunchKernelGGL(k_bn2_bwd_apply, dim3(batch * ((P2 + 63) / 64)), dim3(256), 0, st, d_features, y2, bn2, bn2 + kC, bn2 + 2 * kC, bn2 + 3 * kC, S2m,
                       (double)batch * P2 * (dp ? p->world : 1), P2, dy2_scratch, dy2_absmax);
    if ((err = gnbv_launch_status())) return err;
    // ---- conv2 weight gradient: on the side stream, beside the data gradient ----
    // (running this kernel on a second stream beside the data gradient was measured slower: both are bound by the CU's
    // vector-load path, profiles/r01_notes.md)
    hipStream_t sw = st;
    int nrows2 = batch * O2 * O2;
    // (the fp32 kernel: FOUR rows per wave where there are fewer rows than 512 workgroups x 4 waves x 4 -- at the reference's 20^3 one row per
    // wave made the launch its epilogue, 512 workgroup-level reductions and 14 MB of partial rows for 2 048 rows of four outputs: 128
    // workgroups -2.9 us per minibatch, 64: -1.4; profiles/r06_ab_train_g20_wgrad_blocks*.json)
    int wg_blocks = (nrows2 + 4 * kEncWaves - 1) / (4 * kEncWaves);
    wg_blocks = wg_blocks > 512 ? 512 : ((wg_blocks + 7) & ~7);
    const bool split_bwd = conv_split_path(p, grid);
    if (conv_splitx_path(p, grid)) {
        static bool attr_wx = false;
        if (!attr_wx) {
            hipError_t e = hipFuncSetAttribute((const void *)k_conv2_wgrad_splitx<true>, hipFuncAttributeMaxDynamicSharedMemorySize, splitx::kWgLdsBytes);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_conv2_wgrad_splitx<false>, hipFuncAttributeMaxDynamicSharedMemorySize, splitx::kWgLdsBytes);
            if (e != hipSuccess) return (int)e;
            attr_wx = true;
        }
        const int XT = (O2 + 15) / 16, nitems = splitx::items(batch, O2, XT);
        // one item per workgroup while the partial buffer holds a row per item (2 048 rows of E2: the whole of wg_part -- the conv1 partials
        // behind row 512 are written later on this stream, by the data gradient); else workgroups walk items (<= 512 rows)
        const bool wx_loop = nitems > 2048 || splitx_max_wg() < 512;
        wg_blocks = wx_loop ? min(nitems, splitx_max_wg()) : nitems;
        if (wx_loop)
            hipLaunchKernelGGL(k_conv2_wgrad_splitx<true>, dim3(wg_blocks), dim3(split::kThreads), splitx::kWgLdsBytes, sw, (const float *)y1, bn1, bn1 + kC,
                               dy2_scratch, (const unsigned *)dy2_absmax, batch, O1, O2, XT, nitems, w.wg_part);
        else
            hipLaunchKernelGGL(k_conv2_wgrad_splitx<false>, dim3(wg_blocks), dim3(split::kThreads), splitx::kWgLdsBytes, sw, (const float *)y1, bn1, bn1 + kC,
                               dy2_scratch, (const unsigned *)dy2_absmax, batch, O1, O2, XT, nitems, w.wg_part);
    } else if (split_bwd) {
        static bool attr_wg = false;
        if (!attr_wg) {
            const hipError_t e = hipFuncSetAttribute((const void *)k_conv2_wgrad_split, hipFuncAttributeMaxDynamicSharedMemorySize, split::kWgLdsBytes);
            if (e != hipSuccess) return (int)e;
            attr_wg = true;
        }
        wg_blocks = sample_plane_group_grid(batch, O2, split::kNP);
        if (wgrad_dma_path()) {
            // (round 6) the same contraction with y1 / dy2 streamed into LDS by LDS-DMA requests and converted in place: bit-identical
            static bool attr_wd = false;
            if (!attr_wd) {
                const hipError_t e = hipFuncSetAttribute((const void *)k_conv2_wgrad_split_dma, hipFuncAttributeMaxDynamicSharedMemorySize, wdma::kLdsBytes);
                if (e != hipSuccess) return (int)e;
                attr_wd = true;
            }
            const int nitems = wg_blocks;
            wg_blocks = min(nitems, conv_max_wg());  // (a workgroup walks items block, block + grid, ...: one workgroup per CU by default)
            hipLaunchKernelGGL(k_conv2_wgrad_split_dma, dim3(wg_blocks), dim3(split::kThreads), wdma::kLdsBytes, sw, (const float *)y1, bn1, bn1 + kC, dy2_scratch,
                               (const unsigned *)dy2_absmax, batch, O1, O2, nitems, w.wg_part);
        } else
        hipLaunchKernelGGL(k_conv2_wgrad_split, dim3(wg_blocks), dim3(split::kThreads), split::kWgLdsBytes, sw, (const float *)y1, bn1, bn1 + kC, dy2_scratch,
                           (const unsigned *)dy2_absmax, batch, O1, O2, w.wg_part);
    } else {
        hipLaunchKernelGGL(k_conv2_wgrad<ActF32>, dim3(wg_blocks), dim3(kEncThreads), 0, sw, (const float *)y1, bn1, bn1 + kC, dy2_scratch, batch, O1, O2,
                       w.wg_part);
    }
    if ((err = gnbv_launch_status())) return err;
    const int E2 = kTaps * 256 + kC;
    // the conv1 weight gradient (main stream) uses its own partial / slice regions of the workspace
    float *wg1_part = w.wg_part + (size_t)512 * E2;
    double *tmp1 = w.tmp + (size_t)32 * E2;
    {
        const int sl2 = reduce_stage1(w.wg_part, wg_blocks, E2, w.tmp, sw);  // <= 16 slices: tmp[0, 16 E2)
        if ((err = gnbv_launch_status())) return err;
        // (the weight images ride in the finish launch)
        hipLaunchKernelGGL(k_conv2_wgrad_finish, dim3((E2 + 255) / 256), dim3(256), 0, sw, (const double *)w.tmp, sl2, g->w2, g->b2,
                           p->w2, w.w2img);
        if ((err = gnbv_launch_status())) return err;
    }
    // ---- conv2 data gradient (+ ReLU1 mask, BN1 backward sums) ----
    int gd = sample_plane_group_grid(batch, (O1 + 1) / 2, kPlanesPerGroup);  // groups of plane PAIRS
    // (the data gradient keeps four plane pairs per workgroup at every size: with two -- 384 workgroups at 20^3 -- or one -- 640 -- it took
    // 20-21 us instead of 16: every workgroup starts by copying the 27 KiB weight image, and that prologue is what these launches are made of)
    // What does help is BALANCE at the same number of workgroups: the plane pairs of a sample in ceil(NA / 4) groups of equal size -- at
    // 20^3 (NA = 5) groups of 3 + 2 instead of 4 + 1: 15 super-tiles on a workgroup's 16 waves instead of 20, one round: -2.2 us per minibatch
    // (five in one group: +3.5; profiles/r06_ab_train_g20_dgrad_pz.json).  NA = 16, 32 (G = 64, 128) stay at four.
    const int nad = (O1 + 1) / 2, ngd = (nad + kPlanesPerGroup - 1) / kPlanesPerGroup;
    const int pzd = fused ? kPlanesPerGroup : (nad + ngd - 1) / ngd;
    if (!fused) gd = sample_plane_group_grid(batch, nad, pzd);
    if (fused) {
        if (conv_splitx_path(p, grid)) {
            static bool attr_dx = false;
            if (!attr_dx) {
                hipError_t e = hipFuncSetAttribute((const void *)k_conv2_dgrad_c1w_splitx<true>, hipFuncAttributeMaxDynamicSharedMemorySize, dsplit::kLdsBytes);
                if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_conv2_dgrad_c1w_splitx<false>, hipFuncAttributeMaxDynamicSharedMemorySize, dsplit::kLdsBytes);
                if (e != hipSuccess) return (int)e;
                attr_dx = true;
            }
            const int NA = (O1 + 1) / 2, XT = (NA + 15) / 16, nitems = dsplitx::items(batch, NA, XT);
            // (one item per workgroup while wg1_part -- 1 536 rows of E2 floats -- holds a row of kE1F floats per item; else T1 / S1 / S2 are kept
            // across a workgroup's items: <= 512 partial rows)
            const bool dx_loop = (size_t)nitems * kE1F > (size_t)1536 * (kTaps * 256 + kC) || splitx_max_wg() < 512;
            gd = dx_loop ? min(nitems, splitx_max_wg()) : nitems;
#define GNBV_DGX_ARGS dim3(gd), dim3(dsplit::kThreads), dsplit::kLdsBytes, st, dy2_scratch, (const uint4 *)(w.w2split + split::kW2ImgU4),                    \
                      (const float *)(w.w2split + split::kW2ImgU4 + dsplit::kImgSlotU4), (const unsigned *)dy2_absmax, (const float *)y1, bn1, bn1 + kC,        \
                      bn1 + 2 * kC, bn1 + 3 * kC, p->grid_i8, rows, p->grid_i8_row_stride, batch, grid, O1, O2, XT, nitems, wg1_part
            if (dx_loop)
                hipLaunchKernelGGL(k_conv2_dgrad_c1w_splitx<true>, GNBV_DGX_ARGS);
            else
                hipLaunchKernelGGL(k_conv2_dgrad_c1w_splitx<false>, GNBV_DGX_ARGS);
#undef GNBV_DGX_ARGS
        } else if (split_bwd) {
            static bool attr_dg = false;
            if (!attr_dg) {
                const hipError_t e = hipFuncSetAttribute((const void *)k_conv2_dgrad_c1w_split, hipFuncAttributeMaxDynamicSharedMemorySize, dsplit::kLdsBytes);
                if (e != hipSuccess) return (int)e;
                attr_dg = true;
            }
            hipLaunchKernelGGL(k_conv2_dgrad_c1w_split, dim3(gd), dim3(dsplit::kThreads), dsplit::kLdsBytes, st, dy2_scratch, (const uint4 *)(w.w2split + split::kW2ImgU4),
                               (const float *)(w.w2split + split::kW2ImgU4 + dsplit::kImgSlotU4),
                               (const unsigned *)dy2_absmax, (const float *)y1, bn1, bn1 + kC, bn1 + 2 * kC, bn1 + 3 * kC, p->grid_i8, rows, p->grid_i8_row_stride,
                               batch, grid, O1, O2, wg1_part);
        } else
            hipLaunchKernelGGL(k_conv2_dgrad_c1w, dim3(gd), dim3(kBigThreads), 0, st, dy2_scratch, (const float *)(w.w2img + kTaps * 256), (const float *)y1, bn1,
                               bn1 + kC, bn1 + 2 * kC, bn1 + 3 * kC, p->grid_i8, rows, p->grid_i8_row_stride, batch, grid, O1, O2, wg1_part);
        if ((err = gnbv_launch_status())) return err;
        const int slf = reduce_stage1(wg1_part, gd, kE1F, tmp1, st);
        if ((err = gnbv_launch_status())) return err;
        if (dp) {  // BatchNorm-1 backward means over the global minibatch
            hipLaunchKernelGGL(k_gather_bn1_sums, dim3(1), dim3(64), 0, st, (const double *)tmp1, slf, p->sync_buf + 4 * kC);
            if ((err = gnbv_launch_status())) return err;
            if ((err = p->sync_sum(p->sync_ctx, 4 * kC, 2 * kC, stream))) return err;
        }
        hipLaunchKernelGGL(k_c1w_fused_finish, dim3(1), dim3(1024), 0, st, (const double *)tmp1, slf,
                           saved_total ? (const int *)(bn_state + kBnStateFloats) : (p->autocorr ? (const int *)p->autocorr : (const int *)Rac),
                           p->autocorr_row_stride, rows, (!saved_total && p->autocorr) ? batch : 0,
                           p->w1, bn1, bn1 + 3 * kC, p->bn1_b, g->w1, g->b1, (const double *)S2, g->bn1_w,
                           g->bn1_b, g->bn2_w, g->bn2_b, dp ? (const double *)(p->sync_buf + 4 * kC) : (const double *)nullptr,
                           dp ? (const int *)p->autocorr_global : (const int *)nullptr);
        if ((err = gnbv_launch_status())) return err;
        return gnbv_launch_status();
    }
    {
        hipLaunchKernelGGL(k_conv2_dgrad<ActF32>, dim3(gd), dim3(kBigThreads), 0, st, dy2_scratch, (const float *)(w.w2img + kTaps * 256), (const float *)y1, bn1, bn1 + kC, bn1 + 2 * kC,
                       bn1 + 3 * kC, batch, O1, O2, (float *)dz1_scratch, w.bn_part, pzd);
    }
    if ((err = gnbv_launch_status())) return err;
    double *S1 = w.red + 192;
    hipLaunchKernelGGL(k_stats_reduce, dim3(1), dim3(1024), 0, st, w.bn_part, gd, S1, 0.0, (const float *)nullptr, (const float *)nullptr, 0.0f,
                       0.0f, (float *)nullptr, (float *)nullptr, (int64_t *)nullptr, (const int *)nullptr, (float *)nullptr, (float *)nullptr,
                       (float *)nullptr, (float *)nullptr, (const float *)nullptr, (float *)nullptr);
    if ((err = gnbv_launch_status())) return err;
    // ---- conv1 weight gradient (BN1 backward fused) ----
    int nrows1 = batch * O1 * O1;
    // (five rows per wave below the cap, for the same reason as the conv2 weight gradient above: 20^3, 2 048 -> 520 workgroups -6.7 us per
    // minibatch, 256: -4.2, 128: +4.3)
    int wg1_blocks = (nrows1 + 5 * kEncWaves - 1) / (5 * kEncWaves);
    wg1_blocks = wg1_blocks > 2048 ? 2048 : ((wg1_blocks + 7) & ~7);  // VGPR-light: 32 waves per CU hide the load latency
    const size_t c1w_lds = (size_t)kEncWaves * (2 * ((O1 + 1) / 2) * kC + 9 * grid) * sizeof(float);
    const int c1w_nr = (2 * ((O1 + 1) / 2) * kC + 255) / 256, c1w_ni = (9 * grid + 255) / 256;  // 16-byte requests per lane and row
    const bool c1w_staged = obs_grid != nullptr && (grid % 4 == 0) && (row_stride % 4 == 0) && (((uintptr_t)obs_grid & 15) == 0) && c1w_lds <= 64 * 1024 &&
                            c1w_nr <= 4 && c1w_ni <= 5;
    if (p->grid_i8 != nullptr && grid % 16 == 0 && 9 * grid <= 1024 && p->grid_i8_row_stride % 16 == 0 &&
               (((uintptr_t)p->grid_i8 & 15) == 0) && c1w_lds <= 64 * 1024 && c1w_nr <= 4) {
#define GNBV_C1W8(NR)                                                                                                             \
    hipLaunchKernelGGL((k_conv1_wgrad_lds<ActF32, NR, 1, int8_t>), dim3(wg1_blocks), dim3(kEncThreads), c1w_lds, st, p->grid_i8,  \
                       rows, p->grid_i8_row_stride, (const float *)dz1_scratch, (const float *)y1, bn1, bn1 + 2 * kC,            \
                       bn1 + 3 * kC, S1, (double)batch * O1 * O1 * O1, batch, grid, O1, wg1_part)
        if (c1w_nr <= 1) GNBV_C1W8(1);
        else if (c1w_nr <= 2) GNBV_C1W8(2);  // G = 64
        else GNBV_C1W8(4);
#undef GNBV_C1W8
    } else if (c1w_staged) {
#define GNBV_C1W(NR, NI)                                                                                                          \
    hipLaunchKernelGGL((k_conv1_wgrad_lds<ActF32, NR, NI, float>), dim3(wg1_blocks), dim3(kEncThreads), c1w_lds, st, obs_grid, rows,\
                       row_stride, (const float *)dz1_scratch, (const float *)y1, bn1, bn1 + 2 * kC, bn1 + 3 * kC, S1,            \
                       (double)batch * O1 * O1 * O1, batch, grid, O1, wg1_part)
        if (c1w_nr <= 1 && c1w_ni <= 1) GNBV_C1W(1, 1);
        else if (c1w_nr <= 1 && c1w_ni <= 2) GNBV_C1W(1, 2);
        else if (c1w_nr <= 2 && c1w_ni <= 3) GNBV_C1W(2, 3);  // G = 64
        else GNBV_C1W(4, 5);
#undef GNBV_C1W
    } else if (obs_grid == nullptr) {
        hipLaunchKernelGGL((k_conv1_wgrad<ActF32, int8_t>), dim3(wg1_blocks), dim3(kEncThreads), 0, st, p->grid_i8, rows, p->grid_i8_row_stride,
                           (const float *)dz1_scratch, (const float *)y1, bn1, bn1 + 2 * kC, bn1 + 3 * kC, S1, (double)batch * O1 * O1 * O1, batch,
                           grid, O1, wg1_part);
    } else {
        hipLaunchKernelGGL(k_conv1_wgrad<ActF32>, dim3(wg1_blocks), dim3(kEncThreads), 0, st, obs_grid, rows, row_stride, (const float *)dz1_scratch, (const float *)y1, bn1,
                       bn1 + 2 * kC, bn1 + 3 * kC, S1, (double)batch * O1 * O1 * O1, batch, grid, O1, wg1_part);
    }
    if ((err = gnbv_launch_status())) return err;
    const int E1 = 512 + kC;
    const int sl1 = reduce_stage1(wg1_part, wg1_blocks, E1, tmp1, st);
    if ((err = gnbv_launch_status())) return err;
    // (+ BN affine gradients: d beta = S[0], d gamma = S[1])
    hipLaunchKernelGGL(k_conv1_wgrad_finish, dim3((E1 + 255) / 256), dim3(256), 0, st, (const double *)tmp1, sl1, g->w1, g->b1,
                       (const double *)S1, (const double *)S2, g->bn1_w, g->bn1_b, g->bn2_w, g->bn2_b);
    if ((err = gnbv_launch_status())) return err;
    return gnbv_launch_status();
}

GNBV_API int gnbv_input_autocorr_row_ints(void) { return kAcRow; }

GNBV_API int gnbv_input_autocorr(const int8_t *grid_i8, int64_t grid_i8_row_stride, int n, int grid, int32_t *out, int64_t out_row_stride,
                                 void *stream)
{
    GNBV_CHECK_ARG(grid_i8 && out && n > 0 && grid >= 16 && grid % 16 == 0 && 3 * grid * grid <= 64 * 1024);
    GNBV_CHECK_ARG(grid_i8_row_stride >= (int64_t)grid * grid * grid && grid_i8_row_stride % 16 == 0 && (((uintptr_t)grid_i8 & 15) == 0));
    GNBV_CHECK_ARG(out_row_stride >= kAcRow);
    hipStream_t st = gnbv_stream(stream);
    if (hipMemset2DAsync(out, out_row_stride * sizeof(int), 0, kAcRow * sizeof(int), n, st) != hipSuccess) return (int)hipGetLastError();
    const int O1 = out_size(grid), P = autocorr_planes(grid);
    hipLaunchKernelGGL(k_input_autocorr, dim3(sample_plane_group_grid(n, O1, P)), dim3(kAcThreads), (size_t)(2 * P + 1) * grid * grid, st, grid_i8,
                       (const int64_t *)nullptr, grid_i8_row_stride, n, grid, O1, P, (int *)out, out_row_stride);
    return gnbv_launch_status();
}
