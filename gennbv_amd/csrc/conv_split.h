// The conv stack on the f16 matrix pipe with SPLIT operands (included by encoder.hip; G = 64-class shapes: ceil(O1/2) == 16,
// O2 <= 15).  Kernels, in file order:
//   k_conv2_fwd_split          conv2 forward (training and two-kernel inference)
//   k_conv2_wgrad_split        conv2 weight gradient (+ bias gradient)
//   k_conv2_dgrad_c1w_split    conv2 data gradient fused with the conv1 weight gradient and the BN1 backward sums
//   k_conv12_fwd_split<TRAIN>  conv1 + BN1 + ReLU + conv2 in one launch: inference (no layer-1 buffer at all) and the training
//                              forward with BN1's statistics known beforehand (y1 stored for the backward, never read back here)
//   k_conv1_fwd_split          conv1 forward on its own (training without the one-launch forward: y1 stored)
// plus the weight-image helpers the conv1 kernels run in passing.  Reference operators: gennbv/network/hybrid_encoder.py:38-45
// (`naive_encoder_grid`: Conv3d(1,16,3,2) - BN - ReLU - Conv3d(16,16,3,2) - BN - ReLU).
//
// Why: the fp32 kernels are bound by the CU's vector-load path (27 operand loads per 108 fp32 MFMAs, each re-fetching through the
// L1 what a neighbouring tap already fetched) and, when staged through LDS, by the fp32 matrix rate itself (108 x 32 cycles per
// 16 x 16 tile).  Here every operand is the exact sum of two f16 numbers,
//       x = hi + lo,   hi = f16(x),   lo = f16(x - hi)          (22 significand bits; |x - hi - lo| <= 2^-24 max(|x|, 2^-1) after
//                                                                 the power-of-two pre-scaling below)
// and a product a.b is accumulated in fp32 as  a_hi.b_hi + a_lo.b_hi + a_hi.b_lo  (the dropped a_lo.b_lo is 2^-22 |a.b|):
// three `v_mfma_f32_16x16x32_f16` (K = 32 = 2 taps x 16 channels, 16 cycles each) replace sixteen `v_mfma_f32_16x16x4_f32`
// (32 cycles each), a tile costs 42 x 16 instead of 108 x 32 matrix cycles, and the results stay within ~4 fp32 ulps of the
// fp32 kernels (tests: same tolerances as the fp32 path against the fp64 reference).  The input rows are read from global memory
// ONCE per chunk with contiguous 16-byte requests, BN + ReLU'd and split once, and kept in LDS; all 27 taps are LDS reads.
//
// Scaling (powers of two, exact): f16 has a narrow exponent range, so operands are pre-multiplied so that their `lo` parts stay
// out of the subnormal range for every value that matters and the accumulator is multiplied by the inverse at the end:
//   activations z1 = relu(bn1(y1))  x 2^8   (clamped to 65000 / 2^8 = 253.9 -- unreachable for normalised activations)
//   weights                        x 2^10  (|w| < 63.4)
//   gradients dy2                  x the power of two that puts max |dy2| (k_bn2_bwd_apply) into [2^13, 2^14)
//   layer-1 gradient g             x 2^-(10 + e), 2^e > max over (channel, parity class) of sum |W2|  (written with the images)
#pragma once
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

namespace split {
constexpr int kThreads = 1024, kWaves = 16;          // waves 0-7 compute (output plane = wave / 2, half of the k-steps each), waves 8-15 stage
constexpr int kConsWaves = 8, kProdThreads = 512;
constexpr int kNP = 4, kNPl = 2 * kNP + 1;           // output planes per workgroup / input planes they read
constexpr int kRing = 5;                             // input rows resident per plane: 3 being read + 2 being written
constexpr int kRowBytes = 2048;                      // one input row: [x parity 2][hi | lo][16 voxels][16 channels] f16
constexpr int kStageBytes = kNPl * kRing * kRowBytes;
constexpr int kPadBytes = 64;                        // a voxel: lane m = 15 (a padding output) of tap dx = 2 reads one voxel past its row
constexpr int kSlots = (2 * kNPl * 128 + kProdThreads - 1) / kProdThreads;  // 16-byte requests per staging thread and step (2 rows x 9 planes): 4.5 -> 5
constexpr int kRedBytes = 2 * kNP * 1024;            // k-half partial tiles, double-buffered by step parity
constexpr int kDyBytes = 2 * kNP * 1024;             // staged dy2 rows: [step parity][plane][hi | lo][16 voxels][16 channels] f16
constexpr int kGradBits = 14;                        // gradients are scaled so that their largest magnitude lies in [2^13, 2^14)
constexpr int kLdsBytes = kStageBytes + kPadBytes + kRedBytes;
constexpr int kWgLdsBytes = kStageBytes + kPadBytes + kDyBytes;
constexpr int kKSteps = 14, kKHalf = 7;                          // 27 taps (+ one zero tap) x 16 channels / 32
constexpr float kZScale = 256.0f, kWScale = 1024.0f, kZMax = 65000.0f;
constexpr int kW2ImgU4 = kKSteps * 2 * 64;           // uint4 per image: [k-step][hi | lo][lane]

__device__ __forceinline__ void split2(float x, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)x;
#ifdef SPLIT_HI_ONLY
    lo = (_Float16)0.0f;
#else
    lo = (_Float16)(x - (float)hi);
#endif
}
__device__ __forceinline__ f32x4 mfma_h(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// the two correction products (a_lo b_hi, a_hi b_lo) of a split product.  -DSPLIT_HI_ONLY (measurement build): dropped,
// with the `lo` planes' conversions and ring stores -- the arithmetic of a single-product f16 mode (11 significand bits) on this
// kernel structure, to price what such a mode could buy before building it.  profiles/r05_notes.md
#ifdef SPLIT_HI_ONLY
__device__ __forceinline__ f32x4 mfma_lo(h8, h8, f32x4 c) { return c; }
#else
__device__ __forceinline__ f32x4 mfma_lo(h8 a, h8 b, f32x4 c) { return mfma_h(a, b, c); }
#endif
}  // namespace split

// W2 [co][ci][27] -> the B-operand images of the split kernels, as two f16 planes scaled by 2^10:
//   fwd   image [k-step s][hi|lo][lane = 16 g + n][j] = W2[co = n][ci = 8 (g & 1) + j][tap = 2 s + (g >> 1)]      (tap 27: zero)
//   dgrad image: see prep_w2_dgrad_split_item
// Both are written by the conv1 forward kernel in passing (prep_w2_in_passing, encoder.hip), like the fp32 images.
__device__ __forceinline__ void prep_w2_split_item(int i, const float *__restrict__ W2, uint4 *__restrict__ img_fwd)
{
    const int s = i >> 6, lane = i & 63, n = lane & 15, g = lane >> 4, tap = 2 * s + (g >> 1), c0 = 8 * (g & 1);
    h8 fh, fl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float wf = tap < kTaps ? W2[((size_t)n * kC + c0 + j) * kTaps + tap] * split::kWScale : 0.0f;
        _Float16 hi, lo;
        split::split2(wf, hi, lo);
        fh[j] = hi;
        fl[j] = lo;
    }
    img_fwd[(s * 2 + 0) * 64 + lane] = *reinterpret_cast<uint4 *>(&fh);
    img_fwd[(s * 2 + 1) * 64 + lane] = *reinterpret_cast<uint4 *>(&fl);
}

// The staging role shared by the split kernels: one of 512 threads (8 waves) that move the two new input rows of an iteration
// (rows 2j+1, 2j+2 of the workgroup's <= 9 input planes, 36 KiB of fp32 y1) through registers into the LDS ring as f16 hi | lo
// planes of z1 = 2^8 relu(bn1(y1)).  Request k of a thread: region rs = 4 k + (wave >> 1) = 2 plane + (row & 1 ^ 1), the 16-byte
// piece `within` of that 2 KiB row.  Everything that selects a request is wave-uniform and the requests are UNCONDITIONAL
// (clamped into the sample): a branch around a load makes the compiler wait for every load at the join.
// NON-TEMPORAL requests for the streams of the minibatch that nothing re-reads soon (round 4): y1 (244 MB: written once by the training
// forward, read once by each backward kernel), and in csrc/ppo.hip the optimizer's moments and gradient.  Same-process A/B over
// three captures each (profiles/r04_notes.md): the y1 STORE as `nt` 581 -> 560 us per minibatch, the y1 LOADS -12.6, Adam's m / v / g
// -16.3 -- the streams no longer push what IS reused (fc_grid's weight and its gradient, 55 MB each, the parameters) out of L2 and
// the memory-side cache.  (The same flavour on k_grid_update_coded's stores made the voxel update 5 us SLOWER: measured, not assumed.)
__device__ __forceinline__ float4 ld4_nt(const float *p)
{
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint4 ldu4_nt(const int8_t *p)
{
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
// ---- the training forward's staging waves keep their own vector-memory books (round 5) -------------------------------------------
// Its y1 store must not be skipped for tiles the workgroup does not own (a branch around it makes the compiler's wait-count merge wait
// for the stores just issued), and as a branch-free store of all 64 lanes to a padding slot the dummies cost the store path as much
// as real ones -- 26 % of the launch's 1-KiB store instructions, with the texture path stalled on store data 43 % of the launch
// (profiles/r05_conv_pmc.txt).  So the store runs under an EXEC mask of ONE lane where the tile is not owned (the instruction still
// counts in vmcnt: the counts stay static), which only inline asm can do -- and once the stores are invisible to the compiler, so
// must be the requests of the same waves and the waits between them.
typedef unsigned u4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ldu4_nt_async(u4v_t &dst, const int8_t *p)
{
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_keep1(u4v_t &a)  // at most N younger vector-memory operations stay in flight; `a` is valid behind it
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}
// 16-byte non-temporal store of all lanes (mask_lo = mask_hi = ~0) or of lane 0 only (1, 0); both masks wave-uniform
__device__ __forceinline__ void st4_nt_masked(char *sbase, uint32_t voff, f32x4 v, uint32_t mask_lo, uint32_t mask_hi)
{
    uint32_t s0, s1;
    mask_lo = __builtin_amdgcn_readfirstlane(mask_lo), mask_hi = __builtin_amdgcn_readfirstlane(mask_hi);  // (scalar registers, whatever the caller's select became)
    asm volatile("s_mov_b32 %0, exec_lo\n\ts_mov_b32 %1, exec_hi\n\ts_and_b32 exec_lo, exec_lo, %5\n\ts_and_b32 exec_hi, exec_hi, %6\n\t"
                 "global_store_dwordx4 %2, %3, %4 nt\n\ts_mov_b32 exec_lo, %0\n\ts_mov_b32 exec_hi, %1"
                 : "=&s"(s0), "=&s"(s1)
                 : "v"(voff), "v"(v), "s"(sbase), "s"(mask_lo), "s"(mask_hi)
                 : "memory", "scc");
}

struct ZStager {
    const float *ybase;
    uint32_t rowC, planeC, st_lane;
    int npl, O1, half;  // half = staging wave >> 1
    bool pad_voxel;     // this thread's piece belongs to the padding voxel (x parity 1, slot 15) of its row
    float sc[4], sh[4];
    __device__ __forceinline__ void init(const float *y1, const float *scale1, const float *shift1, int b, int oz0, int npl_, int O1_, int ptid, int pw)
    {
        const int q = ptid & 3;
        const uint32_t within = ptid & 127;                                   // 16-byte piece inside a 2 KiB row
        st_lane = (within >> 6) * 1024 + ((within >> 2) & 15) * 32 + q * 8;   // its 8 hi bytes in the LDS row
        pad_voxel = (within >> 6) == 1 && ((within >> 2) & 15) == 15;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            sc[s] = scale1[4 * q + s] * split::kZScale;
            sh[s] = shift1[4 * q + s] * split::kZScale;
        }
        npl = npl_, O1 = O1_, half = pw >> 1;
        rowC = 2 * 16 * kC, planeC = rowC * O1;  // floats per input row (both parities, 16 voxel slots each) / plane
        ybase = y1 + ((size_t)b * O1 + 2 * oz0) * planeC + within * 4;
    }
    __device__ __forceinline__ void load(float4 (&regs)[split::kSlots], int j) const
    {
#pragma unroll
        for (int k = 0; k < split::kSlots; ++k) {
            const int rs = 4 * k + half, pi = min(rs >> 1, npl - 1), row = min(max(2 * j + 1 + (rs & 1), 0), O1 - 1);
            regs[k] = ld4_nt(ybase + (uint32_t)pi * planeC + (uint32_t)row * rowC);  // (y1 is streamed: read once per launch)
        }
    }
    // ZERO_PAD: the padding voxel is stored as 0 instead of whatever the (never written) y1 slot holds -- needed where an
    // operand built from it meets a zero of the other operand (0 x NaN), not where it only feeds a discarded output row
    template <bool ZERO_PAD>
    __device__ __forceinline__ void store(char *stage, const float4 (&regs)[split::kSlots], int j) const
    {
        using namespace split;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            const int rs = 4 * k + half, pi = rs >> 1, row = 2 * j + 1 + (rs & 1);
            if (rs >= 2 * kNPl) continue;            // (the half-empty last slot; wave-uniform, no request inside)
            const int slot = (row + kRing) % kRing;  // (rows outside the sample land in ring slots nobody reads any more)
            float z[4] = {__builtin_amdgcn_fmed3f(fmaf(sc[0], regs[k].x, sh[0]), 0.f, kZMax), __builtin_amdgcn_fmed3f(fmaf(sc[1], regs[k].y, sh[1]), 0.f, kZMax),
                          __builtin_amdgcn_fmed3f(fmaf(sc[2], regs[k].z, sh[2]), 0.f, kZMax), __builtin_amdgcn_fmed3f(fmaf(sc[3], regs[k].w, sh[3]), 0.f, kZMax)};
            h4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 a, c2;
                split2(ZERO_PAD && pad_voxel ? 0.0f : z[e], a, c2);
                hi[e] = a;
                lo[e] = c2;
            }
            char *dst = stage + (pi * kRing + slot) * kRowBytes + st_lane;
            *reinterpret_cast<h4 *>(dst) = hi;
#ifndef SPLIT_HI_ONLY
            *reinterpret_cast<h4 *>(dst + 512) = lo;
#endif
        }
    }
};
// one barrier per step.  (sched_barrier: the scheduler must not hoist the NEXT step's register-only transform, and with it the
// wait for its requests, above this step's barrier; lgkmcnt only: the prefetched requests stay in flight)
__device__ __forceinline__ void split_step_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------
// conv2 forward.  Workgroup = 16 waves = (sample, 4 output planes), one per CU (100 KiB of LDS), walking the 15 output rows.
//   waves 8-15 (staging): per step the two NEW input rows of the 9 input planes (36 KiB of y1) are requested two steps ahead
//       (2 x 5 x 16 bytes per thread in registers), then BN1 + ReLU'd, scaled, split and stored as f16 hi | lo planes in the
//       MFMA A-operand layout [voxel][16 channels] into a ring of 5 rows per plane -- every y1 byte is read from global
//       memory once per workgroup (9 planes for 8: 1.13 x the unique bytes) with contiguous 16-byte requests;
//   waves 0-7 (compute): a 16-voxel x 16-channel tile per step and wave PAIR; each wave of the pair runs 7 of the 14 k-steps
//       (2 ds_read_b128 + 3 MFMA each) with its half of the weights (2 x 7 fragments) in registers; the odd wave hands its partial
//       tile to the even one through LDS, which adds it one step later (no extra barrier) and writes y2 + the BN2 partial sums.
// Four waves per SIMD: a lone wave issues a dependent instruction only every ~8 cycles, which bounded both roles at 1 / SIMD.
// Step t: staging stores rows 2t+1, 2t+2 while compute reads rows 2t-2 .. 2t (five consecutive rows: distinct ring slots);
// one barrier per step.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(split::kThreads) void k_conv2_fwd_split(
    const float *__restrict__ y1, const float *__restrict__ scale1, const float *__restrict__ shift1, int B, int O1, int O2,
    const uint4 *__restrict__ w2img, const float *__restrict__ b2, float *__restrict__ y2, float *__restrict__ partials)
{
    using namespace split;
    extern __shared__ __attribute__((aligned(16))) char split_lds[];
    char *stage = split_lds;
    float *red = reinterpret_cast<float *>(split_lds + split::kStageBytes + kPadBytes);
    int b, oz0, oz1;
    const bool live = sample_plane_group(B, O2, kNP, b, oz0, oz1);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    if (!live) { write_partials(partials, kWaves, wv, 0.f, 0.f); return; }
    const int np = oz1 - oz0, npl = 2 * np + 1;
    float s_sum = 0.0f, s_sq = 0.0f;
    const int nsteps = (O2 + 2) & ~1;  // O2 compute steps + the deferred epilogue of the last row, rounded up to even
    auto step_barrier = [&]() { split_step_barrier(); };
    if (wv >= kConsWaves) {
        // ---- staging waves ----
        ZStager zs;
        zs.init(y1, scale1, shift1, b, oz0, npl, O1, tid - kConsWaves * kWave, wv - kConsWaves);
        auto load_iter = [&](float4 (&regs)[kSlots], int j) { zs.load(regs, j); };
        auto store_iter = [&](const float4 (&regs)[kSlots], int j) { zs.store<false>(stage, regs, j); };
        float4 ra[kSlots], rb[kSlots];
        load_iter(ra, -1);
        load_iter(rb, 0);
        store_iter(ra, -1);
        load_iter(ra, 1);
        store_iter(rb, 0);
        load_iter(rb, 2);
        step_barrier();
        // steps t = 1 .. nsteps.  Set ra holds odd iterations, rb even ones.  The body is BRANCH-FREE around the requests on
        // purpose: with a conditional load anywhere in the loop the compiler's wait-count merge assumes the worst path and waits
        // for the just-issued requests (depth 1 instead of 2).  Stores of the steps past the last real iteration land in ring
        // slots nobody reads any more, their requests are clamped duplicates.
        for (int t = 1; t <= nsteps; t += 2) {
            store_iter(ra, t);
            load_iter(ra, t + 2);
                step_barrier();
            store_iter(rb, t + 1);
            load_iter(rb, t + 3);
                step_barrier();
        }
    } else {
        // ---- compute waves ----
        const int m = lane & 15, g = lane >> 4;
        const int pl = wv >> 1, kh = wv & 1;
        h8 wh[kKHalf], wl[kKHalf];
#pragma unroll
        for (int s = 0; s < kKHalf; ++s) {
            const uint4 uh = w2img[((kh * kKHalf + s) * 2 + 0) * 64 + lane], ul = w2img[((kh * kKHalf + s) * 2 + 1) * 64 + lane];
            wh[s] = *reinterpret_cast<const h8 *>(&uh);
            wl[s] = *reinterpret_cast<const h8 *>(&ul);
        }
        const float bias = b2[m];
        const int P2 = O2 * O2 * O2;
        const uint32_t a_lane = (uint32_t)(2 * pl * kRing * kRowBytes + m * 32 + (g & 1) * 16);
        const bool second = (g >> 1) != 0;  // lanes 32-63 feed the second tap of a k-step
        f32x4 prev = {0.f, 0.f, 0.f, 0.f};   // the even wave's own half of the previous step's tile
        float *const out_base = y2 + ((size_t)b * kC + m) * P2 + (size_t)(oz0 + pl) * O2 * O2;
        step_barrier();
        for (int t = 1; t <= nsteps; ++t) {
            const int oy = t - 1;
            // the previous row's tile: own half + the odd wave's half (written before the last barrier)
            if (kh == 0 && pl < np && oy >= 1 && oy - 1 < O2) {
                const f32x4 other = *reinterpret_cast<const f32x4 *>(red + (((t - 1) & 1) * kNP + pl) * 256 + lane * 4);
                const f32x4 acc = (prev + other) * (1.0f / (kZScale * kWScale));
                float *out = out_base + (size_t)(oy - 1) * O2;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int oxi = 4 * g + rr;
                    if (oxi < O2) {
                        const float y = acc[rr] + bias;
                        out[oxi] = y;
                        s_sum += y;
                        s_sq += y * y;
                    }
                }
            }
            if (pl < np && oy < O2) {
                uint32_t rowoff[3];  // ring slots of input rows 2 oy + dy
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) rowoff[dy] = (uint32_t)(((2 * oy + dy) % kRing) * kRowBytes);
                auto tap_off = [&](int tp) {
                    const int dz = tp / 9, dy = (tp / 3) % 3, dx = tp % 3;
                    return (uint32_t)(dz * kRing * kRowBytes + (dx == 1 ? 1024 : 0) + (dx == 2 ? 32 : 0)) + rowoff[dy];
                };
                // k-step s of this wave's half: taps 2 (7 kh + s), + 1  (tap 27 has zero weights: any readable address);
                // kh is wave-uniform, so both candidates are scalar values
                auto a_off = [&](int s) {
                    const uint32_t ta = kh ? tap_off(2 * (kKHalf + s)) : tap_off(2 * s);
                    const uint32_t tb = kh ? tap_off(min(2 * (kKHalf + s) + 1, kTaps - 1)) : tap_off(2 * s + 1);
                    return a_lane + (second ? tb : ta);
                };
                f32x4 acc_hh = {0.f, 0.f, 0.f, 0.f}, acc_lh = acc_hh, acc_hl = acc_hh;
                constexpr int kAhead = 2;  // operands of k-step s + 2 are requested before the MFMAs of k-step s issue
                h8 ah[kAhead + 1], al[kAhead + 1];
#pragma unroll
                for (int s = 0; s < kAhead; ++s) {
                    ah[s] = *reinterpret_cast<const h8 *>(stage + a_off(s));
                    al[s] = *reinterpret_cast<const h8 *>(stage + a_off(s) + 512);
                }
#pragma unroll
                for (int s = 0; s < kKHalf; ++s) {
                    if (s + kAhead < kKHalf) {
                        ah[(s + kAhead) % (kAhead + 1)] = *reinterpret_cast<const h8 *>(stage + a_off(s + kAhead));
                        al[(s + kAhead) % (kAhead + 1)] = *reinterpret_cast<const h8 *>(stage + a_off(s + kAhead) + 512);
                    }
                    acc_hh = mfma_h(ah[s % (kAhead + 1)], wh[s], acc_hh);
                    acc_lh = mfma_lo(al[s % (kAhead + 1)], wh[s], acc_lh);
                    acc_hl = mfma_lo(ah[s % (kAhead + 1)], wl[s], acc_hl);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const f32x4 part = acc_hh + (acc_lh + acc_hl);
                if (kh)
                    *reinterpret_cast<f32x4 *>(red + ((t & 1) * kNP + pl) * 256 + lane * 4) = part;
                else
                    prev = part;
            }
                step_barrier();
        }
    }
    write_partials(partials, kWaves, wv, s_sum, s_sq);
}


// ---------------------------------------------------------------------------
// conv2 weight gradient: dW2[tap][ci][co] = sum over output voxels of z1[input voxel(tap)][ci] * dy2[voxel][co]  (+ the bias
// gradient, sum of dy2).  Same workgroup, ring and staging waves as the forward; the contraction now runs over VOXELS, and both
// operands sit in LDS voxel-major ([voxel][16 channels]), so both reach the MFMA through the transposing LDS read
// `ds_read_b64_tr_b16`: with lane l reading the 8 bytes at l * 8 of a 512-byte [16 voxels][16 channels] block, lane (n = l & 15,
// g = l >> 4) receives channel n of voxels 4g .. 4g+3 -- four consecutive k of column n.  A k-block of 32 = two such blocks (the
// rows of two output planes), i.e. two reads per operand half.
//   MFMA: i = ci (A = z1 transposed), j = co (B = dy2), k = 32 output voxels; D -> partial[tap][ci][co] as the fp32 kernels.
//   waves 0-7: every tap belongs to ONE wave (tap = wave + 8 i: four taps for waves 0-2, three for the rest; wave 7 also sums dy2
//       for the bias gradient with a ones operand), so the accumulators live in registers for the whole kernel and there is no
//       cross-wave reduction: 2 k-blocks x (4 reads + taps x (4 reads + 3 MFMA)) per step;
//   waves 8-15: stage z1 as in the forward (the padding voxel zeroed: it meets dy2's zero padding) and the step's dy2 rows
//       (4 planes x 15 voxels x 16 channels, scaled by 2^s and split; voxel 15 and missing planes are zeros).
// dy2 is a gradient: its magnitude is unknown at compile time and f16's exponent range is narrow, so k_bn2_bwd_apply leaves
// max |dy2| (atomicMax on the bit pattern) and the kernels scale by the power of two that puts it in [2^13, 2^14).
// ---------------------------------------------------------------------------
typedef short s4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h8 tr_pair(const char *p0, const char *p1)
{
    typedef __attribute__((address_space(3))) s4v *lds_s4;
    const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)p0), b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)p1);
    typedef short s8v __attribute__((ext_vector_type(8)));
    const s8v r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return *reinterpret_cast<const h8 *>(&r);
}
__device__ __forceinline__ float grad_scale(const unsigned *absmax)
{
    unsigned bits = absmax[(threadIdx.x & 63) * 32];  // 64 slots (k_bn2_bwd_apply); every lane ends with the maximum
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, d, 64));
    const float amax = __uint_as_float(bits);
    int e = 0;
    (void)frexpf(amax, &e);  // amax = f 2^e, f in [0.5, 1)
    const float sc = amax > 0.0f && amax < 3.0e38f ? ldexpf(1.0f, split::kGradBits - e) : 1.0f;
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sc)));  // (the same in every lane: keep it in a scalar register)
}

// 1 / p for p an exact power of two (the scales above): an exponent flip instead of a division, whose scale / fixup temporaries the
// compiler kept alive across the whole kernel (and spilled)
__device__ __forceinline__ float inv_pow2(float p) { return __int_as_float(0x7f000000 - __float_as_int(p)); }

__device__ __forceinline__ void conv2_wgrad_split_body(
    const float *__restrict__ y1, const float *__restrict__ scale1, const float *__restrict__ shift1, const float *__restrict__ dy2 /*[B,O2^3,16]*/,
    const unsigned *__restrict__ absmax, int B, int O1, int O2, float *__restrict__ partial /*[grid][27 * 256 + 16]*/, const int vblock)
{
    using namespace split;
    extern __shared__ __attribute__((aligned(16))) char split_lds[];
    char *stage = split_lds, *dyst = split_lds + split::kStageBytes + kPadBytes;
    int b, oz0, oz1;
    const bool live = sample_plane_group(B, O2, kNP, b, oz0, oz1, vblock);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    constexpr int E2 = kTaps * 256 + kC;
    float *out = partial + (size_t)vblock * E2;
    if (!live) {
        for (int i = tid; i < E2; i += kThreads) out[i] = 0.0f;
        return;
    }
    // (stale LDS may hold NaN patterns; the ring is read one voxel past a row and, in the first steps, next to slots not yet written)
    for (int i = tid; i < kWgLdsBytes / 16; i += kThreads) reinterpret_cast<uint4 *>(split_lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const int np = oz1 - oz0, npl = 2 * np + 1, P2 = O2 * O2 * O2;
    const int nsteps = (O2 + 1) & ~1;
    const float gs = grad_scale(absmax);
    if (wv >= kConsWaves) {
        // ---- staging waves ----
        const int ptid = tid - kConsWaves * kWave, pw = wv - kConsWaves;
        ZStager zs;
        zs.init(y1, scale1, shift1, b, oz0, npl, O1, ptid, pw);
        // dy2: thread -> (plane, voxel, channel quad) of the step's rows; the upper four waves request a duplicate they never store
        const int dpl = (ptid >> 6) & 3, dpiece = ptid & 63, dvox = dpiece >> 2;
        const bool dvalid = dpl < np && dvox < O2;
        const float *dsrc = dy2 + ((size_t)b * P2 + (size_t)(oz0 + min(dpl, np - 1)) * O2 * O2 + min(dvox, O2 - 1)) * kC + 4 * (dpiece & 3);
        auto load_iter = [&](float4 (&regs)[kSlots], float4 &d, int j) {
            zs.load(regs, j);
            d = *reinterpret_cast<const float4 *>(dsrc + (size_t)min(max(j, 0), O2 - 1) * O2 * kC);
        };
        auto store_iter = [&](const float4 (&regs)[kSlots], const float4 &d, int j) {
            zs.store<true>(stage, regs, j);
            if (pw < 4) {
                const float v[4] = {d.x, d.y, d.z, d.w};
                h4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 a, c2;
                    split2(dvalid ? v[e] * gs : 0.0f, a, c2);
                    hi[e] = a;
                    lo[e] = c2;
                }
                char *dst = dyst + ((j & 1) * kNP + dpl) * 1024 + dpiece * 8;
                *reinterpret_cast<h4 *>(dst) = hi;
#ifndef SPLIT_HI_ONLY
                *reinterpret_cast<h4 *>(dst + 512) = lo;
#endif
            }
        };
        float4 ra[kSlots], rb[kSlots], da, db;
        load_iter(ra, da, -1);
        load_iter(rb, db, 0);
        store_iter(ra, da, -1);
        load_iter(ra, da, 1);
        store_iter(rb, db, 0);
        load_iter(rb, db, 2);
        split_step_barrier();
        for (int t = 1; t <= nsteps; t += 2) {  // (branch-free around the requests: see the forward kernel)
            store_iter(ra, da, t);
            load_iter(ra, da, t + 2);
            split_step_barrier();
            store_iter(rb, db, t + 1);
            load_iter(rb, db, t + 3);
            split_step_barrier();
        }
    } else {
        // ---- compute waves ----
        const int n = lane & 15, g = lane >> 4, cw = wv;
        const int ntaps = cw < 3 ? 4 : 3;
        uint32_t tapbase[4];
        int tapdy[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tp = min(cw + 8 * i, kTaps - 1), dz = tp / 9, dy = (tp / 3) % 3, dx = tp % 3;
            tapbase[i] = (uint32_t)(dz * kRing * kRowBytes + (dx == 1 ? 1024 : 0) + (dx == 2 ? 32 : 0)) + lane * 8;
            tapdy[i] = dy;
        }
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        h8 ones;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (_Float16)1.0f;
        split_step_barrier();
        for (int t = 1; t <= nsteps; ++t) {
            const int oy = t - 1;
            if (oy < O2) {
                uint32_t rowoff[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) rowoff[dy] = (uint32_t)(((2 * oy + dy) % kRing) * kRowBytes);
                const char *dybuf = dyst + (oy & 1) * kNP * 1024 + lane * 8;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (2 * kb < np) {  // (np = 3: the second block's missing plane is staged as zeros)
                        const h8 bh = tr_pair(dybuf + (2 * kb) * 1024, dybuf + (2 * kb + 1) * 1024);
                        const h8 bl = tr_pair(dybuf + (2 * kb) * 1024 + 512, dybuf + (2 * kb + 1) * 1024 + 512);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (i < ntaps) {
                                const uint32_t off = tapbase[i] + (tapdy[i] == 0 ? rowoff[0] : tapdy[i] == 1 ? rowoff[1] : rowoff[2]);
                                const char *a0 = stage + (2 * (2 * kb)) * kRing * kRowBytes + off, *a1 = a0 + 2 * kRing * kRowBytes;
                                const h8 ah = tr_pair(a0, a1), al = tr_pair(a0 + 512, a1 + 512);
                                acc[i] = mfma_h(ah, bh, acc[i]);
                                acc[i] = mfma_lo(al, bh, acc[i]);
                                acc[i] = mfma_lo(ah, bl, acc[i]);
                            }
                        }
                        if (cw == 7) {
                            acc[3] = mfma_h(ones, bh, acc[3]);
                            acc[3] = mfma_lo(ones, bl, acc[3]);
                        }
                    }
                }
            }
            split_step_barrier();
        }
        const float unscale = (1.0f / kZScale) * inv_pow2(gs);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < ntaps) {
                const int tp = cw + 8 * i;
#pragma unroll
                for (int r = 0; r < 4; ++r) out[tp * 256 + (4 * g + r) * kC + n] = acc[i][r] * unscale;
            }
        }
        if (cw == 7 && g == 0) out[kTaps * 256 + n] = acc[3][0] * inv_pow2(gs);
    }
}

__global__ __launch_bounds__(split::kThreads) void k_conv2_wgrad_split(
    const float *__restrict__ y1, const float *__restrict__ scale1, const float *__restrict__ shift1, const float *__restrict__ dy2 /*[B,O2^3,16]*/,
    const unsigned *__restrict__ absmax, int B, int O1, int O2, float *__restrict__ partial /*[grid][27 * 256 + 16]*/)
{
    conv2_wgrad_split_body(y1, scale1, shift1, dy2, absmax, B, O1, O2, partial, (int)blockIdx.x);
}


// ---------------------------------------------------------------------------
// conv2 weight gradient, LDS-DMA transport (round 6).  Same workgroup, same compute waves, same arithmetic in the same
// order as k_conv2_wgrad_split (the results are bit-identical), but the staging waves no longer pull y1 / dy2 through their registers:
//   * a staging wave ISSUES `global_load_lds_dwordx4 ... nt` for its chunks of the NEXT iteration -- 1 KiB per wave instruction, 64 lanes
//     x 16 bytes, straight from L2 / HBM into the ring slot the data will live in (raw fp32: a half row [16 voxels][16 channels] of y1
//     is 1 KiB, exactly the size of its f16 hi | lo image);
//   * one step later the SAME wave converts the chunk IN PLACE: one ds_read_b128 per lane (4 channels of a voxel), BN1 + ReLU + scale +
//     split as before, two ds_write_b64 (hi at voxel * 32 + quad * 8, lo 512 bytes further) into the same 1 KiB.  A chunk belongs to one
//     wave from request to converted image, so the only ordering needed is the wave's own `s_waitcnt vmcnt(N)` (all lanes' reads of a
//     ds_read instruction are served before the wave's next LDS instruction) -- no barrier between landing and conversion.
// The ring has SEVEN rows per plane instead of five: rows 2t-2 .. 2t being read by the compute waves, 2t+1, 2t+2 being converted,
// 2t+3, 2t+4 landing (126 KiB + three dy2 buffers of 4 KiB = 138 KiB of the CU's 160).  40 chunks per step (36 y1 half rows + 4 dy2 rows)
// = 5 per staging wave, all issued unconditionally (clamped source addresses, every chunk into its own slot), so the wait counts are
// literals.  What the staging waves lose: 40 VGPRs of request registers, the global_load -> VGPR return path and its `s_waitcnt` on
// register data.  What they gain: one ds_read_b128 per chunk (the LDS array was 18 % busy in the register version).
// ---------------------------------------------------------------------------
namespace wdma {
constexpr int kRing = 7;
constexpr int kStageBytes = split::kNPl * kRing * split::kRowBytes;  // 129 024
constexpr int kDyBufs = 3;                                            // dy2 rows: being read / being converted / landing
constexpr int kDyBytes = kDyBufs * split::kNP * 1024;
constexpr int kLdsBytes = kStageBytes + split::kPadBytes + kDyBytes;  // 141 376
constexpr int kChunks = 5;                                            // per staging wave and step
static_assert(4 * split::kNPl + split::kNP == kChunks * (split::kWaves - split::kConsWaves), "40 chunks over 8 staging waves");
}  // namespace wdma

// 16 bytes per lane from global memory straight into LDS at (wave-uniform byte address lds_dst) + lane * 16; non-temporal (y1 / dy2 are
// streamed).  M0 carries the LDS base; it is compiler-reserved, so it is saved and restored inside the statement.  Counts in vmcnt.
__device__ __forceinline__ void glds16_nt(const void *gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_keep()  // at most N younger vector-memory operations (LDS-DMA requests) stay in flight
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}

__global__ __launch_bounds__(split::kThreads) void k_conv2_wgrad_split_dma(
    const float *__restrict__ y1, const float *__restrict__ scale1, const float *__restrict__ shift1, const float *__restrict__ dy2 /*[B,O2^3,16]*/,
    const unsigned *__restrict__ absmax, int B, int O1, int O2, int nitems, float *__restrict__ partial /*[grid][27 * 256 + 16]*/)
{
    // A workgroup walks the items (sample, plane group) blockIdx.x, + gridDim.x, ... and keeps its accumulators across them (round 6): with
    // 256 workgroups for the 512 items of a 128-sample minibatch every CU starts ONE workgroup instead of two in a row -- the second one's
    // launch, scale loads and first round trip were dead time on the CU (the x-tiled kernels showed it: 90.7 -> 80.6 us from 512 to 256
    // workgroups, profiles/r06_splitx_at_g64_trace.txt).  With gridDim.x >= nitems it is the one-item kernel, bit-identical to
    // k_conv2_wgrad_split; with fewer workgroups a partial row sums two items in fp32 before the fp64 reduction.
    using namespace split;
    extern __shared__ __attribute__((aligned(16))) char split_lds[];
    char *stage = split_lds, *dyst = split_lds + wdma::kStageBytes + kPadBytes;
    constexpr int R = wdma::kRing;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    constexpr int E2 = kTaps * 256 + kC;
    float *out = partial + (size_t)blockIdx.x * E2;
    // Only the padding voxel behind the ring is cleared: every ring slot and every dy2 buffer is written in full (1 KiB chunks, all nine
    // planes, clamped sources) before its first read, so nothing stale can reach an operand -- and the first requests can go out at once,
    // before anything else of the workgroup's start-up (no clear of 138 KiB + barrier in front of them, no round trip for the scales).
    if (tid < kPadBytes / 16) reinterpret_cast<uint4 *>(split_lds + wdma::kStageBytes)[tid] = make_uint4(0, 0, 0, 0);
    const int P2 = O2 * O2 * O2;
    const int nsteps = (O2 + 1) & ~1;
    if (wv >= kConsWaves) {
        // ---- staging waves: chunk k of wave pw is half row h = pw + 8 k of the iteration (plane h >> 2, row parity (h >> 1) & 1, x parity
        // h & 1) for h < 36; the fifth chunk of waves 4-7 is the dy2 row of plane pw - 4 ----
        const int pw = wv - kConsWaves, q = lane & 3, vx = lane >> 2;
        const bool dy_wave = pw >= 4;
        const uint32_t rowC = 2 * 16 * kC, planeC = rowC * (uint32_t)O1;
        const int dpl = pw - 4;
        const uint32_t stage_a = lds_addr(stage), dy_a = lds_addr(dyst);
        float gs = 1.0f, sc[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        bool first = true;
        for (int item = (int)blockIdx.x; item < nitems; item += (int)gridDim.x) {
            int b, oz0, oz1;
            if (!sample_plane_group(B, O2, kNP, b, oz0, oz1, item)) continue;  // (workgroup-uniform: both roles skip it)
            const int np = oz1 - oz0, npl = 2 * np + 1;
            const float *ybase = y1 + ((size_t)b * O1 + 2 * oz0) * planeC + lane * 4;
            const bool dvalid = dy_wave && dpl < np && vx < O2;
            const float *dsrc = dy2 + ((size_t)b * P2 + (size_t)(oz0 + min(max(dpl, 0), np - 1)) * O2 * O2 + min(vx, O2 - 1)) * kC + 4 * q;
            auto issue = [&](int j) {
#ifdef WDMA_ABL_NODMA  // (measurement build: no requests -- conversion + contraction alone)
                return;
#endif
#pragma unroll
                for (int k = 0; k < wdma::kChunks; ++k) {
                    if (k == wdma::kChunks - 1 && dy_wave) {
                        glds16_nt(dsrc + (size_t)min(max(j, 0), O2 - 1) * O2 * kC, __builtin_amdgcn_readfirstlane(dy_a + (uint32_t)((((j + 3) % wdma::kDyBufs) * kNP + dpl) * 1024)));
                    } else {
                        const int h = pw + 8 * k, pi = h >> 2, row = 2 * j + 1 + ((h >> 1) & 1), slot = (row + 2 * R) % R;
                        const float *src = ybase + (uint32_t)min(pi, npl - 1) * planeC + (uint32_t)min(max(row, 0), O1 - 1) * rowC + (h & 1) * (16 * kC);
                        glds16_nt(src, __builtin_amdgcn_readfirstlane(stage_a + (uint32_t)((pi * R + slot) * kRowBytes + (h & 1) * 1024)));
                    }
                }
            };
            auto convert = [&](int j) {
#ifdef WDMA_ABL_NOCONV  // (measurement build: the chunks stay raw -- transport + contraction alone, results meaningless)
                return;
#endif
                char *cp[wdma::kChunks];
                float4 raw[wdma::kChunks];
#pragma unroll
                for (int k = 0; k < wdma::kChunks; ++k) {
                    if (k == wdma::kChunks - 1 && dy_wave) {
                        cp[k] = dyst + (((j + 3) % wdma::kDyBufs) * kNP + dpl) * 1024;
                    } else {
                        const int h = pw + 8 * k, pi = h >> 2, row = 2 * j + 1 + ((h >> 1) & 1), slot = (row + 2 * R) % R;
                        cp[k] = stage + (pi * R + slot) * kRowBytes + (h & 1) * 1024;
                    }
                    raw[k] = *reinterpret_cast<const float4 *>(cp[k] + lane * 16);
                }
#pragma unroll
                for (int k = 0; k < wdma::kChunks; ++k) {
                    float z[4];
                    const float v[4] = {raw[k].x, raw[k].y, raw[k].z, raw[k].w};
                    if (k == wdma::kChunks - 1 && dy_wave) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) z[e] = dvalid ? v[e] * gs : 0.0f;
                    } else {
                        const bool pad_voxel = ((pw + 8 * k) & 1) == 1 && vx == 15;  // (x parity 1, slot 15: meets dy2's zero padding)
#pragma unroll
                        for (int e = 0; e < 4; ++e) z[e] = pad_voxel ? 0.0f : __builtin_amdgcn_fmed3f(fmaf(sc[e], v[e], sh[e]), 0.f, kZMax);
                    }
                    h4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 a, c2;
                        split2(z[e], a, c2);
                        hi[e] = a;
                        lo[e] = c2;
                    }
                    char *dst = cp[k] + vx * 32 + q * 8;
                    *reinterpret_cast<h4 *>(dst) = hi;
#ifndef SPLIT_HI_ONLY
                    *reinterpret_cast<h4 *>(dst + 512) = lo;
#endif
                }
            };
            issue(-1);
            issue(0);
            if (first) {
                // the scales, requested BEHIND the first ten chunks: the compiler's wait for them (it does not count the requests above) also
                // retires those -- which the first conversion needs anyway
                gs = grad_scale(absmax);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    sc[s] = scale1[4 * q + s] * kZScale;
                    sh[s] = shift1[4 * q + s] * kZScale;
                }
                first = false;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(sc[s]), "+v"(sh[s]));  // (computed HERE: a use sunk behind later requests would make the compiler wait for them)
            wait_vm_keep<0>();
            convert(-1);
            issue(1);
            convert(0);  // (iteration 0 landed with iteration -1: the wait above was for everything)
            split_step_barrier();
            for (int t = 1; t <= nsteps; ++t) {
                issue(t + 1);  // rows 2t+3, 2t+4: their slots held rows 2t-4, 2t-3, last read before the previous barrier
                wait_vm_keep<wdma::kChunks>();  // everything but the five just issued: iteration t has landed
                convert(t);
                split_step_barrier();
            }
            wait_vm_keep<0>();  // (nothing may land in a slot the next item's first requests target, or after the workgroup has ended)
        }
    } else {
        // ---- compute waves: k_conv2_wgrad_split's, on the seven-row ring and the three dy2 buffers ----
        const float gs = grad_scale(absmax);
        const int n = lane & 15, g = lane >> 4, cw = wv;
        const int ntaps = cw < 3 ? 4 : 3;
        uint32_t tapbase[4];
        int tapdy[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tp = min(cw + 8 * i, kTaps - 1), dz = tp / 9, dy = (tp / 3) % 3, dx = tp % 3;
            tapbase[i] = (uint32_t)(dz * R * kRowBytes + (dx == 1 ? 1024 : 0) + (dx == 2 ? 32 : 0)) + lane * 8;
            tapdy[i] = dy;
        }
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        h8 ones;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (_Float16)1.0f;
        for (int item = (int)blockIdx.x; item < nitems; item += (int)gridDim.x) {
            int b, oz0, oz1;
            if (!sample_plane_group(B, O2, kNP, b, oz0, oz1, item)) continue;
            const int np = oz1 - oz0;
            split_step_barrier();
            for (int t = 1; t <= nsteps; ++t) {
                const int oy = t - 1;
#ifdef WDMA_ABL_NOCOMP  // (measurement build: the compute waves only keep the barriers -- transport + conversion alone)
                if (false) {
#else
                if (oy < O2) {
#endif
                    uint32_t rowoff[3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) rowoff[dy] = (uint32_t)(((2 * oy + dy) % R) * kRowBytes);
                    const char *dybuf = dyst + (oy % wdma::kDyBufs) * kNP * 1024 + lane * 8;
                    // Every operand of a k-block is requested before its first MFMA, and the MFMAs run tap-interleaved (hh of every tap, then
                    // lh, then hl).  Each accumulator still receives its products in the same order as in k_conv2_wgrad_split.
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        if (2 * kb < np) {
                            const h8 bh = tr_pair(dybuf + (2 * kb) * 1024, dybuf + (2 * kb + 1) * 1024);
                            const h8 bl = tr_pair(dybuf + (2 * kb) * 1024 + 512, dybuf + (2 * kb + 1) * 1024 + 512);
                            h8 ah[4], al[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (i < 3 || ntaps == 4) {
                                    const uint32_t off = tapbase[i] + (tapdy[i] == 0 ? rowoff[0] : tapdy[i] == 1 ? rowoff[1] : rowoff[2]);
                                    const char *a0 = stage + (2 * (2 * kb)) * R * kRowBytes + off, *a1 = a0 + 2 * R * kRowBytes;
                                    ah[i] = tr_pair(a0, a1);
                                    al[i] = tr_pair(a0 + 512, a1 + 512);
                                }
                            }
#pragma unroll
                            for (int i = 0; i < 3; ++i) acc[i] = mfma_h(ah[i], bh, acc[i]);
                            if (ntaps == 4) acc[3] = mfma_h(ah[3], bh, acc[3]);
#pragma unroll
                            for (int i = 0; i < 3; ++i) acc[i] = mfma_lo(al[i], bh, acc[i]);
                            if (ntaps == 4) acc[3] = mfma_lo(al[3], bh, acc[3]);
#pragma unroll
                            for (int i = 0; i < 3; ++i) acc[i] = mfma_lo(ah[i], bl, acc[i]);
                            if (ntaps == 4) acc[3] = mfma_lo(ah[3], bl, acc[3]);
                            if (cw == 7) {
                                acc[3] = mfma_h(ones, bh, acc[3]);
                                acc[3] = mfma_lo(ones, bl, acc[3]);
                            }
                        }
                    }
                }
                split_step_barrier();
            }
        }
        const float unscale = (1.0f / kZScale) * inv_pow2(gs);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < ntaps) {
                const int tp = cw + 8 * i;
#pragma unroll
                for (int r = 0; r < 4; ++r) out[tp * 256 + (4 * g + r) * kC + n] = acc[i][r] * unscale;
            }
        }
        if (cw == 7 && g == 0) out[kTaps * 256 + n] = acc[3][0] * inv_pow2(gs);
    }
}


// ---------------------------------------------------------------------------
// conv2 data gradient + conv1 weight gradient (the fused backward, see k_conv2_dgrad_c1w in encoder.hip for the algebra and
// the outputs: T1[32 taps][16], S1, S2 per workgroup).  Workgroup = 16 waves = (sample, 4 plane pairs of the layer-1 volume),
// one per CU, walking the 16 row pairs c.  A step = the four 2 x 2 x 32-voxel super-tiles (a, c) of the workgroup.
//   waves 8-15 (staging): the 16 y1 rows of the NEXT step (8 planes x 2 rows, 32 KiB, each read from global memory exactly once,
//       contiguous 16-byte requests two steps ahead) go to LDS as fp32 with an 80-byte voxel stride -- the accumulator layout
//       reads 4 voxels of one channel per lane, and 80 bytes keep the two lane halves of a 32-lane LDS access on different
//       banks; the next dy2 row of the 5 output planes the workgroup touches goes to a 3-row ring as scaled f16 hi | lo
//       planes [voxel][16 channels] with a zero voxel in front and behind (ox = -1, 15) -- out-of-range planes / rows are zeros.
//   waves 0-7 (compute): wave = (plane pair, class set).  The eight parity classes (ez, ey, ex) of a super-tile need
//       8 + 4 + 4 + 4 + 2 + 2 + 2 + 1 taps = 14 k-steps of two taps x 16 channels; set X = classes {000, 011, 101, 110} and
//       set Y = {001, 010, 100, 111} hold 7 k-steps each (weights in registers: 2 x 7 fragments).  Per class: the k-steps'
//       3 split MFMAs each, then the epilogue of the fp32 kernel unchanged -- ReLU mask and xhat from the staged y1, the
//       channel sums, and the conv1 weight-gradient contraction T1 += x^T g on fp32 MFMAs with the int8 input slab
//       (wave-private, requested one step ahead).
// ---------------------------------------------------------------------------
namespace dsplit {
constexpr int kPairs = 4;
constexpr int kYVox = 80, kYHalf = 16 * kYVox, kYRow = 2 * kYHalf, kYBuf = 2 * 2 * kPairs * kYRow;  // 40 960 per step buffer
constexpr int kDyHalf = 18 * 32, kDyRow = 2 * kDyHalf, kDyPlanes = kPairs + 1, kDyRing = 3;
constexpr int kDyBytes = kDyPlanes * kDyRing * kDyRow;
// the int8 input under the workgroup's four plane pairs: 17 planes x 5 rows x 80 bytes per step, staged ONCE by the staging waves (round 4:
// each of the 12 compute waves used to fetch the 5 x 5 x 80-byte slab of its plane pair itself -- three waves per plane pair, the same
// 2 KB each: 24 KB of the 64 KB per step that went through the CU's vector-memory path, which is what bounds this kernel at its
// 10.8 B / clk; 6.8 KB now)
constexpr int kSlabPlanes = 4 * kPairs + 1, kSlabBuf = kSlabPlanes * 5 * 80;  // 6800 bytes per step parity
constexpr int kSlabPieces = kSlabPlanes * 5 * 4;  // 340 sixteen-byte pieces: x = 0 .. 63 of every row (byte 64 of a row is only read for voxels past the grid, which are masked)
constexpr int kSets = 3, kKSteps = 5;       // class sets per plane pair / k-steps per set (the image's stride)
constexpr int kConsWaves = kPairs * kSets;  // 12 compute waves
constexpr int kImgU4 = kSets * kKSteps * 2 * 64, kImgBytes = kImgU4 * 16;  // the B-operand image (all class sets): 30 720
constexpr int kImgSlotU4 = 2048;            // its slot in the encoder workspace (behind the forward image)
constexpr int kLdsBytes = 2 * kYBuf + kDyBytes + 2 * kSlabBuf + kImgBytes;
constexpr int kThreads = 1024, kProdThreads = 256;  // 12 compute + 4 staging waves, four per SIMD (<= 128 registers)
constexpr int kYSlots = 2 * 2 * kPairs * 128 / kProdThreads;  // 16-byte requests per staging thread and step: 8
constexpr int kDySlots = (kDyPlanes * 64 + kProdThreads - 1) / kProdThreads;  // 2
// set 0 = classes {000, 111} (4 + 1 k-steps), set 1 = {001, 010, 011} (2 + 2 + 1), set 2 = {100, 101, 110} (2 + 1 + 1): at most three
// class epilogues per wave (the epilogue's instruction issue bounds the kernel)
__host__ __device__ constexpr int ncls(int ty) { return ty == 0 ? 2 : 3; }
__host__ __device__ constexpr int cls(int ty, int ci) { return ty == 0 ? (ci == 0 ? 0 : 7) : ty == 1 ? 1 + ci : 4 + ci; }
__host__ __device__ constexpr int ntaps(int e) { return ((e & 4) ? 1 : 2) * ((e & 2) ? 1 : 2) * ((e & 1) ? 1 : 2); }
__host__ __device__ constexpr int ksteps(int e) { return (ntaps(e) + 1) / 2; }
// (closed form, NOT a recursion: the recursive version was not folded after unrolling -- a real, recursive function call per class)
__host__ __device__ constexpr int first_kstep(int ty, int ci) { return ty == 0 ? 4 * ci : ty == 1 ? 2 * ci : (ci == 0 ? 0 : 1 + ci); }
static_assert(first_kstep(0, 1) == ksteps(cls(0, 0)) && first_kstep(0, 1) + ksteps(cls(0, 1)) <= kKSteps, "prefix sums, set 0");
static_assert(first_kstep(1, 1) == ksteps(cls(1, 0)) && first_kstep(1, 2) == first_kstep(1, 1) + ksteps(cls(1, 1)) && first_kstep(1, 2) + ksteps(cls(1, 2)) <= kKSteps, "prefix sums, set 1");
static_assert(first_kstep(2, 1) == ksteps(cls(2, 0)) && first_kstep(2, 2) == first_kstep(2, 1) + ksteps(cls(2, 1)) && first_kstep(2, 2) + ksteps(cls(2, 2)) <= kKSteps, "prefix sums, set 2");
// tap q of class e: which neighbour (zo, yo, xo) of dy2 it reads, and its index in the 3 x 3 x 3 kernel
__host__ __device__ constexpr int tap_xo(int e, int q) { return q % ((e & 1) ? 1 : 2); }
__host__ __device__ constexpr int tap_yo(int e, int q) { return (q / ((e & 1) ? 1 : 2)) % ((e & 2) ? 1 : 2); }
__host__ __device__ constexpr int tap_zo(int e, int q) { return q / (((e & 1) ? 1 : 2) * ((e & 2) ? 1 : 2)); }
__host__ __device__ constexpr int tap_index(int e, int q)
{
    return (((e & 4) ? 1 : 2 * tap_zo(e, q)) * 3 + ((e & 2) ? 1 : 2 * tap_yo(e, q))) * 3 + ((e & 1) ? 1 : 2 * tap_xo(e, q));
}
}  // namespace dsplit

// dgrad B-operand image: [class set 2][k-step 7][hi | lo][lane = 16 g + n][j] = 2^10 W2[co = 8 (g & 1) + j][ci = n][tap(g >> 1)]
// with the two taps of the k-step from the class walk above (a missing second tap: zeros); + the power-of-two bound of
// sum |W2| over (co, tap) that scales the layer-1 gradient before the fp32 contraction is NOT needed (that part stays fp32).
__device__ __forceinline__ void prep_w2_dgrad_split_item(int i, const float *__restrict__ W2, uint4 *__restrict__ img)
{
    using namespace dsplit;
    const int ty = i / (kKSteps * 64), s = (i >> 6) % kKSteps, lane = i & 63, n = lane & 15, g = lane >> 4;
    int e = 0, ls = s;
    bool used = false;  // (set 2 uses four of its five k-step slots)
    for (int ci = 0; ci < ncls(ty); ++ci) {
        e = cls(ty, ci);
        if (ls < ksteps(e)) { used = true; break; }
        ls -= ksteps(e);
    }
    const int q = 2 * ls + (g >> 1);
    const bool has = used && q < ntaps(e);
    const int tap = has ? tap_index(e, q) : 0, c0 = 8 * (g & 1);
    h8 vh, vl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float wv = has ? W2[((size_t)(c0 + j) * kC + n) * kTaps + tap] * split::kWScale : 0.0f;
        _Float16 hi, lo;
        split::split2(wv, hi, lo);
        vh[j] = hi;
        vl[j] = lo;
    }
    img[((ty * kKSteps + s) * 2 + 0) * 64 + lane] = *reinterpret_cast<uint4 *>(&vh);
    img[((ty * kKSteps + s) * 2 + 1) * 64 + lane] = *reinterpret_cast<uint4 *>(&vl);
}

// definition of the hooks the conv1 forward kernels / k_bn1_analytic call (declared in encoder.hip in front of them): items
// first, first + stride, ...
// WIDE: the caller is a launch of its own on a critical path (k_bn1_analytic's extra workgroups, k_prep_w2_split_only), not a conv kernel
// that writes the images in passing (whose register budget the wide form of the bound sums would break)
template <bool WIDE>
__device__ __forceinline__ void prep_w2_split_items(const float *__restrict__ W2, float *__restrict__ w2img, int first, int stride)
{
    uint4 *img = reinterpret_cast<uint4 *>(w2img + 2 * kTaps * 256);  // EncWs::w2split
    constexpr int kImgItems = (split::kKSteps + dsplit::kSets * dsplit::kKSteps) * 64;
    for (int i = first; i < kImgItems + 128; i += stride) {
        if (i < split::kKSteps * 64)
            prep_w2_split_item(i, W2, img);
        else if (i < kImgItems)
            prep_w2_dgrad_split_item(i - split::kKSteps * 64, W2, img + split::kW2ImgU4);
        else {
            // sum |W2[co][ci][tap]| over the 16 co and the taps of parity class e: bounds |dz1| / max |dy2| for (ci, e); the data-gradient
            // kernel takes the maximum to scale the layer-1 gradient into f16 range for its second contraction
            const int j = i - kImgItems, ci = j & 15, e = j >> 4;
            const int nt = dsplit::ntaps(e);
            float a = 0.0f;
            if (WIDE) {
                // four taps = 64 loads requested before the first addition (same order of additions as the plain double loop, whose
                // load -> add chains were up to 128 dependent round trips on the critical path of the forward's first launch)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (4 * h < nt) {
                        float v[4][kC];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int tq = dsplit::tap_index(e, 4 * h + q < nt ? 4 * h + q : nt - 1);
#pragma unroll
                            for (int co = 0; co < kC; ++co) v[q][co] = W2[((size_t)co * kC + ci) * kTaps + tq];
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int co = 0; co < kC; ++co)
                                if (4 * h + q < nt) a += fabsf(v[q][co]);
                    }
                }
            } else {
                for (int q = 0; q < nt; ++q)
                    for (int co = 0; co < kC; ++co) a += fabsf(W2[((size_t)co * kC + ci) * kTaps + dsplit::tap_index(e, q)]);
            }
            reinterpret_cast<float *>(img + split::kW2ImgU4 + dsplit::kImgSlotU4)[j] = a;
        }
    }
}

__device__ void prep_w2_split_in_passing(const float *__restrict__ W2, float *__restrict__ w2img)
{
    prep_w2_split_items<false>(W2, w2img, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

template <int TY>
__device__ __forceinline__ void dgrad_split_supertile(
    const char *dyst, const char *ybuf, const int8_t *slab0, const int8_t *slab1, const uint4 *wimg /*this lane's column of the class set's image, in LDS*/,
    int ai, int c, bool z1ok, bool y0ok, bool y1ok, int O1, bool tok1, float sc, float sh,
    float gscale, float &s2, f32x4 &T1a, f32x4 &T1b)
{
    using namespace dsplit;
    const int lane = threadIdx.x & (kWave - 1), m = lane & 15, g = lane >> 4;
    const bool second = (g >> 1) != 0;
    h8 xs, xt, gh, gl;  // operands of the second contraction, filled class by class
    const _Float16 ones_row = m == (kTaps - 16) ? (_Float16)1.0f : (_Float16)0.0f;  // A row 27 (lane m = 11 of the second tap tile)
    uint32_t rowslot[2];  // ring slot of dy2 row c - yo
#pragma unroll
    for (int yo = 0; yo < 2; ++yo) rowslot[yo] = (uint32_t)((c - yo + kDyRing) % kDyRing);
    uint32_t a_lane = (uint32_t)((m + 1) * 32 + (g & 1) * 16);
    // (opaque to the optimiser: otherwise the 14 per-lane operand offsets and the row addresses are hoisted out of the step loop
    // as loop invariants, and the register allocator spills them -- any scratch use at all slows these kernels several times)
    asm volatile("" : "+v"(a_lane));
    asm volatile("" : "+s"(ai));  // (same for the scalar plane offsets: 28 hoisted SGPRs overflowed into a spill register that itself spilled)
    asm volatile("" : "+s"(O1));  // (and for the eight x-validity masks, two SGPRs each)
#pragma unroll
    for (int ci = 0; ci < ncls(TY); ++ci) {
        const int e = cls(TY, ci), ez = (e >> 2) & 1, ey = (e >> 1) & 1, ex = e & 1;
        // sub-tiles on out-of-grid planes / rows (the last plane pair / row pair only) are computed and masked, not skipped: a
        // branch here would end the basic block and keep the scheduler from overlapping one class's LDS latencies with the next
        const bool cls_ok = !((ez && !z1ok) || (ey ? !y1ok : !y0ok));
        f32x4 acc_hh = {0.f, 0.f, 0.f, 0.f}, acc_lh = acc_hh, acc_hl = acc_hh;
#pragma unroll
        for (int ls = 0; ls < ksteps(e); ++ls) {
            const int s = first_kstep(TY, ci) + ls, qa = 2 * ls, qb = min(2 * ls + 1, ntaps(e) - 1);  // (a missing second tap has zero weights)
            auto off = [&](int q) {
                return (uint32_t)((ai + 1 - tap_zo(e, q)) * kDyRing * kDyRow) + rowslot[tap_yo(e, q)] * kDyRow - (uint32_t)(tap_xo(e, q) * 32);
            };
            const uint32_t o = a_lane + (second ? off(qb) : off(qa));
            const h8 ah = *reinterpret_cast<const h8 *>(dyst + o), al = *reinterpret_cast<const h8 *>(dyst + o + kDyHalf);
            const uint4 uh = wimg[(s * 2 + 0) * 64], ul = wimg[(s * 2 + 1) * 64];
            const h8 wh = *reinterpret_cast<const h8 *>(&uh), wl = *reinterpret_cast<const h8 *>(&ul);
            acc_hh = split::mfma_h(ah, wh, acc_hh);
            acc_lh = split::mfma_lo(al, wh, acc_lh);
            acc_hl = split::mfma_lo(ah, wl, acc_hl);
        }
        const f32x4 raw = acc_hh + (acc_lh + acc_hl);  // D[i = voxel 4g + r][j = ci = m], scaled by gs 2^10
        uint32_t ylane = (uint32_t)((4 * g) * kYVox + m * 4);
        asm volatile("" : "+v"(ylane));
        const char *yrow = ybuf + ((2 * ai + ez) * 2 + ey) * kYRow + ex * kYHalf + ylane;
        const int kOff = ((2 * ez) * 5 + 2 * ey) * kSlabRow + 2 * ex;
        float gsv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // (the epilogue of dgrad_c1w_subtile: out-of-grid voxels are masked in g, so they never reach the sums or T1)
            const bool ok = cls_ok && 2 * (4 * g + r) + ex < O1;  // x validity of voxel 4g + r (x = 2j + ex)
            const float y = ok ? *reinterpret_cast<const float *>(yrow + r * kYVox) : 0.0f;
            gsv[r] = (ok && fmaf(sc, y, sh) > 0.0f) ? raw[r] * gscale : 0.0f;  // g, scaled into f16 range
            s2 = fmaf(gsv[r], y, s2);                                          // sum g y: S2 = rstd (sum g y - mean S1), after the loop
            xs[(ci & 1) * 4 + r] = (_Float16)(short)slab0[kOff + 4 * r];
            // second tap tile: taps 16 .. 26, and a ROW OF ONES at index 27 -- its row of T1 is sum g = S1, for free
            xt[(ci & 1) * 4 + r] = tok1 ? (_Float16)(short)slab1[kOff + 4 * r] : ones_row;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            _Float16 a, b;
            split::split2(gsv[r], a, b);
            gh[(ci & 1) * 4 + r] = a;
            gl[(ci & 1) * 4 + r] = b;
        }
        if (!(ci & 1) && ci == ncls(TY) - 1) {  // a set's unpaired last class: the second half of the k = 32 stays empty
#pragma unroll
            for (int r = 4; r < 8; ++r) xs[r] = xt[r] = gh[r] = gl[r] = (_Float16)0.0f;
        }
        if ((ci & 1) || ci == ncls(TY) - 1) {
            // conv1 weight gradient: T1[tap][ci] += sum over the 2 x 16 voxels of the class pair of x[voxel, tap] g[voxel][ci], on the
            // f16 pipe as well: x in {-1, 0, 1} is exact in f16, g is split (scaled by `gscale` into f16 range); k = 32 =
            // [class P voxels 4g .. 4g+3, class Q voxels 4g .. 4g+3] for both operands.  A[i = tap][k], B[k][j = ci].
            T1a = split::mfma_h(xs, gh, T1a);
            T1a = split::mfma_lo(xs, gl, T1a);
            T1b = split::mfma_h(xt, gh, T1b);
            T1b = split::mfma_lo(xt, gl, T1b);
        }
    }
}

__device__ __forceinline__ void conv2_dgrad_c1w_split_body(
    const float *__restrict__ dy2, const uint4 *__restrict__ w2img /*prep_w2_dgrad_split_item*/, const float *__restrict__ wbound /*[8 classes][16 ci]*/,
    const unsigned *__restrict__ absmax, const float *__restrict__ y1,
    const float *__restrict__ scale1, const float *__restrict__ shift1, const float *__restrict__ mean1, const float *__restrict__ rstd1,
    const int8_t *__restrict__ grid_i8, const int64_t *__restrict__ rows, int64_t grid_row_stride, int B, int G, int O1, int O2,
    float *__restrict__ partial /*[blocks][kE1F]*/, const int vblock)
{
    using namespace dsplit;
    extern __shared__ __attribute__((aligned(16))) char split_lds[];
    char *ybufs = split_lds, *dyst = split_lds + 2 * kYBuf, *slabs = dyst + kDyBytes;
    uint4 *wlds = reinterpret_cast<uint4 *>(slabs + 2 * kSlabBuf);
    static_assert(kSlabRow == 80 && kSlabBuf % 16 == 0 && kSlabPieces <= kConsWaves * kWave, "one 16-byte slab request per compute lane");
    const int NA = (O1 + 1) >> 1;  // 16 plane pairs / row pairs / voxels per x parity
    int b, a0, a1;
    const bool live = sample_plane_group(B, NA, kPairs, b, a0, a1, vblock);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int m = lane & 15, kq = lane >> 4;
    float s1 = 0.f, s2 = 0.f, t1_unscale = 0.0f, g_unscale = 0.0f;
    f32x4 T1a = {0.f, 0.f, 0.f, 0.f}, T1b = T1a;
    // (stale LDS may hold NaN patterns: the zero voxels around the dy2 rows and the rows of steps not yet staged must be finite)
    for (int i = tid; i < (kLdsBytes - kImgBytes) / 16; i += kThreads) reinterpret_cast<uint4 *>(split_lds)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < kImgBytes / 16; i += kThreads) wlds[i] = w2img[i];
    __syncthreads();
    const int nsteps = NA;  // 16 row pairs (even)
    if (live && wv >= kConsWaves) {
        // ---- staging waves ----
        const int ptid = tid - kConsWaves * kWave;
        const float gs = grad_scale(absmax);
        const int P2 = O2 * O2 * O2;
        const uint32_t rowC = 2 * 16 * kC, planeC = rowC * O1;
        const uint32_t within = ptid & 127;
        const float *ybase = y1 + (size_t)b * O1 * planeC + within * 4;
        const uint32_t yst = (within >> 6) * kYHalf + ((within >> 2) & 15) * kYVox + (within & 3) * 16;
        // dy2: request k of a thread -> piece f = 256 k + ptid = (plane a0 - 1 + (f >> 6), voxel, channel quad); pieces past the
        // fifth plane are duplicates that are never stored
        const int dpiece = ptid & 63, dvox = dpiece >> 2;
        auto dy_req = [&](int k, int c) {
            const int doz = a0 - 1 + min(4 * k + (ptid >> 6), kDyPlanes - 1);
            return *reinterpret_cast<const float4 *>(dy2 + ((size_t)b * P2 + ((size_t)min(max(doz, 0), O2 - 1) * O2 + min(max(c, 0), O2 - 1)) * O2 + min(dvox, O2 - 1)) * kC +
                                                     4 * (dpiece & 3));
        };
        auto dy_store = [&](int k, const float4 &d, int c) {
            const int dpl = 4 * k + (ptid >> 6), doz = a0 - 1 + dpl;
            if (dpl < kDyPlanes) {  // (wave-uniform; no request inside)
                const float v[4] = {d.x, d.y, d.z, d.w};
                const bool ok = doz >= 0 && doz < O2 && dvox < O2 && c >= 0 && c < O2;
                h4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 a, c2;
                    split::split2(ok ? v[e] * gs : 0.0f, a, c2);
                    hi[e] = a;
                    lo[e] = c2;
                }
                char *dst = dyst + (dpl * kDyRing + (c + kDyRing) % kDyRing) * kDyRow + (dvox + 1) * 32 + (dpiece & 3) * 8;
                *reinterpret_cast<h4 *>(dst) = hi;
#ifndef SPLIT_HI_ONLY
                *reinterpret_cast<h4 *>(dst + kDyHalf) = lo;
#endif
            }
        };
        static_assert(kDySlots == 2, "two dy2 requests per staging thread");
        // the inputs of step c: y1 rows 2c, 2c+1 of planes 2 a0 .. 2 a0 + 7; dy2 row c
        auto y_req = [&](int k, int c) {
            const int rowid = 2 * k + (ptid >> 7), pl = 2 * a0 + (rowid >> 1), row = 2 * c + (rowid & 1);  // (wave-uniform)
            return ld4_nt(ybase + (uint32_t)min(pl, O1 - 1) * planeC + (uint32_t)min(max(row, 0), O1 - 1) * rowC);
        };
        auto y_store = [&](int k, const float4 &v, int c) { *reinterpret_cast<float4 *>(ybufs + (c & 1) * kYBuf + (2 * k + (ptid >> 7)) * kYRow + yst) = v; };
        static_assert(kYSlots == 8, "eight y1 requests per staging thread");
        struct StepRegs { float4 y0, y1, y2, y3, y4, y5, y6, y7, d0, d1; };  // (named members: as arrays these 40 registers ended up in scratch memory)
#define GNBV_DS_LOAD(R, C)                                                                                                      \
    {                                                                                                                           \
        R.y0 = y_req(0, (C)); R.y1 = y_req(1, (C)); R.y2 = y_req(2, (C)); R.y3 = y_req(3, (C));                                   \
        R.y4 = y_req(4, (C)); R.y5 = y_req(5, (C)); R.y6 = y_req(6, (C)); R.y7 = y_req(7, (C));                                   \
        R.d0 = dy_req(0, (C)); R.d1 = dy_req(1, (C));                                                                             \
    }
#define GNBV_DS_STORE(R, C)                                                                                                     \
    {                                                                                                                           \
        y_store(0, R.y0, (C)); y_store(1, R.y1, (C)); y_store(2, R.y2, (C)); y_store(3, R.y3, (C));                               \
        y_store(4, R.y4, (C)); y_store(5, R.y5, (C)); y_store(6, R.y6, (C)); y_store(7, R.y7, (C));                               \
        dy_store(0, R.d0, (C)); dy_store(1, R.d1, (C));                                                                           \
    }
        StepRegs ra, rb;
#ifdef DSPLIT_ABL_NOSTAGE  // (measurement build: no y1 / dy2 requests after the first two sets -- the contraction alone on stale rows)
#undef GNBV_DS_LOAD
#define GNBV_DS_LOAD(R, C) { if ((C) < 2) { R.y0 = y_req(0, (C)); R.y1 = y_req(1, (C)); R.y2 = y_req(2, (C)); R.y3 = y_req(3, (C)); R.y4 = y_req(4, (C)); R.y5 = y_req(5, (C)); R.y6 = y_req(6, (C)); R.y7 = y_req(7, (C)); R.d0 = dy_req(0, (C)); R.d1 = dy_req(1, (C)); } }
#endif
        GNBV_DS_LOAD(ra, 0);
        GNBV_DS_LOAD(rb, 1);
        GNBV_DS_STORE(ra, 0);
        GNBV_DS_LOAD(ra, 2);
        split_step_barrier();
        // step c: compute reads y1 buffer c & 1 and dy2 rows c - 1, c; staging fills buffer (c + 1) & 1 and dy2 row c + 1
        // (branch-free around the requests: see the forward kernel).  rb holds odd steps, ra even ones.
        for (int c = 0; c < nsteps; c += 2) {
            GNBV_DS_STORE(rb, c + 1);
            GNBV_DS_LOAD(rb, c + 3);
            split_step_barrier();
            GNBV_DS_STORE(ra, c + 2);
            GNBV_DS_LOAD(ra, c + 4);
            split_step_barrier();
        }
#undef GNBV_DS_LOAD
#undef GNBV_DS_STORE
    } else if (live) {
        // ---- compute waves ----
        const int cw = wv, ai = cw / kSets, ty = cw - ai * kSets, a = a0 + ai;
        const float gs = grad_scale(absmax);
        // |raw| <= 2^14 (scaled max |dy2|) x 2^10 x max over (ci, class) of sum |W2|: scale it back under 2^14 for the f16 split
        float wb = fmaxf(wbound[lane], wbound[64 + lane]);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) wb = fmaxf(wb, __shfl_xor(wb, d, 64));
        int we = 0;
        (void)frexpf(fmaxf(wb, 1.0e-30f), &we);  // wb < 2^we
        we = __builtin_amdgcn_readfirstlane(we);
        const float gscale = ldexpf(1.0f, -(10 + we));
        t1_unscale = ldexpf(1.0f, we) * inv_pow2(gs);
        g_unscale = t1_unscale;  // (g was scaled by gscale = 2^-(10 + we) on top of gs 2^10)
        const uint4 *wimg = wlds + ty * kKSteps * 2 * 64 + lane;
        const float sc = scale1[m], sh = shift1[m];
        const bool tok1 = 16 + m < kTaps;
        const int t1 = tok1 ? 16 + m : 0;
        // the int8 input slab under super-tile (a, c): planes 4 ai .. 4 ai + 4 of the step's shared slab (staged by the staging waves)
        const int8_t *slab = reinterpret_cast<const int8_t *>(slabs) + ai * (4 * 5 * kSlabRow);
        const int8_t *slab0 = slab + ((m / 9) * 5 + (m / 3) % 3) * kSlabRow + m % 3 + 16 * kq;
        const int8_t *slab1 = slab + ((t1 / 9) * 5 + (t1 / 3) % 3) * kSlabRow + t1 % 3 + 16 * kq;
        // ... which the compute waves stage themselves, ONE 16-byte piece per lane and step, one step ahead (unconditional in every wave:
        // a wave-uniform branch around the request would make the compiler wait for it at the join)
        const int8_t *in = grid_i8 + (rows ? rows[b] : (int64_t)b) * grid_row_stride;
        const int g3m16 = G * G * G - 16;
        const int sp = min(cw * kWave + lane, kSlabPieces - 1), srow = sp >> 2, szr = srow / 5, syr = srow - 5 * szr;
        const int sbase = min(4 * a0 + szr, G - 1) * G * G + 16 * (sp & 3);
        char *sdst = slabs + srow * kSlabRow + 16 * (sp & 3);
        auto slab_req = [&](int c) { return ldu4_nt(in + min(sbase + min(4 * c + syr, G - 1) * G, g3m16)); };
        uint4 sv = slab_req(0);
        *reinterpret_cast<uint4 *>(sdst) = sv;
        sv = slab_req(1);
        const bool z1ok = 2 * a + 1 < O1;
        split_step_barrier();
        // (one step loop per class set: with the branch inside the loop the two instantiations' scalars overflow the SGPR file)
        auto run = [&](auto ty_c) {
            constexpr int TY = decltype(ty_c)::value;
            for (int c = 0; c < nsteps; ++c) {
                *reinterpret_cast<uint4 *>(sdst + ((c + 1) & 1) * kSlabBuf) = sv;  // step c + 1's piece (every wave left that buffer at the last barrier)
                sv = slab_req(c + 2);
                __builtin_amdgcn_sched_barrier(0);
                const bool y0ok = 2 * c < O1, y1ok = 2 * c + 1 < O1;
                const char *ybuf = ybufs + (c & 1) * kYBuf;
                const int sboff = (c & 1) * kSlabBuf;
#ifndef DSPLIT_ABL_NOCOMP  // (measurement build: the compute waves keep their slab requests and the barriers only -- the staging alone)
                dgrad_split_supertile<TY>(dyst, ybuf, slab0 + sboff, slab1 + sboff, wimg, ai, c, z1ok, y0ok, y1ok, O1, tok1, sc, sh, gscale, s2, T1a, T1b);
#endif
                split_step_barrier();
            }
        };
        if (ty == 0)
            run(std::integral_constant<int, 0>{});
        else if (ty == 1)
            run(std::integral_constant<int, 1>{});
        else
            run(std::integral_constant<int, 2>{});
    }
    // ---- workgroup-level sums: binary tree over the 16 waves (the staging waves add zeros; fixed order -> deterministic) ----
    T1a *= t1_unscale;
    T1b *= t1_unscale;
    // S1 = row 27 of T1 (the ones row): D[i = tap 16 + 4 kq + r][j = ci = m] -> lanes kq = 2, r = 3; S2 = rstd (sum g y - mean S1)
    s1 = __shfl(T1b[3], 32 + m, 64);
    s2 = kgroup_sum(s2) * g_unscale;
    if (live && wv < kConsWaves) s2 = rstd1[m] * (s2 - mean1[m] * s1);
    if (kq != 0) s1 = 0.0f;  // (kgroup_sum below adds the four k-groups: keep one copy)
    s1 = kgroup_sum(s1);
    if (kq == 2) T1b[3] = 0.0f;  // (row 27 is not a tap)
    __syncthreads();  // every wave is done with the staged rows (reused as the reduction buffer)
    constexpr int kSlot = 2 * kWave * 4 + 2 * kC;  // floats per wave slot
    float *red = reinterpret_cast<float *>(split_lds);
#pragma unroll
    for (int half = kThreads / kWave / 2; half >= 1; half >>= 1) {
        if (wv >= half && wv < 2 * half) {
            float *slot = red + (wv - half) * kSlot;
            reinterpret_cast<f32x4 *>(slot)[lane] = T1a;
            reinterpret_cast<f32x4 *>(slot)[kWave + lane] = T1b;
            if (lane < kC) {
                slot[2 * kWave * 4 + lane] = s1;
                slot[2 * kWave * 4 + kC + lane] = s2;
            }
        }
        __syncthreads();
        if (wv < half) {
            const float *slot = red + wv * kSlot;
            T1a += reinterpret_cast<const f32x4 *>(slot)[lane];
            T1b += reinterpret_cast<const f32x4 *>(slot)[kWave + lane];
            s1 += slot[2 * kWave * 4 + m];
            s2 += slot[2 * kWave * 4 + kC + m];
        }
        __syncthreads();
    }
    float *fin = red + 16 * kSlot;
    if (wv == 0) {  // final layout: [tap][co], S1, S2
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            fin[(4 * kq + r) * kC + m] = T1a[r];
            fin[(16 + 4 * kq + r) * kC + m] = T1b[r];
        }
        if (lane < kC) {
            fin[512 + lane] = s1;
            fin[512 + kC + lane] = s2;
        }
    }
    __syncthreads();
    float *out = partial + (size_t)vblock * kE1F;
    for (int o = tid; o < kE1F; o += kThreads) out[o] = fin[o];
}

__global__ __launch_bounds__(dsplit::kThreads) void k_conv2_dgrad_c1w_split(
    const float *__restrict__ dy2, const uint4 *__restrict__ w2img /*prep_w2_dgrad_split_item*/, const float *__restrict__ wbound /*[8 classes][16 ci]*/,
    const unsigned *__restrict__ absmax, const float *__restrict__ y1,
    const float *__restrict__ scale1, const float *__restrict__ shift1, const float *__restrict__ mean1, const float *__restrict__ rstd1,
    const int8_t *__restrict__ grid_i8, const int64_t *__restrict__ rows, int64_t grid_row_stride, int B, int G, int O1, int O2,
    float *__restrict__ partial /*[blocks][kE1F]*/)
{
    conv2_dgrad_c1w_split_body(dy2, w2img, wbound, absmax, y1, scale1, shift1, mean1, rstd1, grid_i8, rows, grid_row_stride, B, G, O1, O2, partial,
                               (int)blockIdx.x);
}


// ---------------------------------------------------------------------------
// conv1 + BN1 + ReLU + conv2 in ONE kernel.  TRAIN = false (BatchNorm in eval mode: rollout, evaluation): the layer-1 activations
// never exist in global memory (2 x 488 MB per 256-env policy step otherwise: a write by conv1, a read by conv2).  TRAIN = true
// (BN1's batch statistics known before the launch, from the input autocorrelation: k_bn1_analytic): y1 is stored for the backward
// and the BN2 partial sums are left as by k_conv2_fwd_split, but conv2 does not read back what conv1 just wrote.
// Ring and compute arithmetic of k_conv2_fwd_split; 4 compute waves (two output planes each) + 12 staging waves that COMPUTE the
// two new z1 rows of the 9 planes instead of reading them: per step 36 tiles of 16 voxels x 16 channels, three per staging wave,
// four `v_mfma_f32_16x16x32_f16` each (A = W1 as f16 hi | lo x 2^10, M = channel, k = the 27 taps laid out as 32 + 32, see the
// staging waves; B = the input patch, exact in f16, N = voxel), then bias + BN1 + ReLU, x 2^8, split, and an 8-byte store per
// lane into the ring -- the accumulator layout (lane = voxel, 4 channels) is the ring's layout.  The input rows an iteration needs
// (19 planes x 5 rows x 64 voxels: 6 KiB of int8 instead of 36 KiB of y1) go through a double-buffered LDS slab, converted to f16
// by the staging store: iteration j is requested at step j - 2 (one 16-byte request per thread), stored at step j - 1 and
// computed at step j.  What bounds it and what was tried: profiles/r02_notes.md.
// ---------------------------------------------------------------------------
namespace fsplit {
constexpr int kInPlanes = 2 * split::kNPl + 1;  // 19 input planes under the 9 z1 planes
constexpr int kInRows = 5;                      // input rows 4j+2 .. 4j+6 under z1 rows 2j+1, 2j+2
constexpr int kInRowBytes = 128, kInPlaneBytes = kInRows * kInRowBytes;  // the slab holds the input as f16 (converted once, by the staging store)
constexpr int kInBuf = kInPlanes * kInPlaneBytes + 64;  // + slack: the padding voxel reads two values past its row (zeroed once: never NaN)
constexpr int kInPieces = kInPlanes * kInRows * 4;      // 16-byte pieces per iteration: 380 of the 768 staging threads
constexpr int kTilesPerStep = 2 * split::kNPl * 2;      // (plane, row, x parity): 36
constexpr int kCompWaves = 4, kStageWaves = 12;          // the staging arithmetic bounds the kernel: 12 waves x 3 tiles per step; each compute
                                                         // wave takes TWO output planes (same weights) with its half of the k-steps
constexpr int kTilesPerWave = kTilesPerStep / kStageWaves;
static_assert(kTilesPerWave * kStageWaves == kTilesPerStep && kStageWaves % 4 == 0 && (kCompWaves + kStageWaves) * 64 == split::kThreads, "tile split");
constexpr float kW1Scale = 1024.0f;
constexpr int kLdsBytes = split::kStageBytes + split::kPadBytes + split::kRedBytes + 2 * kInBuf;
}  // namespace fsplit

__global__ void k_prep_w2_split_only(const float *__restrict__ W2, float *__restrict__ w2img)
{
    prep_w2_split_items<true>(W2, w2img, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

template <bool TRAIN>
__global__ __launch_bounds__(split::kThreads) void k_conv12_fwd_split(
    const int8_t *__restrict__ grid_i8, const int64_t *__restrict__ rows, int64_t grid_row_stride, const float *__restrict__ W1 /*[16][27]*/,
    const float *__restrict__ b1, const float *__restrict__ scale1, const float *__restrict__ shift1, int B, int G, int O1, int O2,
    const uint4 *__restrict__ w2img, const float *__restrict__ b2, float *__restrict__ y2, float *__restrict__ y1 /*TRAIN: stored for the backward*/,
    float *__restrict__ partials /*TRAIN: BN2 partial sums, one row per workgroup*/)
{
    using namespace split;
    using namespace fsplit;
    extern __shared__ __attribute__((aligned(16))) char split_lds[];
    char *stage = split_lds;
    float *red = reinterpret_cast<float *>(split_lds + split::kStageBytes + kPadBytes);
    char *inbuf = split_lds + split::kStageBytes + kPadBytes + kRedBytes;
    int b, oz0, oz1;
    const bool live = sample_plane_group(B, O2, kNP, b, oz0, oz1);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    if (!live) {
        if (TRAIN) write_partials(partials, kWaves, wv, 0.f, 0.f);
        return;
    }
    const int np = oz1 - oz0;
    const int nsteps = (O2 + 2) & ~1;
    float s_sum = 0.0f, s_sq = 0.0f;
    if (wv >= kCompWaves) {
        // ---- staging waves: conv1 + BN1 + ReLU on the fly ----
        const int ptid = tid - kCompWaves * kWave, pw = min(max(wv - kCompWaves, 0), kStageWaves - 1);
        const int n = lane & 15, g = lane >> 4;
        // The contraction over the 27 taps as k = 32 + 32: k-slot (g, e) of the first MFMA pair = (tap row r = 2g + (e >> 2) of the nine
        // (dz, dy) rows, dx = e & 3), of the second pair = row 8 for g = 0 -- dx = 3 and everything else in the second pair carry ZERO
        // weights.  A lane's four k-slots of a row are then 8 contiguous bytes of the f16 slab (x = 4n + 2 par + 0..3): two dword reads
        // per row, three rows per lane and tile, no conversion and no packing (the byte-wise gather of the int8 slab cost 8 reads with
        // bank conflicts + 12 VALU instructions per tile, and the LDS pipe is what these waves share with the compute waves).
        // A operand: W1[ch = n][tap] x 2^10, split
        h8 wh, wl, wh2, wl2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = 2 * g + (e >> 2), dx = e & 3;
            _Float16 hi, lo;
            split2(dx < 3 ? W1[n * kTaps + 3 * r + dx] * kW1Scale : 0.0f, hi, lo);
            wh[e] = hi;
            wl[e] = lo;
            split2(g == 0 && e < 3 ? W1[n * kTaps + 24 + e] * kW1Scale : 0.0f, hi, lo);
            wh2[e] = hi;
            wl2[e] = lo;
        }
        // LDS byte address of tap row r of voxel n of this wave's FIRST tile in input buffer 0.  Tile k of the wave is T = pw + 12k =
        // (plane pair (pw >> 2) + 3k, row (pw >> 1) & 1, x parity pw & 1): its taps sit 6k planes further on, buffer 1 kInBuf
        // further -- compile-time distances that go into the instructions' offset fields, so a tile costs no address arithmetic
        const int rsel = (pw >> 1) & 1, par = pw & 1;
        auto row_addr = [&](int r) {
            return (uint32_t)(split::kStageBytes + kPadBytes + kRedBytes + (r / 3 + 2 * (pw >> 2)) * kInPlaneBytes + (r % 3 + 2 * rsel) * kInRowBytes + 2 * (4 * n + 2 * par));
        };
        uint32_t tapaddr[3] = {row_addr(2 * g), row_addr(2 * g + 1), row_addr(8)};
        float bb[4], sc[4], sh[4];  // channels 4g .. 4g+3 of this lane's accumulator rows
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bb[r] = b1[4 * g + r];
            sc[r] = scale1[4 * g + r] * kZScale;
            sh[r] = shift1[4 * g + r] * kZScale;
            if (!TRAIN) {
                // inference never materialises y1 = acc / 2^10 + b: z = sc y + sh = (sc / 2^10) acc + (sc b + sh), one fma per value
                // (training keeps the two roundings: the backward recomputes the ReLU mask from the STORED y1)
                sh[r] = fmaf(sc[r], bb[r], sh[r]);
                sc[r] *= 1.0f / kW1Scale;
            }
        }
        // input staging: piece f = ptid < 380 -> (plane f / 20, row (f / 4) % 5, 16-byte quarter f % 4)
        const int f = min(ptid, kInPieces - 1), fpl = f / (kInRows * 4), frow = (f >> 2) % kInRows, fq = f & 3;
        const int8_t *in = grid_i8 + (rows ? rows[b] : (int64_t)b) * grid_row_stride;
        const int plane_g = min(4 * oz0 + fpl, G - 1);
        auto in_req = [&](int j) {  // input rows 4j+2 .. 4j+6 (clamped; UNCONDITIONAL)
            const int row_g = min(max(4 * j + 2 + frow, 0), G - 1);
            const int8_t *src = in + ((size_t)plane_g * G + row_g) * G + 16 * fq;
            if constexpr (TRAIN) {  // (opaque request: the caller waits with wait_vm_keep1, see st4_nt_masked)
                u4v_t v;
                ldu4_nt_async(v, src);
                return v;
            } else {
                const uint4 u = ldu4_nt(src);  // (the int8 input rows: -9.6 us per minibatch, -5.7 us per env step)
                return (u4v_t){u.x, u.y, u.z, u.w};
            }
        };
        // int8 -> f16 on the way into the slab, two values per v_perm_b32 + v_pk_add_f16: the byte b ^ 0x80 under the exponent byte
        // 0x64 is the f16 number 1024 + (b + 128), minus 1152 = b (exact: integers below 2048)
        auto in_store = [&](int j, const u4v_t &v) {
            const uint32_t w[4] = {v[0] ^ 0x80808080u, v[1] ^ 0x80808080u, v[2] ^ 0x80808080u, v[3] ^ 0x80808080u};
            uint32_t o[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const h2 bias = {(_Float16)1152.0f, (_Float16)1152.0f};
                const uint32_t p0 = __builtin_amdgcn_perm(0x64646464u, w[q], 0x04010400u), p1 = __builtin_amdgcn_perm(0x64646464u, w[q], 0x04030402u);
                const h2 f0 = *reinterpret_cast<const h2 *>(&p0) - bias, f1 = *reinterpret_cast<const h2 *>(&p1) - bias;
                o[2 * q] = *reinterpret_cast<const uint32_t *>(&f0);
                o[2 * q + 1] = *reinterpret_cast<const uint32_t *>(&f1);
            }
            if (ptid < kInPieces) {
                uint4 *dst = reinterpret_cast<uint4 *>(inbuf + (j & 1) * kInBuf + fpl * kInPlaneBytes + frow * kInRowBytes + 32 * fq);
                dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
                dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
            }
        };
        if (ptid < 8) reinterpret_cast<uint4 *>(inbuf + (ptid >> 2) * kInBuf + kInPlanes * kInPlaneBytes)[ptid & 3] = make_uint4(0, 0, 0, 0);  // the slack
        // y1 store (TRAIN): lane part of the address -- voxel x = 2n + par of a row, channels 4g .. 4g+3.  x = 31 IS the padding slot of
        // the odd half row (which no kernel reads as data); tiles outside the volume / of the neighbour's plane go there as well.
        const uint32_t y1_lane = (uint32_t)((par * 16 + n) * kC + 4 * g), y1_pad = (uint32_t)((16 + 15) * kC + 4 * g);
        auto compute = [&](int j, const int jp /* = j & 1, a literal at every call */) {
            auto gather = [&](int k, h8 &xa, h8 &xc) {
                uint32_t d[6];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(split_lds + tapaddr[r] + (uint32_t)(jp * kInBuf + 6 * k * kInPlaneBytes));
                    d[2 * r] = src[0];
                    d[2 * r + 1] = src[1];
                }
                const uint4 ua = make_uint4(d[0], d[1], d[2], d[3]), uc = make_uint4(d[4], d[5], d[4], d[5]);
                xa = *reinterpret_cast<const h8 *>(&ua);
                xc = *reinterpret_cast<const h8 *>(&uc);
            };
            // software pipeline: the input bytes of tile k + 1 are requested between the MFMAs of tile k and its epilogue -- the
            // compiler cannot move an LDS read above the previous tile's ring stores itself (same LDS array)
            const int row = 2 * j + 1 + rsel, slot = (row + kRing) % kRing;
            // (opaque once per step, in place: otherwise the 3 x 3 x 2 sums base + distance are hoisted out of the step loop as values
            // of their own and spill, instead of being folded into the reads' offset fields)
#pragma unroll
            for (int e = 0; e < 3; ++e) asm volatile("" : "+v"(tapaddr[e]));
            h8 xb, xb2;
            gather(0, xb, xb2);
#pragma unroll
            for (int k = 0; k < kTilesPerWave; ++k) {
                const int pi = (pw >> 2) + 3 * k;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = mfma_h(wh, xb, acc);  // D[i = channel 4g + r][j = voxel n]
                acc = mfma_h(wh2, xb2, acc);
                acc = mfma_lo(wl, xb, acc);
                acc = mfma_lo(wl2, xb2, acc);
                if (k + 1 < kTilesPerWave) gather(k + 1, xb, xb2);
                h4 hi, lo;
                float yv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = TRAIN ? acc[r] * (1.0f / kW1Scale) + bb[r] : acc[r];
                    yv[r] = y;
                    _Float16 a, c2;
                    split2(__builtin_amdgcn_fmed3f(fmaf(sc[r], y, sh[r]), 0.f, kZMax), a, c2);
                    hi[r] = a;
                    lo[r] = c2;
                }
                if (TRAIN) {
                    // the layer-1 pre-activations go to memory as well (k_conv1_fwd_split's store): every plane by ONE workgroup -- the
                    // plane two neighbouring groups share belongs to the upper one, except at the top of the volume.
                    // BRANCH-FREE: a conditional store makes the compiler's wait-count merge wait for the stores just issued when
                    // the next input piece is taken from its registers.  Row base in scalar registers, lane part precomputed.
                    // (bitwise, not && / ||: the short-circuit form becomes control flow with the two addresses in a stack array)
                    const int own = __builtin_amdgcn_readfirstlane((int)(row >= 0) & (int)(row < O1) & ((int)(pi < 2 * np) | ((int)(pi == 2 * np) & (int)(oz1 == O2))));
                    const uint32_t rowbase = vox1(b, min(2 * oz0 + pi, O1 - 1), min(max(row, 0), O1 - 1), 0, O1) * kC;
                    char *dst1 = reinterpret_cast<char *>(y1 + __builtin_amdgcn_readfirstlane(rowbase));
                    // all 64 lanes where the workgroup owns the tile; lane 0 alone, into the row's padding slot, where it does not
                    // (the first 32 / 64 samples' y1 stored WITHOUT the non-temporal hint, so that the backward finds a part of it in the
                    // memory-side cache: +1.4 / +7.3 us per minibatch, profiles/r05_ab_train_y1_cached_samples.json)
#ifdef FSPLIT_ABL_NOSTORE  // (measurement build: every y1 store under the one-lane mask -- the forward without its 244 MB write)
                    st4_nt_masked(dst1, 4 * y1_pad, (f32x4){yv[0], yv[1], yv[2], yv[3]}, 1u, 0u);
#else
                    st4_nt_masked(dst1, own ? 4 * y1_lane : 4 * y1_pad, (f32x4){yv[0], yv[1], yv[2], yv[3]}, own ? ~0u : 1u, own ? ~0u : 0u);
#endif
                }
                char *dst = stage + (pi * kRing + slot) * kRowBytes + par * 1024 + n * 32 + g * 8;
                *reinterpret_cast<h4 *>(dst) = hi;
#ifndef SPLIT_HI_ONLY
                *reinterpret_cast<h4 *>(dst + 512) = lo;
#endif
            }
        };
        // prologue: three barriers (the compute waves run the same count)
        // (TRAIN: explicit waits.  A wave's vector-memory operations complete in issue order; behind the request a wait releases there
        // are, in issue order: [the other register's request], then per half step 3 y1 stores + 1 request)
        // The counts follow from the issue order alone: compute() issues exactly kTilesPerWave y1 stores (st4_nt_masked: one per tile, owned
        // or not) and NOTHING else on the vector-memory counter; in_req() issues one request.  kVmHalf = what one half step puts behind a
        // request, kVmFull = what lies behind the OLDER register's request in the steady state.
        constexpr int kVmHalf = kTilesPerWave + 1, kVmFull = 2 * kTilesPerWave + 1;
        static_assert(kTilesPerWave == 3 && kVmHalf == 4 && kVmFull == 7, "the explicit vmcnt waits of the training forward's staging waves assume "
                      "kTilesPerWave y1 stores + one input request per half step (re-derive them when kNPl / kStageWaves change)");
        u4v_t ra = in_req(-1), rb = in_req(0);
        if constexpr (TRAIN) wait_vm_keep1<1>(ra);
        in_store(-1, ra);
        ra = in_req(1);
        split_step_barrier();
        compute(-1, 1);
        if constexpr (TRAIN) wait_vm_keep1<kVmHalf>(rb);  // ra's request + 3 stores
        in_store(0, rb);
        rb = in_req(2);
        split_step_barrier();
        compute(0, 0);
        if constexpr (TRAIN) wait_vm_keep1<kVmFull>(ra);  // 3 stores + rb's request + 3 stores: the steady state
        in_store(1, ra);
        ra = in_req(3);
        split_step_barrier();
        // step t: compute iteration t (slab t & 1), store iteration t + 1 (requested at step t - 1), request t + 3.  rb holds
        // even iterations, ra odd ones.  Branch-free around the requests.
        for (int t = 1; t <= nsteps; t += 2) {
            compute(t, 1);
            if constexpr (TRAIN) wait_vm_keep1<kVmFull>(rb);
            in_store(t + 1, rb);
            rb = in_req(t + 3);
            split_step_barrier();
            compute(t + 1, 0);
            if constexpr (TRAIN) wait_vm_keep1<kVmFull>(ra);
            in_store(t + 2, ra);
            ra = in_req(t + 4);
            split_step_barrier();
        }
    } else {
        // ---- compute waves: k_conv2_fwd_split's arithmetic; wave = (plane pair wv >> 1, k-half wv & 1), the two planes one after the
        // other with the same weight fragments (the BN2 partial sums only when training) ----
        const int m = lane & 15, g = lane >> 4;
        const int pl2 = wv >> 1, kh = wv & 1;
        h8 wh[kKHalf], wl[kKHalf];
#pragma unroll
        for (int s = 0; s < kKHalf; ++s) {
            const uint4 uh = w2img[((kh * kKHalf + s) * 2 + 0) * 64 + lane], ul = w2img[((kh * kKHalf + s) * 2 + 1) * 64 + lane];
            wh[s] = *reinterpret_cast<const h8 *>(&uh);
            wl[s] = *reinterpret_cast<const h8 *>(&ul);
        }
        const float bias = b2[m];
        const int P2 = O2 * O2 * O2;
        const bool second = (g >> 1) != 0;
        f32x4 prev0 = {0.f, 0.f, 0.f, 0.f}, prev1 = prev0;
        auto plane_step = [&](const int q, f32x4 &prev, int t) {
            const int oy = t - 1, pl = 2 * pl2 + q;
            const uint32_t a_lane = (uint32_t)(2 * pl * kRing * kRowBytes + m * 32 + (g & 1) * 16);
            if (kh == 0 && pl < np && oy >= 1 && oy - 1 < O2) {
                const f32x4 other = *reinterpret_cast<const f32x4 *>(red + (((t - 1) & 1) * kNP + pl) * 256 + lane * 4);
                const f32x4 acc = (prev + other) * (1.0f / (kZScale * kWScale));
                float *out = y2 + ((size_t)b * kC + m) * P2 + (size_t)(oz0 + pl) * O2 * O2 + (size_t)(oy - 1) * O2;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int oxi = 4 * g + rr;
                    if (oxi < O2) {
                        const float y = acc[rr] + bias;
                        out[oxi] = y;
                        if (TRAIN) {
                            s_sum += y;
                            s_sq += y * y;
                        }
                    }
                }
            }
            if (pl < np && oy < O2) {
                uint32_t rowoff[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) rowoff[dy] = (uint32_t)(((2 * oy + dy) % kRing) * kRowBytes);
                auto tap_off = [&](int tp) {
                    const int dz = tp / 9, dy = (tp / 3) % 3, dx = tp % 3;
                    return (uint32_t)(dz * kRing * kRowBytes + (dx == 1 ? 1024 : 0) + (dx == 2 ? 32 : 0)) + rowoff[dy];
                };
                auto a_off = [&](int s) {
                    const uint32_t ta = kh ? tap_off(2 * (kKHalf + s)) : tap_off(2 * s);
                    const uint32_t tb = kh ? tap_off(min(2 * (kKHalf + s) + 1, kTaps - 1)) : tap_off(2 * s + 1);
                    return a_lane + (second ? tb : ta);
                };
                f32x4 acc_hh = {0.f, 0.f, 0.f, 0.f}, acc_lh = acc_hh, acc_hl = acc_hh;
                constexpr int kAhead = 2;
                h8 ah[kAhead + 1], al[kAhead + 1];
#pragma unroll
                for (int s = 0; s < kAhead; ++s) {
                    ah[s] = *reinterpret_cast<const h8 *>(stage + a_off(s));
                    al[s] = *reinterpret_cast<const h8 *>(stage + a_off(s) + 512);
                }
#pragma unroll
                for (int s = 0; s < kKHalf; ++s) {
                    if (s + kAhead < kKHalf) {
                        ah[(s + kAhead) % (kAhead + 1)] = *reinterpret_cast<const h8 *>(stage + a_off(s + kAhead));
                        al[(s + kAhead) % (kAhead + 1)] = *reinterpret_cast<const h8 *>(stage + a_off(s + kAhead) + 512);
                    }
                    acc_hh = mfma_h(ah[s % (kAhead + 1)], wh[s], acc_hh);
                    acc_lh = mfma_lo(al[s % (kAhead + 1)], wh[s], acc_lh);
                    acc_hl = mfma_lo(ah[s % (kAhead + 1)], wl[s], acc_hl);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const f32x4 part = acc_hh + (acc_lh + acc_hl);
                if (kh)
                    *reinterpret_cast<f32x4 *>(red + ((t & 1) * kNP + pl) * 256 + lane * 4) = part;
                else
                    prev = part;
            }
        };
        split_step_barrier();
        split_step_barrier();
        split_step_barrier();
        for (int t = 1; t <= nsteps; ++t) {
            plane_step(0, prev0, t);
            plane_step(1, prev1, t);
            split_step_barrier();
        }
    }
    if (TRAIN) write_partials(partials, kWaves, wv, s_sum, s_sq);
}


// ---------------------------------------------------------------------------
// conv1 forward (training, y1 stored): the (first, byte-gather) staging arithmetic of k_conv12_fwd_split as a kernel of its own.
// Workgroup = (sample, output plane): the three int8 input planes under it (12 KiB, kept as int8) go to LDS once; 62 tiles of
// 16 voxels x 16 channels (31 rows x 2 x parities), two `v_mfma_f32_16x16x32_f16` each (W1 split x 2^10, the input exact)
// instead of seven fp32 MFMAs twice as long, + bias, and a 16-byte store per lane (4 channels of one voxel) into the
// x-parity-split y1 layout.  The fp32 kernel spends 25 us of a 72 us launch in the matrix pipe and holds a 48 KiB fp32 slab
// (3 workgroups per CU); this one is the 244 MB write.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kEncThreads) void k_conv1_fwd_split(
    const int8_t *__restrict__ grid_i8, const int64_t *__restrict__ rows, int64_t grid_row_stride, int B, int G, int O1, const float *__restrict__ W1,
    const float *__restrict__ b1, float *__restrict__ y1, const float *__restrict__ W2, float *__restrict__ w2img)
{
    using namespace split;
    prep_w2_in_passing(W2, w2img);
    __shared__ __attribute__((aligned(16))) int8_t s_in[3 * 64 * 64 + 64];  // (G = 64) + slack: the padding voxel reads past its row
    int b, oz;
    if (!sample_plane(B, O1, b, oz)) return;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int n = lane & 15, g = lane >> 4;
    const int8_t *in = grid_i8 + (rows ? rows[b] : (int64_t)b) * grid_row_stride + (size_t)(2 * oz) * G * G;
    {   // 3 planes x 4 KiB = 768 16-byte pieces, three per thread, all requested before the first is stored
        uint4 v[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) v[u] = *reinterpret_cast<const uint4 *>(in + 16 * (u * kEncThreads + tid));
#pragma unroll
        for (int u = 0; u < 3; ++u) reinterpret_cast<uint4 *>(s_in)[u * kEncThreads + tid] = v[u];
        if (tid < 4) reinterpret_cast<uint4 *>(s_in)[768 + tid] = make_uint4(0, 0, 0, 0);
    }
    h8 wh, wl;  // A operand: W1[ch = n][tap 8g + e] x 2^10, split
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int t = 8 * g + e;
        _Float16 hi, lo;
        split2(t < kTaps ? W1[n * kTaps + t] * fsplit::kW1Scale : 0.0f, hi, lo);
        wh[e] = hi;
        wl[e] = lo;
    }
    uint32_t tapoff[8];  // byte offset of tap 8g + e of voxel n relative to the tile's (row, parity) origin
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int t = min(8 * g + e, kTaps - 1);
        tapoff[e] = (uint32_t)((t / 9) * G * G + ((t / 3) % 3) * G + t % 3 + 4 * n);
    }
    const float4 bias = *reinterpret_cast<const float4 *>(b1 + 4 * g);
    __syncthreads();
    for (int T = wv; T < 2 * O1; T += kEncWaves) {  // tile = (row oy, x parity)
        const int oy = T >> 1, par = T & 1;
        const int8_t *origin = s_in + (2 * oy) * G + 2 * par;
        h8 xb;
#pragma unroll
        for (int e = 0; e < 8; ++e) xb[e] = (_Float16)(short)origin[tapoff[e]];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mfma_h(wh, xb, acc);  // D[i = channel 4g + r][j = voxel n]
        acc = mfma_lo(wl, xb, acc);
        const int x = 2 * n + par;
        if (x < O1) {
            const float s = 1.0f / fsplit::kW1Scale;
            *reinterpret_cast<float4 *>(y1 + (size_t)vox1(b, oz, oy, x, O1) * kC + 4 * g) =
                make_float4(acc[0] * s + bias.x, acc[1] * s + bias.y, acc[2] * s + bias.z, acc[3] * s + bias.w);
        }
    }
}
