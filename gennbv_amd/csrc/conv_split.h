// conv2 kernels on the f16 matrix pipe with SPLIT operands (included by encoder.hip; G = 64-class shapes: ceil(O1/2) == 16, O2 <= 15).
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
#pragma once

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

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
    lo = (_Float16)(x - (float)hi);
}
__device__ __forceinline__ f32x4 mfma_h(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
}  // namespace split

// W2 [co][ci][27] -> the B-operand images of the split kernels, as two f16 planes scaled by 2^10:
//   fwd   image [k-step s][hi|lo][lane = 16 g + n][j] = W2[co = n][ci = 8 (g & 1) + j][tap = 2 s + (g >> 1)]      (tap 27: zero)
//   dgrad image [k-step s][hi|lo][lane = 16 g + n][j] = W2[co = 8 (g & 1) + j][ci = n][tap = 2 s + (g >> 1)]
__global__ void k_prep_w2_split(const float *__restrict__ W2, uint4 *__restrict__ img_fwd, uint4 *__restrict__ img_dgrad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= split::kKSteps * 64) return;
    const int s = i >> 6, lane = i & 63, n = lane & 15, g = lane >> 4, tap = 2 * s + (g >> 1), c0 = 8 * (g & 1);
    h8 fh, fl, dh, dl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float wf = tap < kTaps ? W2[((size_t)n * kC + c0 + j) * kTaps + tap] * split::kWScale : 0.0f;
        const float wd = tap < kTaps ? W2[((size_t)(c0 + j) * kC + n) * kTaps + tap] * split::kWScale : 0.0f;
        _Float16 hi, lo;
        split::split2(wf, hi, lo);
        fh[j] = hi;
        fl[j] = lo;
        split::split2(wd, hi, lo);
        dh[j] = hi;
        dl[j] = lo;
    }
    img_fwd[(s * 2 + 0) * 64 + lane] = *reinterpret_cast<uint4 *>(&fh);
    img_fwd[(s * 2 + 1) * 64 + lane] = *reinterpret_cast<uint4 *>(&fl);
    if (img_dgrad != nullptr) {
        img_dgrad[(s * 2 + 0) * 64 + lane] = *reinterpret_cast<uint4 *>(&dh);
        img_dgrad[(s * 2 + 1) * 64 + lane] = *reinterpret_cast<uint4 *>(&dl);
    }
}

// The staging role shared by the split kernels: one of 512 threads (8 waves) that move the two new input rows of an iteration
// (rows 2j+1, 2j+2 of the workgroup's <= 9 input planes, 36 KiB of fp32 y1) through registers into the LDS ring as f16 hi | lo
// planes of z1 = 2^8 relu(bn1(y1)).  Request k of a thread: region rs = 4 k + (wave >> 1) = 2 plane + (row & 1 ^ 1), the 16-byte
// piece `within` of that 2 KiB row.  Everything that selects a request is wave-uniform and the requests are UNCONDITIONAL
// (clamped into the sample): a branch around a load makes the compiler wait for every load at the join.
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
            regs[k] = *reinterpret_cast<const float4 *>(ybase + (uint32_t)pi * planeC + (uint32_t)row * rowC);
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
            *reinterpret_cast<h4 *>(dst + 512) = lo;
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
                    acc_lh = mfma_h(al[s % (kAhead + 1)], wh[s], acc_lh);
                    acc_hl = mfma_h(ah[s % (kAhead + 1)], wl[s], acc_hl);
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
    return amax > 0.0f && amax < 3.0e38f ? ldexpf(1.0f, split::kGradBits - e) : 1.0f;
}

__global__ __launch_bounds__(split::kThreads) void k_conv2_wgrad_split(
    const float *__restrict__ y1, const float *__restrict__ scale1, const float *__restrict__ shift1, const float *__restrict__ dy2 /*[B,O2^3,16]*/,
    const unsigned *__restrict__ absmax, int B, int O1, int O2, float *__restrict__ partial /*[grid][27 * 256 + 16]*/)
{
    using namespace split;
    extern __shared__ __attribute__((aligned(16))) char split_lds[];
    char *stage = split_lds, *dyst = split_lds + split::kStageBytes + kPadBytes;
    int b, oz0, oz1;
    const bool live = sample_plane_group(B, O2, kNP, b, oz0, oz1);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    constexpr int E2 = kTaps * 256 + kC;
    float *out = partial + (size_t)blockIdx.x * E2;
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
                *reinterpret_cast<h4 *>(dst + 512) = lo;
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
                                acc[i] = mfma_h(al, bh, acc[i]);
                                acc[i] = mfma_h(ah, bl, acc[i]);
                            }
                        }
                        if (cw == 7) {
                            acc[3] = mfma_h(ones, bh, acc[3]);
                            acc[3] = mfma_h(ones, bl, acc[3]);
                        }
                    }
                }
            }
            split_step_barrier();
        }
        const float unscale = 1.0f / (kZScale * gs);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < ntaps) {
                const int tp = cw + 8 * i;
#pragma unroll
                for (int r = 0; r < 4; ++r) out[tp * 256 + (4 * g + r) * kC + n] = acc[i][r] * unscale;
            }
        }
        if (cw == 7 && g == 0) out[kTaps * 256 + n] = acc[3][0] * (1.0f / gs);
    }
}
