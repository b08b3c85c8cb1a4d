// The conv2 kernels on the f16 matrix pipe with SPLIT operands for rows WIDER than 16 voxel slots per x parity (round 6; SURVEY row N1,
// BASELINE configs[4]: G = 128 -> O1 = 63, 32 slots per half row, O2 = 31).  Included by encoder.hip behind conv_split.h, whose arithmetic
// (split2, the three-product MFMA, the operand scalings), ring, roles and step structure these kernels keep.  Reference operators:
// gennbv/network/hybrid_encoder.py:31-38 (the conv stack is size-agnostic; only :40,83-84 hard-code 20^3).
//
// What changes against conv_split.h is the x direction only.  A workgroup handles an X TILE of 16 outputs ox = 16 tx + m of its four
// output planes; output m reads the even-parity input voxels 16 tx + m (tap dx = 0) and 16 tx + m + 1 (dx = 2) and the odd-parity voxel
// 16 tx + m (dx = 1).  With 15 valid outputs per row (G = 64) the sixteenth output is padding and so is the seventeenth even voxel it
// reads; with 16 valid outputs (tile 0 of a 31-output row) that voxel is DATA.  So the ring row here holds 17 voxels per plane:
//       row = [x parity 2][hi | lo][17 voxels][16 channels] f16  = 4 x 544 bytes   (conv_split.h: 4 x 512)
// staged from the WINDOW of slots 16 tx .. 16 tx + 16 (even) / .. + 15 (odd) of the y1 row; a staged voxel outside the row
// (x = 2 slot + parity >= O1) is stored as ZERO.  132 sixteen-byte pieces per row instead of 128, so a staging thread's pieces are no
// longer one row per wave pair: piece f = 512 k + thread of the step's 18 rows, (row, parity, voxel) per lane, precomputed once.
// Neighbouring tiles share one even voxel: 66 staged voxels per 63-voxel row (+ 5 %).  Everything a tile needs beyond that -- planes, rows,
// taps, the k-half hand-over, the BN2 partial sums -- is conv_split.h's, with 544 / 1088 / 2176 where that file has 512 / 1024 / 2048.
// The same kernels run G = 64 (one tile per row; GENNBV_SPLITX=1) -- which is how the G = 64 tests cover this file against the fp64 reference.
#pragma once
// The backward kernels come in two instantiations.  LOOP = false: one item per workgroup (gridDim.x >= nitems) -- the item "loop" runs
// once and the compiler knows it; LOOP = true: a workgroup walks items block, block + grid, ... and keeps its accumulators.  The loop
// form is the general one (any number of items into <= 512 partial rows) but costs registers where these kernels have none to spare:
// at G = 64 the data gradient takes 130.8 us with it and 100.4 without, the weight gradient 85.0 and 72.0 (296 / 152 bytes of scratch
// against none; profiles/r06_splitx_noloop_trace.txt), so the host launches LOOP = false whenever the partial buffers hold a row per item.
#define SPLITX_FOR_ITEMS for (int item = (int)blockIdx.x, once_ = 1; (LOOP || once_) && item < nitems; item += (int)gridDim.x, once_ = 0)

namespace splitx {
using split::kThreads; using split::kWaves; using split::kConsWaves; using split::kProdThreads; using split::kNP; using split::kNPl; using split::kRing;
using split::kPadBytes; using split::kRedBytes; using split::kDyBytes; using split::kKSteps; using split::kKHalf; using split::kZScale; using split::kWScale; using split::kZMax;
constexpr int kVox = 17;
constexpr int kPlaneB = kVox * 32, kParB = 2 * kPlaneB, kRowBytes = 2 * kParB;  // 544, 1088, 2176
constexpr int kStageBytes = kNPl * kRing * kRowBytes;                              // 97 920
constexpr int kEven = kVox * 4, kPieces = kEven + 16 * 4;                          // 16-byte fp32 pieces of a staged row: 68 even + 64 odd
constexpr int kStepPieces = 2 * kNPl * kPieces;                                    // 18 rows: 2 376
constexpr int kSlots = (kStepPieces + kProdThreads - 1) / kProdThreads;            // 5
constexpr int kLdsBytes = kStageBytes + kPadBytes + kRedBytes;
constexpr int kWgLdsBytes = kStageBytes + kPadBytes + kDyBytes;
static_assert(kSlots <= 8 && kPlaneB % 32 == 0, "flag bits / operand alignment");

// item -> (sample b, plane group [oz0, oz1), x tile tx); all items of a sample on XCD b % 8 (sample_plane_group's mapping with XT tiles
// per plane group)
__device__ __forceinline__ bool item_of(int B, int O2, int XT, int item, int &b, int &oz0, int &oz1, int &tx)
{
    const int ng = (O2 + kNP - 1) / kNP, per = ng * XT;
    const int xcd = item & 7, slot = item >> 3, gi = slot % per;
    b = (slot / per) * 8 + xcd;
    tx = gi % XT;
    oz0 = (gi / XT) * kNP;
    oz1 = min(O2, oz0 + kNP);
    return b < B;
}
static inline int items(int B, int O2, int XT) { return ((B + 7) / 8) * 8 * ((O2 + kNP - 1) / kNP) * XT; }

// The staging role: one of 512 threads that move the step's 18 new input rows (windows of 33 voxels) through registers into the ring as
// f16 hi | lo planes of z1 = 2^8 relu(bn1(y1)).  Requests are unconditional (clamped into the sample and the row); a thread whose piece
// index lies past the step's last piece requests a duplicate and stores nothing.
struct XStager {
    const float *ybase;
    uint32_t rowC, src_off[kSlots], lds_off[kSlots];
    uint32_t flags;  // bit k: slot k holds a piece; bit 8 + k: it belongs to the step's SECOND row; bit 16 + k: its voxel lies outside the row
    int O1;
    float sc[4], sh[4];
    __device__ __forceinline__ void init(const float *y1, const float *scale1, const float *shift1, int b, int oz0, int npl, int O1_, int x0, int ptid)
    {
        const int q = ptid & 3, XH = (O1_ + 1) >> 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            sc[s] = scale1[4 * q + s] * kZScale;
            sh[s] = shift1[4 * q + s] * kZScale;
        }
        O1 = O1_;
        rowC = 2 * XH * kC;
        const uint32_t planeC = rowC * O1, parC = XH * kC;
        flags = 0;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            const int f = k * kProdThreads + ptid, live = f < kStepPieces, ff = min(f, kStepPieces - 1);
            const int rs = ff / kPieces, within = ff - rs * kPieces;  // rs = 2 plane + (row & 1 ^ 1)
            const int par = within >= kEven, vox = (within - par * kEven) >> 2, slotx = x0 + vox;
            const int zero = 2 * slotx + par >= O1;
            src_off[k] = (uint32_t)min(rs >> 1, npl - 1) * planeC + par * parC + (uint32_t)min(slotx, XH - 1) * kC;
            lds_off[k] = (uint32_t)((rs >> 1) * kRing * kRowBytes + par * kParB + vox * 32 + q * 8);
            flags |= (uint32_t)live << k | (uint32_t)(rs & 1) << (8 + k) | (uint32_t)zero << (16 + k);
        }
        ybase = y1 + ((size_t)b * O1 + 2 * oz0) * planeC + q * 4;
    }
    __device__ __forceinline__ void load(float4 (&regs)[kSlots], int j) const
    {
        const uint32_t r0 = (uint32_t)min(max(2 * j + 1, 0), O1 - 1) * rowC, r1 = (uint32_t)min(max(2 * j + 2, 0), O1 - 1) * rowC;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) regs[k] = ld4_nt(ybase + src_off[k] + ((flags >> (8 + k)) & 1 ? r1 : r0));  // (y1 is streamed: read once per launch)
    }
    __device__ __forceinline__ void store(char *stage, const float4 (&regs)[kSlots], int j) const
    {
        const uint32_t s0 = (uint32_t)(((2 * j + 1 + kRing) % kRing) * kRowBytes), s1 = (uint32_t)(((2 * j + 2 + kRing) % kRing) * kRowBytes);
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            if (!((flags >> k) & 1)) continue;
            const bool zero = (flags >> (16 + k)) & 1;
            const float v[4] = {regs[k].x, regs[k].y, regs[k].z, regs[k].w};
            h4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 a, c2;
                split::split2(zero ? 0.0f : __builtin_amdgcn_fmed3f(fmaf(sc[e], v[e], sh[e]), 0.f, kZMax), a, c2);
                hi[e] = a;
                lo[e] = c2;
            }
            char *dst = stage + lds_off[k] + ((flags >> (8 + k)) & 1 ? s1 : s0);
            *reinterpret_cast<h4 *>(dst) = hi;
            *reinterpret_cast<h4 *>(dst + kPlaneB) = lo;
        }
    }
};
}  // namespace splitx

// ---------------------------------------------------------------------------
// conv2 forward, one x tile per workgroup (k_conv2_fwd_split's roles: waves 8-15 stage, waves 0-7 compute -- output plane = wave / 2, half
// of the 14 k-steps each, the odd wave's partial tile handed to the even one through LDS one step later).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(split::kThreads) void k_conv2_fwd_splitx(
    const float *__restrict__ y1, const float *__restrict__ scale1, const float *__restrict__ shift1, int B, int O1, int O2, int XT,
    const uint4 *__restrict__ w2img, const float *__restrict__ b2, float *__restrict__ y2, float *__restrict__ partials)
{
    using namespace splitx;
    extern __shared__ __attribute__((aligned(16))) char split_lds[];
    char *stage = split_lds;
    float *red = reinterpret_cast<float *>(split_lds + kStageBytes + kPadBytes);
    int b, oz0, oz1, tx;
    const bool live = item_of(B, O2, XT, (int)blockIdx.x, b, oz0, oz1, tx);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    if (!live) { write_partials(partials, kWaves, wv, 0.f, 0.f); return; }
    const int np = oz1 - oz0, npl = 2 * np + 1;
    float s_sum = 0.0f, s_sq = 0.0f;
    const int nsteps = (O2 + 2) & ~1;  // O2 compute steps + the deferred epilogue of the last row, rounded up to even
    if (wv >= kConsWaves) {
        // ---- staging waves ----
        XStager zs;
        zs.init(y1, scale1, shift1, b, oz0, npl, O1, 16 * tx, tid - kConsWaves * kWave);
        float4 ra[kSlots], rb[kSlots];
        zs.load(ra, -1);
        zs.load(rb, 0);
        zs.store(stage, ra, -1);
        zs.load(ra, 1);
        zs.store(stage, rb, 0);
        zs.load(rb, 2);
        split_step_barrier();
        for (int t = 1; t <= nsteps; t += 2) {  // ra holds odd iterations, rb even ones; branch-free around the requests
            zs.store(stage, ra, t);
            zs.load(ra, t + 2);
            split_step_barrier();
            zs.store(stage, rb, t + 1);
            zs.load(rb, t + 3);
            split_step_barrier();
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
        f32x4 prev = {0.f, 0.f, 0.f, 0.f};
        float *const out_base = y2 + ((size_t)b * kC + m) * P2 + (size_t)(oz0 + pl) * O2 * O2 + 16 * tx;
        split_step_barrier();
        for (int t = 1; t <= nsteps; ++t) {
            const int oy = t - 1;
            if (kh == 0 && pl < np && oy >= 1 && oy - 1 < O2) {
                const f32x4 other = *reinterpret_cast<const f32x4 *>(red + (((t - 1) & 1) * kNP + pl) * 256 + lane * 4);
                const f32x4 acc = (prev + other) * (1.0f / (kZScale * kWScale));
                float *out = out_base + (size_t)(oy - 1) * O2;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int oxi = 4 * g + rr;
                    if (16 * tx + oxi < O2) {
                        const float y = acc[rr] + bias;
                        out[oxi] = y;
                        s_sum += y;
                        s_sq += y * y;
                    }
                }
            }
            if (pl < np && oy < O2) {
                uint32_t rowoff[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) rowoff[dy] = (uint32_t)(((2 * oy + dy) % kRing) * kRowBytes);
                auto tap_off = [&](int tp) {
                    const int dz = tp / 9, dy = (tp / 3) % 3, dx = tp % 3;
                    return (uint32_t)(dz * kRing * kRowBytes + (dx == 1 ? kParB : 0) + (dx == 2 ? 32 : 0)) + rowoff[dy];
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
                    al[s] = *reinterpret_cast<const h8 *>(stage + a_off(s) + kPlaneB);
                }
#pragma unroll
                for (int s = 0; s < kKHalf; ++s) {
                    if (s + kAhead < kKHalf) {
                        ah[(s + kAhead) % (kAhead + 1)] = *reinterpret_cast<const h8 *>(stage + a_off(s + kAhead));
                        al[(s + kAhead) % (kAhead + 1)] = *reinterpret_cast<const h8 *>(stage + a_off(s + kAhead) + kPlaneB);
                    }
                    acc_hh = split::mfma_h(ah[s % (kAhead + 1)], wh[s], acc_hh);
                    acc_lh = split::mfma_lo(al[s % (kAhead + 1)], wh[s], acc_lh);
                    acc_hl = split::mfma_lo(ah[s % (kAhead + 1)], wl[s], acc_hl);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const f32x4 part = acc_hh + (acc_lh + acc_hl);
                if (kh)
                    *reinterpret_cast<f32x4 *>(red + ((t & 1) * kNP + pl) * 256 + lane * 4) = part;
                else
                    prev = part;
            }
            split_step_barrier();
        }
    }
    write_partials(partials, kWaves, wv, s_sum, s_sq);
}

// ---------------------------------------------------------------------------
// conv2 weight gradient over x tiles (k_conv2_wgrad_split's contraction: every tap belongs to one compute wave, operands through
// `ds_read_b64_tr_b16`, accumulators in registers).  A workgroup walks items blockIdx.x, + gridDim.x, ... and keeps its accumulators
// across them -- one partial row per WORKGROUP, so the partial buffer stays at <= 512 rows however many (sample, plane group, tile)
// items a minibatch has (2 048 at G = 128, batch 128).
// ---------------------------------------------------------------------------
template <bool LOOP>
__global__ __launch_bounds__(split::kThreads) void k_conv2_wgrad_splitx(
    const float *__restrict__ y1, const float *__restrict__ scale1, const float *__restrict__ shift1, const float *__restrict__ dy2 /*[B,O2^3,16]*/,
    const unsigned *__restrict__ absmax, int B, int O1, int O2, int XT, int nitems, float *__restrict__ partial /*[grid][27 * 256 + 16]*/)
{
    using namespace splitx;
    extern __shared__ __attribute__((aligned(16))) char split_lds[];
    char *stage = split_lds, *dyst = split_lds + kStageBytes + kPadBytes;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    constexpr int E2 = kTaps * 256 + kC;
    float *out = partial + (size_t)blockIdx.x * E2;
    // (stale LDS may hold NaN patterns; the ring is read next to slots not yet written in an item's first steps)
    for (int i = tid; i < kWgLdsBytes / 16; i += kThreads) reinterpret_cast<uint4 *>(split_lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const int P2 = O2 * O2 * O2;
    const int nsteps = (O2 + 1) & ~1;
    const float gs = grad_scale(absmax);
    // (the two roles walk the items in their own loops, in lock step through the step barriers: with one loop around both, the compute
    // waves' accumulators are live across the staging code as well, and the staging waves spill)
    if (wv >= kConsWaves) {
        // ---- staging waves ----
        const int ptid = tid - kConsWaves * kWave, pw = wv - kConsWaves;
        SPLITX_FOR_ITEMS {
            int b, oz0, oz1, tx;
            if (!item_of(B, O2, XT, item, b, oz0, oz1, tx)) continue;  // (workgroup-uniform)
            const int np = oz1 - oz0, npl = 2 * np + 1;
            XStager zs;
            zs.init(y1, scale1, shift1, b, oz0, npl, O1, 16 * tx, ptid);
            // dy2: thread -> (plane, voxel, channel quad) of the step's rows; the upper four waves request a duplicate they never store
            const int dpl = (ptid >> 6) & 3, dpiece = ptid & 63, dvox = dpiece >> 2, dxg = 16 * tx + dvox;
            const bool dvalid = dpl < np && dxg < O2;
            const float *dsrc = dy2 + ((size_t)b * P2 + (size_t)(oz0 + min(dpl, np - 1)) * O2 * O2 + min(dxg, O2 - 1)) * kC + 4 * (dpiece & 3);
            auto load_iter = [&](float4 (&regs)[kSlots], float4 &d, int j) {
                zs.load(regs, j);
                d = *reinterpret_cast<const float4 *>(dsrc + (size_t)min(max(j, 0), O2 - 1) * O2 * kC);
            };
            auto store_iter = [&](const float4 (&regs)[kSlots], const float4 &d, int j) {
                zs.store(stage, regs, j);
                if (pw < 4) {
                    const float v[4] = {d.x, d.y, d.z, d.w};
                    h4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 a, c2;
                        split::split2(dvalid ? v[e] * gs : 0.0f, a, c2);
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
            for (int t = 1; t <= nsteps; t += 2) {
                store_iter(ra, da, t);
                load_iter(ra, da, t + 2);
                split_step_barrier();
                store_iter(rb, db, t + 1);
                load_iter(rb, db, t + 3);
                split_step_barrier();
            }
        }
    } else {
        // ---- compute waves ----
        const int n = lane & 15, g = lane >> 4, cw = wv;
        const int ntaps = cw < 3 ? 4 : 3;
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        uint32_t tapbase[4];
        int tapdy[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tp = min(cw + 8 * i, kTaps - 1), dz = tp / 9, dy = (tp / 3) % 3, dx = tp % 3;
            tapbase[i] = (uint32_t)(dz * kRing * kRowBytes + (dx == 1 ? kParB : 0) + (dx == 2 ? 32 : 0)) + lane * 8;
            tapdy[i] = dy;
        }
        h8 ones;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (_Float16)1.0f;
        SPLITX_FOR_ITEMS {
            int b, oz0, oz1, tx;
            if (!item_of(B, O2, XT, item, b, oz0, oz1, tx)) continue;
            const int np = oz1 - oz0;
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
                                    const h8 ah = tr_pair(a0, a1), al = tr_pair(a0 + kPlaneB, a1 + kPlaneB);
                                    acc[i] = split::mfma_h(ah, bh, acc[i]);
                                    acc[i] = split::mfma_lo(al, bh, acc[i]);
                                    acc[i] = split::mfma_lo(ah, bl, acc[i]);
                                }
                            }
                            if (cw == 7) {
                                acc[3] = split::mfma_h(ones, bh, acc[3]);
                                acc[3] = split::mfma_lo(ones, bl, acc[3]);
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
// conv2 data gradient + conv1 weight gradient over x tiles: k_conv2_dgrad_c1w_split's workgroup (12 compute waves = 4 plane pairs x 3 class
// sets, 4 staging waves, the same super-tile arithmetic: dgrad_split_supertile is called unchanged with the row length counted from the
// tile's first voxel) for rows of more than 16 voxels per x parity.  A tile = layer-1 voxels x = 2 (16 tx + j) + ex, j = 0 .. 15:
//   * y1: the window of 16 slots per parity from slot 16 tx (the 80-byte-stride fp32 image of the step's 16 rows, as before);
//   * dy2: the class taps read output voxels 16 tx + j - xo, xo in {0, 1}: voxels 16 tx - 1 .. 16 tx + 15 -- position 0 of a staged dy2 row,
//     a structural zero at G = 64, is the LEFT NEIGHBOUR's last voxel for tx > 0.  It and position 17 are staged by the 40 threads of the
//     second dy2 request that used to fetch duplicates (5 planes x 2 sides x 4 channel quads), zero where outside the row;
//   * the int8 input slab: bytes 64 tx .. 64 tx + 79 of a row (FIVE 16-byte pieces: input x = 4 j + 2 ex + dx reaches 64 tx + 64 for a
//     valid voxel when the row goes on), 425 pieces over the 768 compute lanes.
// A workgroup walks items blockIdx.x, + gridDim.x, ... and keeps T1 / S1 / S2 across them (one partial row per workgroup); the dy2 ring is
// cleared between items (an item's row -1 is a ring slot nobody stages).
// ---------------------------------------------------------------------------
namespace dsplitx {
using namespace dsplit;
constexpr int kSlabPiecesX = kSlabPlanes * 5 * 5;  // 425
static_assert(kSlabPiecesX <= kConsWaves * kWave, "one slab request per compute lane");
__device__ __forceinline__ bool item_of(int B, int NA, int XT, int item, int &b, int &a0, int &a1, int &tx)
{
    const int ng = (NA + kPairs - 1) / kPairs, per = ng * XT;
    const int xcd = item & 7, slot = item >> 3, gi = slot % per;
    b = (slot / per) * 8 + xcd;
    tx = gi % XT;
    a0 = (gi / XT) * kPairs;
    a1 = min(NA, a0 + kPairs);
    return b < B;
}
static inline int items(int B, int NA, int XT) { return ((B + 7) / 8) * 8 * ((NA + kPairs - 1) / kPairs) * XT; }
}  // namespace dsplitx

template <bool LOOP>
__global__ __launch_bounds__(dsplit::kThreads) void k_conv2_dgrad_c1w_splitx(
    const float *__restrict__ dy2, const uint4 *__restrict__ w2img /*prep_w2_dgrad_split_item*/, const float *__restrict__ wbound /*[8 classes][16 ci]*/,
    const unsigned *__restrict__ absmax, const float *__restrict__ y1,
    const float *__restrict__ scale1, const float *__restrict__ shift1, const float *__restrict__ mean1, const float *__restrict__ rstd1,
    const int8_t *__restrict__ grid_i8, const int64_t *__restrict__ rows, int64_t grid_row_stride, int B, int G, int O1, int O2, int XT, int nitems,
    float *__restrict__ partial /*[grid][kE1F]*/)
{
    using namespace dsplitx;
    extern __shared__ __attribute__((aligned(16))) char split_lds[];
    char *ybufs = split_lds, *dyst = split_lds + 2 * kYBuf, *slabs = dyst + kDyBytes;
    uint4 *wlds = reinterpret_cast<uint4 *>(slabs + 2 * kSlabBuf);
    const int NA = (O1 + 1) >> 1, XH = NA;  // plane pairs / row pairs; voxel slots per x parity of a y1 row
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int m = lane & 15, kq = lane >> 4;
    float s1 = 0.f, s2 = 0.f, t1_unscale = 0.0f, g_unscale = 0.0f;
    f32x4 T1a = {0.f, 0.f, 0.f, 0.f}, T1b = T1a;
    for (int i = tid; i < (kLdsBytes - kImgBytes) / 16; i += kThreads) reinterpret_cast<uint4 *>(split_lds)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < kImgBytes / 16; i += kThreads) wlds[i] = w2img[i];
    __syncthreads();
    const int nsteps = (NA + 1) & ~1;  // row pairs, rounded up to even (the staging loop advances by two)
    const float gs = grad_scale(absmax);
    const int P2 = O2 * O2 * O2;
    bool first = true;
    SPLITX_FOR_ITEMS {
        int b, a0, a1, tx;
        if (!item_of(B, NA, XT, item, b, a0, a1, tx)) continue;  // (workgroup-uniform)
        const int x0 = 16 * tx;
        if (!first) {
            // the previous item's dy2 rows must not be read as this item's rows -1 / as planes it does not stage
            __syncthreads();
            for (int i = tid; i < kDyBytes / 16; i += kThreads) reinterpret_cast<uint4 *>(dyst)[i] = make_uint4(0, 0, 0, 0);
            __syncthreads();
        }
        first = false;
        if (wv >= kConsWaves) {
            // ---- staging waves ----
            const int ptid = tid - kConsWaves * kWave;
            const uint32_t rowC = 2 * XH * kC, planeC = rowC * O1, parC = XH * kC;
            const uint32_t within = ptid & 127, wpar = within >> 6, wvox = (within >> 2) & 15;
            const float *ybase = y1 + (size_t)b * O1 * planeC + wpar * parC + (uint32_t)min(x0 + (int)wvox, XH - 1) * kC + (within & 3) * 4;
            const uint32_t yst = wpar * kYHalf + wvox * kYVox + (within & 3) * 16;
            // dy2 request k = 0: plane ptid >> 6 (0 .. 3), voxel x0 + (piece >> 2) at position (piece >> 2) + 1; request k = 1: threads 0 .. 63 the
            // same for plane 4, threads 64 .. 103 the two halo voxels (positions 0 and 17) of the five planes
            const int dpiece = ptid & 63;
            const int hh = min(max(ptid - 64, 0), 39);
            const bool halo = ptid >= 64;
            const int dpl1 = halo ? hh >> 3 : 4, dpos1 = halo ? ((hh >> 2) & 1) * 17 : (dpiece >> 2) + 1, dq1 = halo ? hh & 3 : dpiece & 3;
            const bool dlive1 = ptid < 104;
            auto dy_req = [&](int k, int c) {
                const int dpl = k == 0 ? ptid >> 6 : dpl1, pos = k == 0 ? (dpiece >> 2) + 1 : dpos1, dq = k == 0 ? dpiece & 3 : dq1;
                const int doz = a0 - 1 + dpl, gx = x0 + pos - 1;
                return *reinterpret_cast<const float4 *>(dy2 + ((size_t)b * P2 + ((size_t)min(max(doz, 0), O2 - 1) * O2 + min(max(c, 0), O2 - 1)) * O2 + min(max(gx, 0), O2 - 1)) * kC + 4 * dq);
            };
            auto dy_store = [&](int k, const float4 &d, int c) {
                const int dpl = k == 0 ? ptid >> 6 : dpl1, pos = k == 0 ? (dpiece >> 2) + 1 : dpos1, dq = k == 0 ? dpiece & 3 : dq1;
                const int doz = a0 - 1 + dpl, gx = x0 + pos - 1;
                if (k == 0 || dlive1) {
                    const float v[4] = {d.x, d.y, d.z, d.w};
                    const bool ok = doz >= 0 && doz < O2 && gx >= 0 && gx < O2 && c >= 0 && c < O2;
                    h4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 a, c2;
                        split::split2(ok ? v[e] * gs : 0.0f, a, c2);
                        hi[e] = a;
                        lo[e] = c2;
                    }
                    char *dst = dyst + (dpl * kDyRing + (c + kDyRing) % kDyRing) * kDyRow + pos * 32 + dq * 8;
                    *reinterpret_cast<h4 *>(dst) = hi;
                    *reinterpret_cast<h4 *>(dst + kDyHalf) = lo;
                }
            };
            auto y_req = [&](int k, int c) {
                const int rowid = 2 * k + (ptid >> 7), pl = 2 * a0 + (rowid >> 1), row = 2 * c + (rowid & 1);  // (wave-uniform)
                return ld4_nt(ybase + (uint32_t)min(pl, O1 - 1) * planeC + (uint32_t)min(max(row, 0), O1 - 1) * rowC);
            };
            auto y_store = [&](int k, const float4 &v, int c) { *reinterpret_cast<float4 *>(ybufs + (c & 1) * kYBuf + (2 * k + (ptid >> 7)) * kYRow + yst) = v; };
            struct StepRegs { float4 y0, y1, y2, y3, y4, y5, y6, y7, d0, d1; };
#define GNBV_DX_LOAD(R, C)                                                                                                      \
    {                                                                                                                           \
        R.y0 = y_req(0, (C)); R.y1 = y_req(1, (C)); R.y2 = y_req(2, (C)); R.y3 = y_req(3, (C));                                   \
        R.y4 = y_req(4, (C)); R.y5 = y_req(5, (C)); R.y6 = y_req(6, (C)); R.y7 = y_req(7, (C));                                   \
        R.d0 = dy_req(0, (C)); R.d1 = dy_req(1, (C));                                                                             \
    }
#define GNBV_DX_STORE(R, C)                                                                                                     \
    {                                                                                                                           \
        y_store(0, R.y0, (C)); y_store(1, R.y1, (C)); y_store(2, R.y2, (C)); y_store(3, R.y3, (C));                               \
        y_store(4, R.y4, (C)); y_store(5, R.y5, (C)); y_store(6, R.y6, (C)); y_store(7, R.y7, (C));                               \
        dy_store(0, R.d0, (C)); dy_store(1, R.d1, (C));                                                                           \
    }
            StepRegs ra, rb;
            GNBV_DX_LOAD(ra, 0);
            GNBV_DX_LOAD(rb, 1);
            GNBV_DX_STORE(ra, 0);
            GNBV_DX_LOAD(ra, 2);
            split_step_barrier();
            for (int c = 0; c < nsteps; c += 2) {
                GNBV_DX_STORE(rb, c + 1);
                GNBV_DX_LOAD(rb, c + 3);
                split_step_barrier();
                GNBV_DX_STORE(ra, c + 2);
                GNBV_DX_LOAD(ra, c + 4);
                split_step_barrier();
            }
#undef GNBV_DX_LOAD
#undef GNBV_DX_STORE
        } else {
            // ---- compute waves ----
            const int cw = wv, ai = cw / kSets, ty = cw - ai * kSets, a = a0 + ai;
            float wb = fmaxf(wbound[lane], wbound[64 + lane]);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) wb = fmaxf(wb, __shfl_xor(wb, d, 64));
            int we = 0;
            (void)frexpf(fmaxf(wb, 1.0e-30f), &we);  // wb < 2^we
            we = __builtin_amdgcn_readfirstlane(we);
            const float gscale = ldexpf(1.0f, -(10 + we));
            t1_unscale = ldexpf(1.0f, we) * inv_pow2(gs);
            g_unscale = t1_unscale;
            const uint4 *wimg = wlds + ty * kKSteps * 2 * 64 + lane;
            const float sc = scale1[m], sh = shift1[m];
            const bool tok1 = 16 + m < kTaps;
            const int t1 = tok1 ? 16 + m : 0;
            const int8_t *slab = reinterpret_cast<const int8_t *>(slabs) + ai * (4 * 5 * kSlabRow);
            const int8_t *slab0 = slab + ((m / 9) * 5 + (m / 3) % 3) * kSlabRow + m % 3 + 16 * kq;
            const int8_t *slab1 = slab + ((t1 / 9) * 5 + (t1 / 3) % 3) * kSlabRow + t1 % 3 + 16 * kq;
            const int8_t *in = grid_i8 + (rows ? rows[b] : (int64_t)b) * grid_row_stride;
            const int g3m16 = G * G * G - 16;
            const int sp = min(cw * kWave + lane, kSlabPiecesX - 1), srow = sp / 5, spc = sp - 5 * srow, szr = srow / 5, syr = srow - 5 * szr;
            const int sbase = min(4 * a0 + szr, G - 1) * G * G + min(64 * tx + 16 * spc, G - 16);
            char *sdst = slabs + srow * kSlabRow + 16 * spc;
            auto slab_req = [&](int c) { return ldu4_nt(in + min(sbase + min(4 * c + syr, G - 1) * G, g3m16)); };
            uint4 sv = slab_req(0);
            *reinterpret_cast<uint4 *>(sdst) = sv;
            sv = slab_req(1);
            const bool z1ok = 2 * a + 1 < O1 && a < a1;
            const bool pair_ok = a < a1;  // (the last plane group of a sample may hold fewer than four pairs)
            const int O1x = O1 - 2 * x0;  // the row's length counted from the tile's first voxel: x = 2 (x0 + j) + ex < O1
            split_step_barrier();
            auto run = [&](auto ty_c) {
                constexpr int TY = decltype(ty_c)::value;
                for (int c = 0; c < nsteps; ++c) {
                    *reinterpret_cast<uint4 *>(sdst + ((c + 1) & 1) * kSlabBuf) = sv;
                    sv = slab_req(c + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    const bool y0ok = pair_ok && 2 * c < O1, y1ok = pair_ok && 2 * c + 1 < O1;
                    const char *ybuf = ybufs + (c & 1) * kYBuf;
                    const int sboff = (c & 1) * kSlabBuf;
                    dgrad_split_supertile<TY>(dyst, ybuf, slab0 + sboff, slab1 + sboff, wimg, ai, c, z1ok, y0ok, y1ok, O1x, tok1, sc, sh, gscale, s2, T1a, T1b);
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
    }
    // ---- workgroup-level sums: binary tree over the 16 waves (the staging waves add zeros; fixed order -> deterministic) ----
    T1a *= t1_unscale;
    T1b *= t1_unscale;
    s1 = __shfl(T1b[3], 32 + m, 64);
    s2 = kgroup_sum(s2) * g_unscale;
    if (wv < kConsWaves) s2 = rstd1[m] * (s2 - mean1[m] * s1);
    if (kq != 0) s1 = 0.0f;
    s1 = kgroup_sum(s1);
    if (kq == 2) T1b[3] = 0.0f;  // (row 27 is not a tap)
    __syncthreads();
    constexpr int kSlot = 2 * kWave * 4 + 2 * kC;
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
    if (wv == 0) {
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
    float *out = partial + (size_t)blockIdx.x * kE1F;
    for (int o = tid; o < kE1F; o += kThreads) out[o] = fin[o];
}
