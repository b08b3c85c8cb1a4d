// voxel.hip -- GenNBV state encoding on MI355X (gfx950): depth -> world points ->
// voxel indices -> hit bitmask -> Bresenham ray cast -> probabilistic grid update.
//
// Replaces the reference's per-environment Python loop
//   gennbv/env/env_train_gennbv.py:277-326 (update_occ_grid), :494-533 (back_projection_fg),
//   gennbv/utils.py:24-227 (bresenham3D_pycuda), :230-270, :273-306, :309-325,
//   gennbv/env/env_train_base.py:513-534 (post_process_camera_tensor)
// with three launches per environment step for the whole batch:
//
//   k_hit_mask     N x chunks workgroups.  Streams the raw depth+seg batch once
//                  (16 B/lane coalesced), back-projects in the canonical fp32 order,
//                  and ORs voxel bits into an LDS-resident G^3-bit mask
//                  (64^3 bits = 32 KiB); non-zero words are flushed with one
//                  device-scope atomicOr each.  HBM-bound: H*W*8 B per env.
//   k_raycast      N x S workgroups of 1024 threads.  Compacts the set bits of the
//                  hit mask into an LDS queue and walks one integer Bresenham ray per
//                  lane, OR-ing the visited voxels into an LDS path mask.
//                  Integer-ALU / LDS-atomic bound, touches ~2 * G^3/8 B of HBM.
//   k_grid_update  streaming pass over prob / scanned / gt -> prob / scanned / tri,
//                  float4 per lane, + per-env coverage count.  HBM-bound:
//                  G^3 * 4 * 6 B per env (the dominant term of SURVEY 8d's B_vox).
//
// Set semantics: a voxel crossed by any number of rays is
// decremented once per step, hit voxels are then set to 1.0 -- exactly what the
// reference's non-accumulating index_put does -- hence bitmasks, not counters.
//
// Workgroup -> XCD placement: block b runs on XCD b % 8 (observed, used for speed
// only).  k_hit_mask and k_raycast map env e to blocks with b % 8 == e % 8, so one
// env's mask words stay in one XCD's L2 between the launches.
// Correctness never depends on it: the only inter-workgroup traffic inside a
// launch is device-scope atomicOr, the rest crosses kernel boundaries.
#include "common.h"
#include "../../include/gennbv_hip.h"
#include <stdlib.h>
#include <stdio.h>

// ---------------------------------------------------------------------------
// canonical fp32 arithmetic (DESIGN.md "canonical order"); file is built with
// -ffp-contract=off and every product/sum below is an explicit _rn intrinsic.
// ---------------------------------------------------------------------------
struct Intrinsics { float k[9]; };

__device__ __forceinline__ float nan_to_num_neginf0(float x)
{
    // torch.nan_to_num(x, neginf=0): NaN -> 0, +inf -> FLT_MAX, -inf -> 0
    if (x != x) return 0.0f;
    if (__builtin_isinf(x)) return x > 0.0f ? FLT_MAX : 0.0f;
    return x;
}

__device__ __forceinline__ float process_depth(float raw, float sense_dist)
{
    float d = nan_to_num_neginf0(raw);
    d = d < sense_dist ? sense_dist : d;  // clamp(min=-50)
    return fabsf(d);
}

__device__ __forceinline__ void pixel_to_world(float d, float u, float v, const Intrinsics &K, const float *M, float *out)
{
    const float pu = __fmul_rn(d, u), pv = __fmul_rn(d, v), pw = d;  // d * 1.0f == d
    float cam[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = __fmul_rn(K.k[i * 3 + 0], pu);
        acc = __fmaf_rn(K.k[i * 3 + 1], pv, acc);
        acc = __fmaf_rn(K.k[i * 3 + 2], pw, acc);
        cam[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = __fmul_rn(M[i * 4 + 0], cam[0]);
        acc = __fmaf_rn(M[i * 4 + 1], cam[1], acc);
        acc = __fmaf_rn(M[i * 4 + 2], cam[2], acc);
        acc = __fmaf_rn(M[i * 4 + 3], 1.0f, acc);
        out[i] = acc;
    }
}

struct VoxelFrame {  // per-env constants of scanned_pts_to_idx_3D
    float vmin[3], vmax[3], vox[3];
};

__device__ __forceinline__ VoxelFrame load_frame(const float *range6, const float *vox3)
{
    VoxelFrame f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = vox3[a];
        const float half = __fmul_rn(0.5f, v);
        f.vox[a] = v;
        f.vmax[a] = __fadd_rn(range6[2 * a], half);
        f.vmin[a] = __fsub_rn(range6[2 * a + 1], half);
    }
    return f;
}

// returns the linear voxel index (x*G + y)*G + z, or -1 when the point is dropped
__device__ __forceinline__ int point_to_voxel(const float *p, const VoxelFrame &f, int g, int *ix)
{
    bool keep = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float q = __fdiv_rn(__fsub_rn(p[a], f.vmin[a]), f.vox[a]);
        const float fl = floorf(q);
        keep = keep && (f.vmax[a] > p[a]) && (p[a] > f.vmin[a]);
        int i = (fl == fl && fabsf(fl) < 1.0e9f) ? (int)fl : 0;
        i = i < 0 ? 0 : (i > g - 1 ? g - 1 : i);
        ix[a] = i;
    }
    return keep ? (ix[0] * g + ix[1]) * g + ix[2] : -1;
}

__device__ __forceinline__ int pose_axis_to_idx(float p, float range_min, float v)
{
    const float vmin = __fsub_rn(range_min, __fmul_rn(0.5f, v));
    const float fl = floorf(__fdiv_rn(__fsub_rn(p, vmin), v));
    // float -> int64 -> int32 (pts_source.int(), utils.py:32); poses are finite
    return (int)(long long)fl;
}

// ---------------------------------------------------------------------------
// integer 3-D Bresenham (gennbv/utils.py:48-167).  Visit(x, y, z) is invoked for
// every emitted (in-bounds) voxel; returns the emitted count.
// ---------------------------------------------------------------------------
template <typename Visit>
__device__ __forceinline__ int bresenham_walk(int x0, int y0, int z0, int x1, int y1, int z1, int g, int max_pts, Visit &&visit)
{
    const int dx = abs(x1 - x0), dy = abs(y1 - y0), dz = abs(z1 - z0);
    const int sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1, sz = z0 < z1 ? 1 : -1;
    const int dm = max(max(dx, dy), dz);
    // permute so that `a` is the dominant axis and (b, c) keep the reference's order
    int pa, pb, pc, da, db, dc, sa, sb, sc, axis;
    if (dm == dx)      { axis = 0; pa = x0; pb = y0; pc = z0; da = dx; db = dy; dc = dz; sa = sx; sb = sy; sc = sz; }
    else if (dm == dy) { axis = 1; pa = y0; pb = x0; pc = z0; da = dy; db = dx; dc = dz; sa = sy; sb = sx; sc = sz; }
    else               { axis = 2; pa = z0; pb = x0; pc = y0; da = dz; db = dx; dc = dy; sa = sz; sb = sx; sc = sy; }
    int p1 = 2 * db - da, p2 = 2 * dc - da;
    const unsigned ug = (unsigned)g;
    int emitted = 0;
    auto emit = [&]() {
        if ((unsigned)pa < ug && (unsigned)pb < ug && (unsigned)pc < ug) {
            if (axis == 0) visit(pa, pb, pc);
            else if (axis == 1) visit(pb, pa, pc);
            else visit(pb, pc, pa);
            ++emitted;
        }
    };
    emit();
    for (int i = 0; i < da && emitted < max_pts; ++i) {
        if (p1 >= 0) { pb += sb; p1 -= 2 * da; }
        if (p2 >= 0) { pc += sc; p2 -= 2 * da; }
        pa += sa;
        p1 += 2 * db;
        p2 += 2 * dc;
        emit();
    }
    return emitted;
}

// ===========================================================================
// fused path, launch 1: hit mask
// ===========================================================================
constexpr int kHitThreads = 256;

// floor(num / den) for 0 <= num < 2^23, den > 0, inv_den = 1/den rounded: one mul + fix-up.
__device__ __forceinline__ int floor_div_small(int num, int den, float inv_den)
{
    int q = (int)(__fmul_rn((float)num, inv_den));
    const int r = num - q * den;
    q += (r >= den) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// Per-env constants of the hot loop.  inv_vox is only a *predictor*: the result is
// always floor(RN(x / v)) -- see voxel_axis_fast.
struct HitFrame {
    float vmin[3], vmax[3], vox[3], inv_vox[3], gmax;
};

// floor((p - vmin) / v) with IEEE division semantics, without paying for the division:
// q' = x * RN(1/v) differs from RN(x / v) by at most ~1.6 * 2^-23 * |q|, so the two floors
// can only differ when q' lies within that distance of an integer; there (and only there)
// the exact quotient is evaluated.  Bit-exact by construction, ~2^-20 slow-path rate.
__device__ __forceinline__ float voxel_axis_fast(float p, float vmin, float v, float inv_v)
{
    const float x = __fsub_rn(p, vmin);
    const float q = __fmul_rn(x, inv_v);
    float fl = floorf(q);
    const float frac = __fsub_rn(q, fl);
    const float thr = __fmul_rn(fabsf(q), 4.76837158203125e-07f);  // 2^-21
    if (__builtin_expect(!(frac >= thr && frac <= __fsub_rn(1.0f, thr)), 0)) fl = floorf(__fdiv_rn(x, v));
    return fl;
}

// WIN: the env's bitmask does not fit the 160 KiB of LDS a workgroup may use (G > 104): `windows` workgroups per
// (env, chunk), each keeps the mask words [win * nw, win * nw + nw) in LDS and drops the bits outside (every
// workgroup still evaluates all pixels of the chunk: 2x the arithmetic at G = 128, but no device-scope atomic per pixel).
template <bool KFAST, bool WIN = false>
__global__ __launch_bounds__(kHitThreads) void k_hit_mask(
    const float *__restrict__ depth_raw, const float *__restrict__ seg_raw, const float *__restrict__ c2w,
    Intrinsics K, const float *__restrict__ range_gt, const float *__restrict__ voxel_size,
    int n, int h, int w, int g, float sense_dist, int chunks, int words, uint32_t *__restrict__ hit_mask, int windows, int nw)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mask[];
    // XCD-aware block -> (env, chunk[, window]): all workgroups of env e run on XCD e % 8
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int per_env = chunks * (WIN ? windows : 1);
    const int e = (slot / per_env) * 8 + xcd;
    const int c = (slot % per_env) % chunks;
    const int w0 = WIN ? ((slot % per_env) / chunks) * nw : 0;
    if (e >= n) return;
    if (WIN) nw = min(nw, words - w0);
    else nw = words;
    for (int i = threadIdx.x; i < nw; i += kHitThreads) s_mask[i] = 0u;

    float M[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) M[i] = c2w[(size_t)e * 16 + i];
    const VoxelFrame f = load_frame(range_gt + e * 6, voxel_size + e * 3);
    HitFrame hf;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        hf.vmin[a] = f.vmin[a]; hf.vmax[a] = f.vmax[a]; hf.vox[a] = f.vox[a];
        hf.inv_vox[a] = __frcp_rn(f.vox[a]);
    }
    hf.gmax = (float)(g - 1);
    const int hw = h * w;
    int ppc = (hw + chunks - 1) / chunks;
    ppc = (ppc + 3) & ~3;
    const int px0 = c * ppc, px1 = min(hw, px0 + ppc);
    const float *dptr = depth_raw + (size_t)e * hw;
    const float *sptr = seg_raw + (size_t)e * hw;
    const float inv_w = __frcp_rn((float)w);
    const int gg = g * g;
    __syncthreads();

    // (x, y) pixel coordinates as floats; draw/sraw the RAW camera values
    auto one_pixel = [&](float fx, float fy, float draw, float sraw) {
        // seg: nan_to_num(neginf=0) then > 50 (env_train_base.py:530-531, env_train_gennbv.py:504)
        // == plain `> 50` (NaN and -inf compare false, +inf -> FLT_MAX stays true)
        if (!(sraw > 50.0f)) return;
        const float d = process_depth(draw, sense_dist);
        float wp[3];
        if (KFAST && d <= 50.0f) {
            // inv_intri = [[a,0,c],[0,b,d],[0,0,1]] (checked on the host) and finite products:
            // the zero terms of the fma chain vanish exactly, cam_z = d.
            const float pu = __fmul_rn(d, fx), pv = __fmul_rn(d, fy);
            const float cx = __fmaf_rn(K.k[2], d, __fmul_rn(K.k[0], pu));
            const float cy = __fmaf_rn(K.k[5], d, __fmul_rn(K.k[4], pv));
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float acc = __fmul_rn(M[i * 4 + 0], cx);
                acc = __fmaf_rn(M[i * 4 + 1], cy, acc);
                acc = __fmaf_rn(M[i * 4 + 2], d, acc);
                wp[i] = __fadd_rn(acc, M[i * 4 + 3]);  // fma(t, 1, acc) == acc + t
            }
        } else {
            pixel_to_world(d, fx, fy, K, M, wp);
        }
        const bool keep = (hf.vmax[0] > wp[0]) && (wp[0] > hf.vmin[0]) && (hf.vmax[1] > wp[1]) && (wp[1] > hf.vmin[1]) &&
                          (hf.vmax[2] > wp[2]) && (wp[2] > hf.vmin[2]);
        if (!keep) return;
        // kept points are finite: clamp in fp32 (v_med3) then convert
        const int ix = (int)__builtin_fminf(__builtin_fmaxf(voxel_axis_fast(wp[0], hf.vmin[0], hf.vox[0], hf.inv_vox[0]), 0.0f), hf.gmax);
        const int iy = (int)__builtin_fminf(__builtin_fmaxf(voxel_axis_fast(wp[1], hf.vmin[1], hf.vox[1], hf.inv_vox[1]), 0.0f), hf.gmax);
        const int iz = (int)__builtin_fminf(__builtin_fmaxf(voxel_axis_fast(wp[2], hf.vmin[2], hf.vox[2], hf.inv_vox[2]), 0.0f), hf.gmax);
        const int lin = ix * gg + iy * g + iz;
        if (WIN) {
            const unsigned wi = (unsigned)((lin >> 5) - w0);
            if (wi < (unsigned)nw) atomicOr(&s_mask[wi], 1u << (lin & 31));
        } else {
            atomicOr(&s_mask[lin >> 5], 1u << (lin & 31));
        }
    };

    if ((w & 3) == 0 && hw < (1 << 23)) {
        for (int p = px0 + threadIdx.x * 4; p < px1; p += kHitThreads * 4) {
            const float4 d4 = *reinterpret_cast<const float4 *>(dptr + p);
            const float4 s4 = *reinterpret_cast<const float4 *>(sptr + p);
            const int y = floor_div_small(p, w, inv_w);  // 4 consecutive pixels share the row (w % 4 == 0)
            const float fy = (float)y, fx = (float)(p - y * w);
            one_pixel(fx, fy, d4.x, s4.x);
            one_pixel(fx + 1.0f, fy, d4.y, s4.y);
            one_pixel(fx + 2.0f, fy, d4.z, s4.z);
            one_pixel(fx + 3.0f, fy, d4.w, s4.w);
        }
    } else {
        for (int p = px0 + threadIdx.x; p < px1; p += kHitThreads) {
            const int y = p / w;
            one_pixel((float)(p - y * w), (float)y, dptr[p], sptr[p]);
        }
    }
    __syncthreads();
    uint32_t *gm = hit_mask + (size_t)e * words + w0;
    for (int i = threadIdx.x; i < nw; i += kHitThreads) {
        const uint32_t v = s_mask[i];
        if (v) atomicOr(&gm[i], v);
    }
}

// ===========================================================================
// fused path, launch 2: ray cast
//
// N x S workgroups of 1024 threads.  A workgroup loads its env's hit mask in
// super-chunks of 8192 words, 8 words per lane, coalesced and all in flight at
// once (registers, no re-read), numbers the set bits with a block-wide prefix
// sum, compacts the targets it owns (ray number % S == split) into an LDS queue
// and walks ONE RAY PER LANE.  The walk is the reference's integer Bresenham
// (gennbv/utils.py:48-167) with the axes permuted up front, so the loop body is
// branch-free and carries the linear voxel index incrementally.  Visited voxels
// are OR-ed into an LDS-resident path mask (ds_or_b32, no return) which is
// flushed once: plain coalesced stores when S == 1, device-scope atomicOr of the
// non-zero words otherwise.
//
// The `emitted < 3G` cap of the reference can never trigger: a straight line has
// at most G in-bound voxels because its dominant coordinate is strictly monotone.
//
// Measured alternatives (profiles/r01_notes.md): one wave per ray with lane = step
// via the closed form b_j = b0 + s_b*floor((2 d_b j + d_a)/(2 d_a)) is 10x slower
// (per-ray setup is replicated over 64 lanes); thread-contiguous word ranges made
// the mask reads uncoalesced and latency-serialised (141 us of pure overhead).
// ===========================================================================
constexpr int kRayThreads = 1024;
constexpr int kRayWaves = kRayThreads / kWave;
constexpr int kRayWordsPerLane = 8;
constexpr int kRayChunkWords = kRayThreads * kRayWordsPerLane;
constexpr int kQueueCap = 4096;  // targets per round (16 KiB of LDS)

template <bool LDS_PATH, bool WIN = false>
__device__ __forceinline__ void trace_ray_lane(const int (&src)[3], int lin_t, int g, int gg, uint32_t *s_path, uint32_t *pm, int w0 = 0,
                                               int nw = 0)
{
    const unsigned ug = (unsigned)g;
    int tgt[3];
    tgt[0] = lin_t / gg;
    const int rem = lin_t - tgt[0] * gg;
    tgt[1] = rem / g;
    tgt[2] = rem - tgt[1] * g;
    const int d0 = abs(tgt[0] - src[0]), d1 = abs(tgt[1] - src[1]), d2 = abs(tgt[2] - src[2]);
    const int dm = max(max(d0, d1), d2);
    // dominant axis tested x, y, z (utils.py:69,102,133); minors keep the reference's order
    const bool ax = dm == d0, ay = !ax && dm == d1;
    int pa = ax ? src[0] : (ay ? src[1] : src[2]);
    int pb = ax ? src[1] : src[0];
    int pc = (ax || ay) ? src[2] : src[1];
    const int ta = ax ? tgt[0] : (ay ? tgt[1] : tgt[2]), tb = ax ? tgt[1] : tgt[0], tc = (ax || ay) ? tgt[2] : tgt[1];
    const int da = dm, db = ax ? d1 : d0, dc = (ax || ay) ? d2 : d1;
    const int st_a = ax ? gg : (ay ? g : 1), st_b = ax ? g : gg, st_c = (ax || ay) ? 1 : g;
    const int sa = pa < ta ? 1 : -1, sb = pb < tb ? 1 : -1, sc = pc < tc ? 1 : -1;
    int p1 = 2 * db - da, p2 = 2 * dc - da;
    int l = pa * st_a + pb * st_b + pc * st_c;
    const int la = sa * st_a, lb = sb * st_b, lc = sc * st_c;
    for (int i = 0; i <= da; ++i) {
        if ((unsigned)pa < ug && (unsigned)pb < ug && (unsigned)pc < ug) {
            if (LDS_PATH && WIN) {
                const unsigned wi = (unsigned)((l >> 5) - w0);
                if (wi < (unsigned)nw) atomicOr(&s_path[wi], 1u << (l & 31));
            } else if (LDS_PATH) {
                // (testing the bit first -- same-address reads broadcast -- so that only the first ray through a voxel pays the
                // atomic was slower: 32.5 against 28.6 us, the read's latency sits on the walk's critical path)
#if defined(RAY_ABL) && RAY_ABL == 1
                if (i >= 12)
#elif defined(RAY_ABL) && RAY_ABL == 2
                if (l == 0x7ffffff)
#endif
                atomicOr(&s_path[l >> 5], 1u << (l & 31));
            } else {
                atomicOr(&pm[l >> 5], 1u << (l & 31));
            }
        }
        const bool ib = p1 >= 0, ic = p2 >= 0;
        pb += ib ? sb : 0; l += ib ? lb : 0; p1 -= ib ? 2 * da : 0;
        pc += ic ? sc : 0; l += ic ? lc : 0; p2 -= ic ? 2 * da : 0;
        pa += sa; l += la;
        p1 += 2 * db; p2 += 2 * dc;
    }
}

// WIN (LDS_PATH only): the path mask does not fit LDS either -- `windows` workgroups per (env, split) trace the same
// rays and each keeps the words [win * nw, ...) of the path mask (see k_hit_mask).
template <bool LDS_PATH, bool WIN = false>
__global__ __launch_bounds__(kRayThreads) void k_raycast(
    const uint32_t *__restrict__ hit_mask, const float *__restrict__ poses_xyz, int64_t pose_stride,
    const float *__restrict__ range_gt, const float *__restrict__ voxel_size, int n, int g, int words, int splits,
    uint32_t *__restrict__ path_mask, int windows, int nw)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *s_queue = smem;                      // kQueueCap
    int *s_wave = (int *)(smem + kQueueCap);       // kRayWaves + 1
    uint32_t *s_path = smem + kQueueCap + 32;      // words (LDS_PATH only)
    // block -> (env, split); all splits of env e sit on XCD e % 8 like k_hit_mask's chunks
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int per_env = splits * (WIN ? windows : 1);
    const int e = (slot / per_env) * 8 + xcd;
    const int sp = (slot % per_env) % splits;
    const int w0 = WIN ? ((slot % per_env) / splits) * nw : 0;
    if (e >= n) return;
    nw = WIN ? min(nw, words - w0) : words;
    const uint32_t *hm = hit_mask + (size_t)e * words;
    uint32_t *pm = path_mask + (size_t)e * words;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
    if (LDS_PATH) for (int i = tid; i < nw; i += kRayThreads) s_path[i] = 0u;

    // source voxel (pose_coord_to_idx_3D, no clamp)
    const float *pp = poses_xyz + (size_t)e * pose_stride;
    const int src[3] = {pose_axis_to_idx(pp[0], range_gt[e * 6 + 1], voxel_size[e * 3 + 0]),
                        pose_axis_to_idx(pp[1], range_gt[e * 6 + 3], voxel_size[e * 3 + 1]),
                        pose_axis_to_idx(pp[2], range_gt[e * 6 + 5], voxel_size[e * 3 + 2])};
    const int gg = g * g;

    for (int chunk0 = 0; chunk0 < words; chunk0 += kRayChunkWords) {
        uint32_t wreg[kRayWordsPerLane];
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < kRayWordsPerLane; ++k) {
            const int wi = chunk0 + k * kRayThreads + tid;
            wreg[k] = wi < words ? hm[wi] : 0u;
        }
#pragma unroll
        for (int k = 0; k < kRayWordsPerLane; ++k) cnt += __popc(wreg[k]);
        const int incl = wave_inclusive_scan(cnt);
        __syncthreads();  // s_wave / s_queue free (previous chunk done), s_path zeroed
        if (lane == kWave - 1) s_wave[wv] = incl;
        __syncthreads();
        if (wv == 0) {
            int v = lane < kRayWaves ? s_wave[lane] : 0;
            const int sc = wave_inclusive_scan(v);
            if (lane < kRayWaves) s_wave[lane] = sc - v;
            if (lane == kRayWaves - 1) s_wave[kRayWaves] = sc;
        }
        __syncthreads();
        const int my_off = s_wave[wv] + incl - cnt;
        const int total = s_wave[kRayWaves];
        // this split traces rays o with o % splits == sp; local index (o - sp) / splits
        const int my_total = total > sp ? (total - sp + splits - 1) / splits : 0;
        for (int base = 0; base < my_total; base += kQueueCap) {
            const int o_lo = base * splits + sp, o_hi = (base + kQueueCap) * splits + sp;
            if (cnt && my_off < o_hi && my_off + cnt > o_lo) {
                // ray o is mine iff rel = o - sp >= 0 and rel % splits == 0; its queue slot is
                // rel / splits - base.  (ph, slot) are advanced incrementally: no division per bit.
                const int rel0 = my_off - sp;
                int slot_q = rel0 >= 0 ? rel0 / splits : 0;
                int ph = rel0 >= 0 ? rel0 - slot_q * splits : rel0;  // < 0: counts up to the first owned ray
#pragma unroll
                for (int k = 0; k < kRayWordsPerLane; ++k) {
                    uint32_t v = wreg[k];
                    const int wi = chunk0 + k * kRayThreads + tid;
                    while (v) {
                        const int bit = __ffs(v) - 1;
                        v &= v - 1;
                        if (ph == 0 && slot_q >= base && slot_q < base + kQueueCap) s_queue[slot_q - base] = (uint32_t)(wi * 32 + bit);
                        ++ph;
                        if (ph == splits) { ph = 0; ++slot_q; }
                    }
                }
            }
            __syncthreads();
            const int nq = min(kQueueCap, my_total - base);
            for (int q = tid; q < nq; q += kRayThreads)
                trace_ray_lane<LDS_PATH, WIN>(src, (int)s_queue[q], g, gg, s_path, pm, w0, nw);
            __syncthreads();
        }
    }
    if (LDS_PATH) {
        __syncthreads();
        if (splits == 1) {
            for (int i = tid; i < nw; i += kRayThreads) pm[w0 + i] = s_path[i];
        } else {
            for (int i = tid; i < nw; i += kRayThreads) {
                const uint32_t v = s_path[i];
                if (v) atomicOr(&pm[w0 + i], v);
            }
        }
    }
}

// ===========================================================================
// fused path, launches 1 + 2 (grids whose bitmask fits a workgroup's LDS next to 31 other waves: G <= 104)
//
// k_hit_list: N x chunks workgroups of 1024 threads.  A workgroup streams its share of the env's depth + seg pixels
// (two tiles of 4096 pixels in flight: the next tile's 16-byte requests are issued before the current tile is
// back-projected), ORs the voxel bits into an LDS hit mask -- a pixel whose left neighbour maps to the same voxel leaves
// the bit to it (same-address LDS atomics were 16 of the kernel's 64 us) -- then numbers the set bits with a block-wide
// prefix sum and appends them as a RAY LIST (one int32 target voxel per ray) to the env's slice of the workspace; the
// mask goes out with plain stores (one workgroup per env) or atomicOr of the non-zero words.
//
// k_ray_list: work items = slices of 384 rays of one env's list, one ray per lane (the reference's integer Bresenham as packed
// 16-bit arithmetic, walk_packed), into an LDS path mask whose non-zero words are ORed out.  The per-env ray counts differ 4x
// around their mean (an env that faces a large surface has thousands): with one workgroup -- or a fixed split -- per env the
// launch lasts as long as its busiest env (28 of k_raycast's 65 us were the walk, 59 us in a one-launch hit + walk kernel);
// slices spread an env over as many CUs as it needs.  The grid is compact: a block finds its (env, slice) from the counts
// (see the kernel).  What bounds the walk: the CU's LDS atomic rate and its VALU issue, about equally (profiles/r05_notes.md).
//
// A voxel hit from pixels of two chunks is listed by both workgroups: harmless, the path is a set.
// ===========================================================================
#ifndef FUSED_THREADS
#define FUSED_THREADS 1024
#endif
constexpr int kFusedThreads = FUSED_THREADS;
constexpr int kListThreads = 384;    // k_ray_list: rays per work item = lanes per workgroup (four workgroups of six waves per CU: their
                                     // 32 KiB masks are what limits the residency; 256 / 320 / 512 measured within 8 %, profiles/r05_notes.md)
constexpr int kListGridSlices = 5;   // k_ray_list: grid = this many workgroups per env on average
constexpr int kSlabThreads = 256;    // k_ray_slab: rays per slice
constexpr int kListSlices = 16;      // k_ray_slab: slices per env and slab at most

typedef float v4f_t __attribute__((ext_vector_type(4)));
// streaming 16-byte load (read-once camera data: keep it out of the L2's way)
__device__ __forceinline__ float4 ld4_stream(const float *p)
{
    const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// The same request as an opaque instruction + an explicit wait: the compiler's wait-count insertion gives up on requests that are
// re-issued behind a branch region inside a loop (it waits for EVERY outstanding request, `s_waitcnt vmcnt(0)`, in front of each
// tile).  The wait names the registers it releases ("+v"), so no use can be scheduled above it.
__device__ __forceinline__ void ld4_stream_async(v4f_t &dst, const float *p)
{
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}
// (the same with the base in scalar registers and a 32-bit byte offset per lane: no 64-bit address arithmetic -- a quarter-rate
// v_mad_u64_u32 per request -- in the hot loop; `base` must be wave-uniform)
__device__ __forceinline__ void ld4_stream_async_sbase(v4f_t &dst, const float *base, unsigned byte_off)
{
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(byte_off), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_keep(v4f_t &a, v4f_t &b)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

struct PixelFrame {
    float M[12];
    HitFrame hf;
    float lo[3], hi[3];  // next_up(vmin), next_down(vmax): the closed interval of the kept points
    int g, gg;
    float sense_dist;
};

// Linear voxel index of one raw pixel, or -1 (background / outside the grid); canonical fp32 order (see k_hit_mask).
// Two RARE branches (a depth beyond the sensing range; a quotient within round-off of an integer) instead of an early exit
// per test.  Measured alternatives (profiles/r02_notes.md): the early-exit form of k_hit_mask (a third of the instructions
// were exec-mask bookkeeping) and a fully straight-line form over the four pixels of a lane with one merged slow-path
// branch (48 us against 46: it computes everything for background pixels too).
template <bool KFAST>
__device__ __forceinline__ int pixel_to_lin(const PixelFrame &pf, const Intrinsics &K, float fx, float fy, float draw, float sraw)
{
    const bool fg = sraw > 50.0f;  // nan_to_num(neginf=0) then > 50 == plain `> 50`
    const float d = process_depth(draw, pf.sense_dist);
    float wp[3];
    if (KFAST) {
        // inv_intri = [[a,0,c],[0,b,d],[0,0,1]] (checked on the host) and finite products: the zero terms of the fma chain
        // vanish exactly, cam_z = d.
        const float pu = __fmul_rn(d, fx), pv = __fmul_rn(d, fy);
        const float cx = __fmaf_rn(K.k[2], d, __fmul_rn(K.k[0], pu));
        const float cy = __fmaf_rn(K.k[5], d, __fmul_rn(K.k[4], pv));
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float acc = __fmul_rn(pf.M[i * 4 + 0], cx);
            acc = __fmaf_rn(pf.M[i * 4 + 1], cy, acc);
            acc = __fmaf_rn(pf.M[i * 4 + 2], d, acc);
            wp[i] = __fadd_rn(acc, pf.M[i * 4 + 3]);  // fma(t, 1, acc) == acc + t
        }
    }
    if (!KFAST || __builtin_expect(fg && !(d <= 50.0f), 0)) pixel_to_world(d, fx, fy, K, pf.M, wp);  // (+inf -> FLT_MAX etc.)
    const HitFrame &hf = pf.hf;
    // vmax > p > vmin on the three axes  <=>  p == clamp(p, next_up(vmin), next_down(vmax))  (a NaN compares unequal, like
    // the reference's comparisons): three v_med3 + three compares instead of six compares and their mask arithmetic
    const bool keep = fg && (__builtin_amdgcn_fmed3f(wp[0], pf.lo[0], pf.hi[0]) == wp[0]) &&
                      (__builtin_amdgcn_fmed3f(wp[1], pf.lo[1], pf.hi[1]) == wp[1]) &&
                      (__builtin_amdgcn_fmed3f(wp[2], pf.lo[2], pf.hi[2]) == wp[2]);
    // floor((p - vmin) / v) with IEEE division semantics via the reciprocal predictor (see voxel_axis_fast); the three
    // "quotient within 2^-21 |q| of an integer" tests fold into one comparison of the smallest margin
    float fl[3], margin[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = __fsub_rn(wp[a], hf.vmin[a]);
        const float q = __fmul_rn(x, hf.inv_vox[a]);
        fl[a] = floorf(q);
        const float frac = __fsub_rn(q, fl[a]);
        const float thr = __fmul_rn(fabsf(q), 4.76837158203125e-07f);  // 2^-21
        // frac >= thr && frac <= 1 - thr   <=>   min(frac - thr, (1 - thr) - frac) >= 0   (each difference has the sign of its comparison)
        margin[a] = __builtin_fminf(__fsub_rn(frac, thr), __fsub_rn(__fsub_rn(1.0f, thr), frac));
    }
    const bool exact_needed = !(__builtin_fminf(__builtin_fminf(margin[0], margin[1]), margin[2]) >= 0.0f);
    if (__builtin_expect(keep && exact_needed, 0)) {
#pragma unroll
        for (int a = 0; a < 3; ++a) fl[a] = floorf(__fdiv_rn(__fsub_rn(wp[a], hf.vmin[a]), hf.vox[a]));
    }
    // kept points are finite: clamp in fp32 (v_med3) then convert
    const int ix = (int)__builtin_fminf(__builtin_fmaxf(fl[0], 0.0f), hf.gmax);
    const int iy = (int)__builtin_fminf(__builtin_fmaxf(fl[1], 0.0f), hf.gmax);
    const int iz = (int)__builtin_fminf(__builtin_fmaxf(fl[2], 0.0f), hf.gmax);
    return keep ? ix * pf.gg + iy * pf.g + iz : -1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Voxel-space PREDICTOR of a pixel's voxel (round 3).  For a pinhole camera the canonical chain of pixel_to_lin is, in real
// arithmetic, affine in the depth:   q_a = (world_a - vmin_a) / vox_a = d * (al_a u + be_a v + ga_a) + ta_a   (u, v = pixel column /
// row), with constants per env and axis.  Evaluated as two fma per axis it is ~6 instructions instead of ~30 for the three
// quotients -- but it rounds differently from the canonical chain, so it only DECIDES where it provably cannot disagree:
//
//   |q_pred - q_canonical| <= 2^-24 (13 d S_a + 5 |M_a3| + 4 |vmin_a|) / vox_a        (both chains: one rounding per operation,
//                                                                                      magnitudes bounded term by term; S_a =
//                                                                                      |M_a0| Cx + |M_a1| Cy + |M_a2|, Cx = |K0| (w-1)
//                                                                                      + |K2|, Cy = |K4| (h-1) + |K5|)
//   thr(d) = kap d + lam  with  kap = 1.02 * 2^-24 * 16 max_a S_a / vox_a,  lam = 1.02 * 2^-24 * 8 max_a (|M_a3| + |vmin_a|) / vox_a,
//            lam >= 8 |G - (vmax_a - vmin_a) / vox_a|   (so that "floor in [0, G)" and the reference's `vmax > p > vmin` agree)
//
// A pixel whose three predicted quotients all keep a distance >= thr from every integer gets floor(q_pred) = floor(q_canonical)
// on every axis, and its in / out-of-grid decision is the reference's: it is decided here.  Everything else -- a quotient near a
// voxel boundary (~0.1 % of the foreground pixels at 64^3), a depth the clamp / nan_to_num of post_process_camera_tensor changes, a
// non-finite anything (NaN fails the `>=`) -- is QUEUED and evaluated with the canonical chain (pixel_to_lin) after the stream.
// The result is bit-identical to the canonical chain by construction; tests/test_voxel_gpu.py compares both paths with the oracle.
// ---------------------------------------------------------------------------------------------------------------------------
struct VoxPredict {
    float al[3], be[3], ga[3], ta[3];
    float kap, lim0;  // a pixel is decided when every quotient keeps |fract - 0.5| <= lim0 - kap d  (lim0 = 0.5 - lam)
};

__device__ __forceinline__ float uniform_f32(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// (every lane computes the same ~100 fp64 operations once per workgroup; the results live in scalar registers)
__device__ __forceinline__ VoxPredict make_predictor(const float *__restrict__ M16, const Intrinsics &K, const float *__restrict__ range6,
                                                     const float *__restrict__ vox3, int g, int h, int w)
{
    VoxPredict vp;
    const VoxelFrame f = load_frame(range6, vox3);
    const double k0 = K.k[0], k2 = K.k[2], k4 = K.k[4], k5 = K.k[5];
    const double cx = fabs(k0) * (double)(w - 1) + fabs(k2), cy = fabs(k4) * (double)(h - 1) + fabs(k5);
    double kap = 0.0, lam = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double m0 = M16[a * 4 + 0], m1 = M16[a * 4 + 1], m2 = M16[a * 4 + 2], m3 = M16[a * 4 + 3];
        const double inv = 1.0 / (double)f.vox[a], vmin = f.vmin[a];
        vp.al[a] = uniform_f32((float)(m0 * k0 * inv));
        vp.be[a] = uniform_f32((float)(m1 * k4 * inv));
        vp.ga[a] = uniform_f32((float)((m0 * k2 + m1 * k5 + m2) * inv));
        vp.ta[a] = uniform_f32((float)((m3 - vmin) * inv));
        const double sa = fabs(m0) * cx + fabs(m1) * cy + fabs(m2);
        kap = fmax(kap, 16.0 * sa * fabs(inv));
        lam = fmax(lam, 8.0 * (fabs(m3) + fabs(vmin)) * fabs(inv));
        const double qhi = ((double)f.vmax[a] - vmin) * inv;
        lam = fmax(lam, 8.0 * fabs((double)g - qhi) * 16777216.0 / 1.02);  // (scaled back below)
        if (!(inv > 0.0)) lam = __builtin_inf();  // a non-positive / NaN voxel size: every pixel takes the canonical chain
    }
    vp.kap = uniform_f32((float)(kap * (1.02 / 16777216.0)));
    vp.lim0 = uniform_f32((float)(0.5 - fmax(lam * (1.02 / 16777216.0), 1.0e-6)));
    return vp;
}

// One pixel through the predictor: lin >= 0 decided "in the grid, this voxel"; -1 decided "not a hit"; `queue` set: undecided.
__device__ __forceinline__ int pixel_predict(const VoxPredict &vp, const float (&row)[3], float u, float draw, float sraw, float sense_dist,
                                             unsigned ug, bool &queue)
{
    const bool fg = sraw > 50.0f;
    const float d = fabsf(draw);
    const bool plain = draw >= sense_dist;  // nan_to_num and the clamp leave it alone: d = |raw| (NaN / -inf / below the range fail)
    // per axis: two fma, v_fract (q - floor q, exact), its distance from 0.5, v_cvt_flr_i32_f32 (floor + convert), one unsigned compare
    float dev[3];
    int ii[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float r = __fmaf_rn(vp.al[a], u, row[a]);
        const float q = __fmaf_rn(d, r, vp.ta[a]);
        dev[a] = __fsub_rn(__builtin_amdgcn_fractf(q), 0.5f);
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ii[a]) : "v"(q));
    }
    // in the grid on the three axes: one unsigned v_max3 + one compare (a negative index is a huge unsigned one)
    unsigned imax;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(imax) : "v"(ii[0]), "v"(ii[1]), "v"(ii[2]));
    const bool in = imax < ug;
    // distance to the nearest integer = 0.5 - |fract - 0.5| >= thr   <=>   max |dev| <= 0.5 - thr   (false for NaN; the two roundings
    // of this test are 2^-25 each against a slack of 0.2 thr: make_predictor keeps thr >= 1e-6)
    const float lim = __fmaf_rn(-vp.kap, d, vp.lim0);
    const bool sure = __builtin_fmaxf(__builtin_fmaxf(fabsf(dev[0]), fabsf(dev[1])), fabsf(dev[2])) <= lim;
    queue = fg && !(plain && sure);
    // (two v_mad_u32_u24: the compiler turned the outer one into the quarter-rate v_mad_u64_u32; the value is only used when `in`)
    int lin;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(lin) : "v"(ii[0]), "s"(ug), "v"(ii[1]));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(lin) : "v"(lin), "s"(ug), "v"(ii[2]));
    return (fg && plain && sure && in) ? lin : -1;
}

__device__ __forceinline__ void load_pixel_frame(PixelFrame &pf, const float *__restrict__ c2w, const float *__restrict__ range_gt,
                                                 const float *__restrict__ voxel_size, int e, int g, float sense_dist)
{
#pragma unroll
    for (int i = 0; i < 12; ++i) pf.M[i] = c2w[(size_t)e * 16 + i];
    const VoxelFrame f = load_frame(range_gt + e * 6, voxel_size + e * 3);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        pf.hf.vmin[a] = f.vmin[a]; pf.hf.vmax[a] = f.vmax[a]; pf.hf.vox[a] = f.vox[a];
        pf.hf.inv_vox[a] = __frcp_rn(f.vox[a]);
        pf.lo[a] = nextafterf(f.vmin[a], INFINITY);
        pf.hi[a] = nextafterf(f.vmax[a], -INFINITY);
    }
    pf.hf.gmax = (float)(g - 1);
    pf.g = g; pf.gg = g * g; pf.sense_dist = sense_dist;
}

// undecided pixels of the predictor per workgroup (LDS queue); GNBV_NO_PREDICTOR builds (A/B, tests) run the canonical chain only
constexpr int kQueueCapPx = 2048;
#ifdef GNBV_NO_PREDICTOR
constexpr bool kUsePredictor = false;
#else
constexpr bool kUsePredictor = true;
#endif

// (amdgpu_num_sgpr: see launch_masks -- SGPR-limited occupancy)
template <bool KFAST>
__global__ __launch_bounds__(kFusedThreads) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void k_hit_list(
    const float *__restrict__ depth_raw, const float *__restrict__ seg_raw, const float *__restrict__ c2w, Intrinsics K,
    const float *__restrict__ range_gt, const float *__restrict__ voxel_size, int n, int h, int w, int g, float sense_dist,
    int chunks, int words, uint32_t *__restrict__ hit_mask, int32_t *__restrict__ ray_count, int32_t *__restrict__ ray_list,
    int64_t ray_cap, int32_t *__restrict__ coverage_zero)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *s_hit = smem;                 // words
    int *s_wave = (int *)(smem + words);    // counters (64 ints reserved)
    uint16_t *s_widx = (uint16_t *)(s_wave + 64);  // words: indices of the non-zero mask words
    int *s_qcnt = s_wave + 8;                      // undecided pixels of the predictor (count, then the queue behind the word list)
    int *s_queue = (int *)(s_widx + ((words + 1) & ~1));
    // XCD-aware block -> (env, chunk): all workgroups of env e run on XCD e % 8
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int e = (slot / chunks) * 8 + xcd;
    // (Rotating which of an env's chunks a slot takes every 16 envs, so that the two workgroups of a CU hold different chunks whichever
    // way the dispatcher deals them -- the top of an image is mostly background --: -0.5 us of 89, inside the noise; profiles/r05_notes.md)
    const int c = slot % chunks;
    if (e >= n) return;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    // (Round 5 wrote this workgroup's share of the env's all-zero PATH mask as zeros here, to have its lines in this XCD's L2 when the ray
    // walk ORs into them -- inside the rollout the masks come from HBM: +4.3 us back to back, +1.3 ... +4.5 us inside the rollout; the
    // walk's 12 extra microseconds there are not its atomics' cache misses.  profiles/r05_notes.md section 1g)
#ifdef PHASE_TIMING
    const uint64_t pt0 = wall_clock64();
#endif
    // (Round 5 wrote this workgroup's share of the env's all-zero PATH mask as zeros here, to have its lines in this XCD's L2 when the ray
    // walk ORs into them -- inside the rollout the masks come from HBM: +4.3 us back to back, +1.3 ... +4.5 us inside the rollout: the
    // walk's time there is not its atomics' cache misses.  profiles/r05_notes.md section 1g)
    for (int i = tid; i < words; i += kFusedThreads) smem[i] = 0u;
    if (tid == 0) *s_qcnt = 0;
    if (coverage_zero != nullptr && c == 0 && tid == 0) coverage_zero[e] = 0;  // (accumulated by the grid-update launch)

    const int hw = h * w;
    int ppc = (hw + chunks - 1) / chunks;
    ppc = (ppc + 3) & ~3;
    const int px0 = c * ppc, px1 = min(hw, px0 + ppc);
    const float *dptr = depth_raw + (size_t)e * hw;
    const float *sptr = seg_raw + (size_t)e * hw;
    const float inv_w = __frcp_rn((float)w);
    __syncthreads();

    // ---- phase A: hit mask ------------------------------------------------------------------------------
    bool need_full_exact = false;  // (uniform) the canonical chain over every pixel of the chunk
    if (KFAST && (w & 3) == 0 && hw < (1 << 23) && kUsePredictor) {
        // kAhead tiles of 4096 pixels in flight per workgroup (2 x 16-byte streaming requests per lane and tile); addresses are
        // clamped, so every request is unconditional and the waits land behind a tile's arithmetic.  Round 2 ran the canonical chain
        // on every pixel: 36-44 us of instruction issue for a 25 us stream.  Round 3: the voxel-space predictor decides ~99.9 % of
        // the foreground pixels in ~30 instructions, branch-free; the rest goes to an LDS queue (see VoxPredict).
        const VoxPredict vp = make_predictor(c2w + (size_t)e * 16, K, range_gt + e * 6, voxel_size + e * 3, g, h, w);
        const unsigned ug = (unsigned)g;
        constexpr int kTile = kFusedThreads * 4, kAhead = 3;
        v4f_t dq[kAhead], sq[kAhead];
        // (Dealing an env's tiles to its workgroups round-robin instead of as contiguous chunks -- the top of an image is mostly
        // background -- was measured: both workgroups then list most of the env's voxels, k_ray_list 28 -> 33 us, phase A unchanged.)
        const int plast = px1 - 4;
        const int ntiles = (px1 - px0 + kTile - 1) / kTile;
        auto tile_px = [&](int t) { return px0 + t * kTile + tid * 4; };
#pragma unroll
        for (int q = 0; q < kAhead; ++q) {
            const int pq = max(min(tile_px(max(min(q, ntiles - 1), 0)), plast), 0);
            ld4_stream_async_sbase(dq[q], dptr, (unsigned)pq * 4u);
            ld4_stream_async_sbase(sq[q], sptr, (unsigned)pq * 4u);
        }
        // (column, row) of the lane's first pixel, advanced by one tile per step without a division
        const int tdy = kTile / w, tdx = kTile - tdy * w;
        int py = floor_div_small(max(min(tile_px(0), plast), 0), w, inv_w), pxc = max(min(tile_px(0), plast), 0) - py * w;
        for (int t0 = 0; t0 < ntiles; t0 += kAhead) {
#pragma unroll
            for (int q = 0; q < kAhead; ++q) {
                const int t = t0 + q;
                const int p = t < ntiles ? tile_px(t) : px1;
                wait_vm_keep<2 * (kAhead - 1)>(dq[q], sq[q]);  // requests complete in order: the two other slots' may stay in flight
                const v4f_t d4 = dq[q], s4 = sq[q];
                // (a tile of pure background -- all 256 pixels of the wave -- skips the arithmetic)
                if (p < px1 && __any((s4.x > 50.0f) | (s4.y > 50.0f) | (s4.z > 50.0f) | (s4.w > 50.0f))) {
                    const float fy = (float)py, fx = (float)pxc;  // 4 consecutive pixels share the row (w % 4 == 0)
                    const float row[3] = {__fmaf_rn(vp.be[0], fy, vp.ga[0]), __fmaf_rn(vp.be[1], fy, vp.ga[1]), __fmaf_rn(vp.be[2], fy, vp.ga[2])};
                    bool u0, u1, u2, u3;
#ifdef GNBV_ABL_NOARITH
                    u0 = u1 = u2 = u3 = false;
                    const int l0 = d4.x == 12345.0f ? 1 : -1, l1 = d4.y == 12345.0f ? 1 : -1, l2 = d4.z == 12345.0f ? 1 : -1, l3 = d4.w == 12345.0f ? 1 : -1;
#else
                    const int l0 = pixel_predict(vp, row, fx, d4.x, s4.x, sense_dist, ug, u0);
                    const int l1 = pixel_predict(vp, row, fx + 1.0f, d4.y, s4.y, sense_dist, ug, u1);
                    const int l2 = pixel_predict(vp, row, fx + 2.0f, d4.z, s4.z, sense_dist, ug, u2);
                    const int l3 = pixel_predict(vp, row, fx + 3.0f, d4.w, s4.w, sense_dist, ug, u3);
#endif
                    // neighbouring pixels mostly fall into the same voxel: a pixel whose left neighbour (previous pixel of the
                    // lane, or the last pixel of the previous lane in the 16-lane DPP row) has the same voxel leaves the bit to it
                    const int pl = __builtin_amdgcn_update_dpp(-2, l3, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
#ifdef GNBV_ABL_NOATOM
                    if (((l0 ^ l1 ^ l2 ^ l3 ^ pl) & 0x7fffffff) == 0x12345678) atomicOr(&s_hit[0], 1u);
#else
                    if (l0 >= 0 && l0 != pl) atomicOr(&s_hit[l0 >> 5], 1u << (l0 & 31));
                    if (l1 >= 0 && l1 != l0) atomicOr(&s_hit[l1 >> 5], 1u << (l1 & 31));
                    if (l2 >= 0 && l2 != l1) atomicOr(&s_hit[l2 >> 5], 1u << (l2 & 31));
                    if (l3 >= 0 && l3 != l2) atomicOr(&s_hit[l3 >> 5], 1u << (l3 & 31));
#endif
                    if (__any(u0 | u1 | u2 | u3)) {  // (~20 % of the wave-tiles hold an undecided pixel, ~0.1 % of the pixels are)
                        const int cnt = (int)u0 + (int)u1 + (int)u2 + (int)u3;
                        if (cnt) {
                            int slot = atomicAdd(s_qcnt, cnt);
                            if (u0) { if (slot < kQueueCapPx) s_queue[slot] = p; ++slot; }
                            if (u1) { if (slot < kQueueCapPx) s_queue[slot] = p + 1; ++slot; }
                            if (u2) { if (slot < kQueueCapPx) s_queue[slot] = p + 2; ++slot; }
                            if (u3) { if (slot < kQueueCapPx) s_queue[slot] = p + 3; }
                        }
                    }
                }
                pxc += tdx; py += tdy;
                if (pxc >= w) { pxc -= w; ++py; }
                // refill THIS slot behind its last use (the slot's registers are reused in place: issuing the request in front of
                // the arithmetic made the compiler keep a fourth register set and rotate all of them -- behind a wait for every
                // outstanding request -- at the end of each round); kAhead - 1 tiles per lane stay in flight during the arithmetic,
                // 128 KiB per CU against the ~25 KiB its 10.8 B/clk x latency needs
                const int pn = max(min(tile_px(max(min(t + kAhead, ntiles - 1), 0)), plast), 0);
                ld4_stream_async_sbase(dq[q], dptr, (unsigned)pn * 4u);
                ld4_stream_async_sbase(sq[q], sptr, (unsigned)pn * 4u);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the clamped duplicate requests of the last round)
        __syncthreads();
        // the undecided pixels through the canonical chain (all lanes busy; the pixels come back from L2 / HBM: ~50 per workgroup)
        const int nq = *s_qcnt;
        // a queue that overflowed (a surface lying ON a voxel boundary plane, say): the canonical chain over the whole chunk
        if (nq > kQueueCapPx) {
            PixelFrame pf;
            load_pixel_frame(pf, c2w, range_gt, voxel_size, e, g, sense_dist);
            for (int p = px0 + tid; p < px1; p += kFusedThreads) {
                const int y = p / w;
                const int l0 = pixel_to_lin<KFAST>(pf, K, (float)(p - y * w), (float)y, dptr[p], sptr[p]);
                if (l0 >= 0) atomicOr(&s_hit[l0 >> 5], 1u << (l0 & 31));
            }
        } else if (nq > 0) {
            PixelFrame pf;
            load_pixel_frame(pf, c2w, range_gt, voxel_size, e, g, sense_dist);
            for (int j = tid; j < nq; j += kFusedThreads) {
                const int p = s_queue[j];
                const int y = p / w;
                const int l0 = pixel_to_lin<KFAST>(pf, K, (float)(p - y * w), (float)y, dptr[p], sptr[p]);
                if (l0 >= 0) atomicOr(&s_hit[l0 >> 5], 1u << (l0 & 31));
            }
        }
    } else {
        need_full_exact = true;
    }
    if (need_full_exact) {
        PixelFrame pf;
        load_pixel_frame(pf, c2w, range_gt, voxel_size, e, g, sense_dist);
        for (int p = px0 + tid; p < px1; p += kFusedThreads) {
            const int y = p / w;
            const int l0 = pixel_to_lin<KFAST>(pf, K, (float)(p - y * w), (float)y, dptr[p], sptr[p]);
            if (l0 >= 0) atomicOr(&s_hit[l0 >> 5], 1u << (l0 & 31));
        }
    }

    __syncthreads();
#ifdef PHASE_TIMING
    const uint64_t pt1 = wall_clock64();
#endif

    // ---- phase B: write the mask out, append its set bits to the env's ray list ------------------------------------
    // Two levels, so that no wave runs a long serial loop alone (a lone wave issues a dependent instruction every ~8
    // cycles; a lane expanding 8 strided words bit by bit took 15 us on an env with 4000 rays): (1) the non-zero words are
    // compacted into an LDS index list -- wave-aggregated slot allocation, no per-word loop; (2) one lane per non-zero word
    // (evenly spread over the workgroup) reserves list slots through a wave scan + one LDS atomic and writes its <= 32 bits.
    const int wpl = (words + kFusedThreads - 1) / kFusedThreads;  // words per lane, strided by the workgroup
    uint32_t *gh = hit_mask + (size_t)e * words;
    int *s_cnt = s_wave;  // [0] non-zero words, [1] rays, [2] this workgroup's base in the env's list
    if (tid < 4) s_cnt[tid] = 0;
    __syncthreads();
    int cnt = 0;
    for (int k = 0; k < wpl; ++k) {
        const int wi = k * kFusedThreads + tid;
        const uint32_t v = wi < words ? s_hit[wi] : 0u;
        cnt += __popc(v);
        if (wi < words) {
            if (chunks == 1) gh[wi] = v;
            else if (v) atomicOr(&gh[wi], v);
        }
        const unsigned long long nz = __ballot(v != 0u);
        if (nz) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt[0], __popcll(nz));
            base = __builtin_amdgcn_readfirstlane(base);
            if (v) s_widx[base + __popcll(nz & ((1ull << lane) - 1ull))] = (uint16_t)wi;
        }
    }
    cnt = wave_reduce_sum(cnt);
    if (lane == 0 && cnt) atomicAdd(&s_cnt[1], cnt);
#ifdef PHASE_TIMING
    const uint64_t pt2 = wall_clock64();
#endif
    __syncthreads();
    const int nwords = s_cnt[0], total = s_cnt[1];
    if (tid == 0) {
        // this workgroup's slots in the env's list (any order: the path is a set)
        s_cnt[2] = chunks == 1 ? 0 : atomicAdd(&ray_count[e], total);
        if (chunks == 1) ray_count[e] = total;
        s_cnt[3] = 0;
    }
    __syncthreads();
#ifdef PHASE_TIMING
    const uint64_t pt3 = wall_clock64();
#endif
    {
        int32_t *list = ray_list + (size_t)e * ray_cap + s_cnt[2];
        const int64_t room = ray_cap - s_cnt[2];  // (cannot overflow: one entry per pixel at most)
        for (int j0 = 0; j0 < nwords; j0 += kFusedThreads) {
            const int j = j0 + tid;
            const int wi = j < nwords ? (int)s_widx[j] : 0;
            uint32_t v = j < nwords ? s_hit[wi] : 0u;
            const int c1 = __popc(v);
            const int incl = wave_inclusive_scan(c1);
            int base = 0;
            if (lane == kWave - 1 && incl) base = atomicAdd(&s_cnt[3], incl);
            base = __builtin_amdgcn_readlane(base, kWave - 1);
            int o = base + incl - c1;
            while (v) {
                const int bit = __ffs(v) - 1;
                v &= v - 1;
                if (o < room) list[o] = wi * 32 + bit;
                ++o;
            }
        }
    }
#ifdef PHASE_TIMING
    __syncthreads();
    if (tid == 0) {
        int32_t *dbg = ray_list + (size_t)n * ray_cap - 8 * (size_t)gridDim.x;
        dbg[8 * blockIdx.x + 0] = (int32_t)(pt1 - pt0);
        dbg[8 * blockIdx.x + 1] = (int32_t)(wall_clock64() - pt1);
        dbg[8 * blockIdx.x + 2] = s_wave[1];
        dbg[8 * blockIdx.x + 3] = (int32_t)(pt0 & 0x7fffffff);
        dbg[8 * blockIdx.x + 4] = (int32_t)(pt2 - pt1);
        dbg[8 * blockIdx.x + 5] = (int32_t)(pt3 - pt2);
        dbg[8 * blockIdx.x + 6] = (int32_t)(wall_clock64() - pt3);
        dbg[8 * blockIdx.x + 7] = 0;
    }
#endif
}

// One ray of trace_ray_lane as a state machine (round 3).  What the walk costs is INSTRUCTION ISSUE, not latency and not the LDS
// atomics: a 256-ray workgroup took 7.7 us with and 7.4 us without its atomics, and two interleaved rays per lane took 12.4 us
// (ablations in profiles/r03_notes.md) -- five waves per SIMD already fill its issue slots with these dependent integer chains.
// So the step is kept short: branch-free, and WITHOUT the three bounds tests and the two minor coordinates when the camera voxel itself lies
// in the grid (INB: both ends of every ray are in the grid then, hence -- the grid is convex, the coordinates monotone -- all of it).
template <bool INB>
struct RayWalk {
    int pa, pb, pc, l, p1, p2, left /*voxels still to emit: da + 1 - i*/;
    int two_db, two_dc, dl1 /*2 db - 2 da*/, dl2, sa, sb, sc, la, lb, lc;
    __device__ __forceinline__ void init(const int (&src)[3], int lin_t, int g, int gg, bool live)
    {
        int tgt[3];
        tgt[0] = lin_t / gg;
        const int rem = lin_t - tgt[0] * gg;
        tgt[1] = rem / g;
        tgt[2] = rem - tgt[1] * g;
        const int d0 = abs(tgt[0] - src[0]), d1 = abs(tgt[1] - src[1]), d2 = abs(tgt[2] - src[2]);
        const int dm = max(max(d0, d1), d2);
        // dominant axis tested x, y, z (utils.py:69,102,133); minors keep the reference's order
        const bool ax = dm == d0, ay = !ax && dm == d1;
        pa = ax ? src[0] : (ay ? src[1] : src[2]);
        pb = ax ? src[1] : src[0];
        pc = (ax || ay) ? src[2] : src[1];
        const int ta = ax ? tgt[0] : (ay ? tgt[1] : tgt[2]), tb = ax ? tgt[1] : tgt[0], tc = (ax || ay) ? tgt[2] : tgt[1];
        const int da = dm, db = ax ? d1 : d0, dc = (ax || ay) ? d2 : d1;
        const int st_a = ax ? gg : (ay ? g : 1), st_b = ax ? g : gg, st_c = (ax || ay) ? 1 : g;
        sa = pa < ta ? 1 : -1; sb = pb < tb ? 1 : -1; sc = pc < tc ? 1 : -1;
        two_db = 2 * db; two_dc = 2 * dc;
        dl1 = two_db - 2 * da; dl2 = two_dc - 2 * da;
        p1 = two_db - da; p2 = two_dc - da;
        l = pa * st_a + pb * st_b + pc * st_c;
        la = sa * st_a; lb = sb * st_b; lc = sc * st_c;
        left = live ? da + 1 : 0;
    }
    // emit the current voxel (if any is left and it lies in the grid), then advance:  p >= 0 ? (minor += s, p += 2 d - 2 da) : p += 2 d
    __device__ __forceinline__ void step(unsigned ug, uint32_t *s_path)
    {
        bool ok = left > 0;
        if (!INB) ok = ok && (unsigned)pa < ug && (unsigned)pb < ug && (unsigned)pc < ug;
        atomicOr(&s_path[ok ? l >> 5 : 0], ok ? 1u << (l & 31) : 0u);  // (a step outside the grid ORs a zero into word 0: rare, and free of branches)
        const bool ib = p1 >= 0, ic = p2 >= 0;
        l += (ib ? lb : 0) + (ic ? lc : 0) + la;
        p1 += ib ? dl1 : two_db;
        p2 += ic ? dl2 : two_dc;
        if (!INB) {
            pb += ib ? sb : 0;
            pc += ic ? sc : 0;
            pa += sa;
        }
        --left;
    }
};

template <bool INB>
__device__ __forceinline__ void walk_slice(const int (&src)[3], const int32_t *__restrict__ list, int cnt, int first, int stride, int g, int gg,
                                           uint32_t *s_path)
{
    // (the generic walk: a source voxel outside the grid.  Neighbouring lanes walk neighbouring voxels' rays -- same-address LDS atomics;
    // poses are clipped to the grid's range, so this is the rare path)
    const unsigned ug = (unsigned)g;
    for (int r = first; r < cnt; r += stride) {
        RayWalk<INB> rw;
        rw.init(src, list[r], g, gg, true);
        for (int i = rw.left; i > 0; --i) rw.step(ug, s_path);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The walk of a ray whose source voxel lies in the grid, as PACKED 16-bit arithmetic (round 5).  The walk is bound by
// instruction issue (profiles/r03_notes.md), so the step is written for the instruction count: the branch-free form above is
// 17 VALU per step, this one 9.
//
//   P  = (p1, p2)  the two error terms of utils.py:72-73 as one i16 pair (|p| < 2 G);   m = P >> 15 = (-1 where p < 0, else 0)
//   P' = m * (-2 da, -2 da) + (P + (2 db - 2 da, 2 dc - 2 da))     p >= 0: p + 2 d - 2 da;   p < 0: p + 2 d     (utils.py:76-88)
//   W  = l + (left << 24):  linear voxel index (< 2^24) and the steps still to take in ONE register;
//   W' = W + dot2((lb, lc), m) + (la + lb + lc - 2^24)  =  l + la + (p1 >= 0 ? lb : 0) + (p2 >= 0 ? lc : 0),  left - 1
//        (la / lb / lc = the axes' strides in the linear index times their step signs; v_dot2_i32_i16 adds -lb, -lc where p < 0)
//
// and the loop is "advance, then emit" (the ray's FIRST voxel is the source voxel, the same for every ray of the env: the
// workgroup sets that bit once instead of a 64-way same-address atomic per wave), so l only ever takes the values of emitted
// voxels -- all inside the grid, since both ends are and the coordinates are monotone -- and a borrow can never reach the count.
// Same integer recurrence as RayWalk / the reference kernel, voxel for voxel (tests/test_voxel_gpu.py: every path mask against the
// oracle; tools/check_packed_walk.py restates the packing in Python against the plain walk).  Limits: G <= 181 (gg as i16).
// ---------------------------------------------------------------------------------------------------------------------------
typedef short v2s_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s_t pack2(int lo, int hi) { return __builtin_bit_cast(v2s_t, (lo & 0xffff) | (hi << 16)); }

__device__ __forceinline__ void walk_packed(int sx, int sy, int sz, int l_src, int lin_t, int g, int gg, float inv_g, float inv_gg,
                                            uint32_t *s_path)
{
    int tx, ty, tz;
    if ((g & (g - 1)) == 0) {  // (uniform) a power-of-two edge: shifts
        const int sh = __builtin_ctz(g);
        tx = lin_t >> (2 * sh); ty = (lin_t >> sh) & (g - 1); tz = lin_t & (g - 1);
    } else {
        tx = floor_div_small(lin_t, gg, inv_gg);
        const int rem = lin_t - tx * gg;
        ty = floor_div_small(rem, g, inv_g);
        tz = rem - ty * g;
    }
    const int d0 = abs(tx - sx), d1 = abs(ty - sy), d2 = abs(tz - sz);
    const int da = max(max(d0, d1), d2);
    // dominant axis tested x, y, z (utils.py:69,102,133); the minors keep the reference's order; an axis that does not move has
    // d = 0 and never steps, whatever its sign
    const bool ax = da == d0, ay = !ax && da == d1;
    const int l0 = sx < tx ? gg : -gg, l1 = sy < ty ? g : -g, l2 = sz < tz ? 1 : -1;
    const int la = ax ? l0 : (ay ? l1 : l2), lb = ax ? l1 : l0, lc = (ax || ay) ? l2 : l1;
    const int db = ax ? d1 : d0, dc = (ax || ay) ? d2 : d1;
    // remaining steps of the dominant axis in the top byte, the linear voxel index below it.  UNSIGNED: da reaches g - 1 = 180 at the
    // kernel's limit (g <= 181), and as a signed word a count of 128 or more would read as "no steps left" 
    unsigned W = (unsigned)l_src + ((unsigned)da << 24);
    v2s_t P = pack2(2 * db - da, 2 * dc - da);
    const v2s_t neg2da = pack2(-2 * da, -2 * da), dl = pack2(2 * db - 2 * da, 2 * dc - 2 * da), lbc = pack2(lb, lc);
    const int kp = la + lb + lc - (1 << 24);
    while (W >= (1u << 24)) {
        const v2s_t m = P >> (v2s_t)(15);
        P = m * neg2da + (P + dl);
        W = (unsigned)__builtin_amdgcn_sdot2(lbc, m, (int)W, false) + (unsigned)kp;
        // Neighbouring lanes walk neighbouring voxels' rays (list order = voxel order), which run through the SAME voxel for most of
        // their length: a lane whose voxel is its left neighbour's leaves the bit to it.  (An LDS atomic of 64 lanes on one address
        // costs ~450 cycles of the CU's LDS pipe, on 64 scattered addresses ~16: tools/ubench/ray_step_rate.hip.  A lane that is
        // switched off -- its ray has ended -- reads as -1: no voxel.)
        const int l = (int)(W & 0xffffffu);
        const int lp = __builtin_amdgcn_update_dpp(-1, l, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
#ifdef RAY_ABL_NOATOM  // (measurement build: the walk without its LDS atomics -- and with nothing to flush: 20.7 -> 13.4 us)
        if (l == 0x7fffff) atomicOr(&s_path[l >> 5], 1u << (l & 31));
#else
        if (l != lp) atomicOr(&s_path[l >> 5], 1u << (l & 31));
#endif
    }
}

// launch 2: load-balanced ray cast over the ray lists (see the header above).
//
// Work items = slices of `width` = 256 rpl rays of one env's list.  Which envs have how many slices is only known on the device, and
// a workgroup that finds nothing to do still costs a dispatch slot: the chip starts ~40 workgroups per us and XCD whatever they do,
// so a grid of N x 16 (env, slice) pairs of which a third is live lasts >= 14 us however fast the live ones are (round-4 layout;
// profiles/r05_notes.md).  Round 5: the grid is COMPACT.  Block b runs on XCD b % 8 and takes item j = b / 8 of that XCD's envs
// (e = 8 k + xcd: an env's path words stay in one XCD's L2): every wave reads the counts of those envs (<= 64 per pass), scans the
// slice counts and finds the (env, slice) that holds item j -- ~40 instructions per wave, no inter-workgroup traffic.  Blocks past
// the XCD's last item exit; they sit at the END of the dispatch order, behind every live workgroup.  A grid smaller than the item
// count (an env with tens of thousands of rays) makes its workgroups take several items.
// (Round 5, measured and not kept -- tools/ray_phase_rollout.py, profiles/r05_notes.md 1g: the launch ends with its heaviest ITEMS, 23-25 us
// where the mean item takes 11; rays walked in segments of <= 16 steps dealt evenly over the workgroup's lanes, so that no wave waits
// for its longest ray: item mean 13.0 / max 25.2 us, the update +1.1 us; items numbered over all envs, 115 per XCD instead of 96-136:
// last end 24-25 us all the same, the update +1.4 ... +2.8 us.  A heavy item is heavy in its LDS atomics, not in its longest ray.)
// (amdgpu_num_sgpr: with the 96 the compiler takes, a SIMD's SGPR file holds 7 waves; the six waves of a workgroup do not spread evenly
// over the four SIMDs, so some CUs then held three workgroups instead of five and a tenth of the launch's workgroups started when the
// first ones ended: -3.6 us per update, profiles/r05_notes.md)
__global__ __launch_bounds__(kListThreads) __attribute__((amdgpu_num_sgpr(80))) void k_ray_list(
    const int32_t *__restrict__ ray_count, const int32_t *__restrict__ ray_list, int64_t ray_cap, const float *__restrict__ poses_xyz,
    int64_t pose_stride, const float *__restrict__ range_gt, const float *__restrict__ voxel_size, int n, int g, int words,
    uint32_t *__restrict__ path_mask)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_path[];
    const int b = blockIdx.x, xcd = b & 7, per_xcd = (int)(gridDim.x >> 3);
    const int env_groups = (n + 7) >> 3;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    constexpr int width = kListThreads;  // rays per item: one per lane
    const int gg = g * g;
    const float inv_g = __frcp_rn((float)g), inv_gg = __frcp_rn((float)gg);
#ifdef PHASE_TIMING
    const uint64_t rt0 = wall_clock64();
    int32_t *dbg = const_cast<int32_t *>(ray_list) + (size_t)n * ray_cap - 8 * 512 - 8 * (size_t)(blockIdx.x + 1);
    bool first_item = true;
#endif
    for (int j = b >> 3;; j += per_xcd) {
        // ---- item j of this XCD -> (env, slice); the same in every wave ----
        int e = -1, sl = 0, cnt = 0, before = 0;
        for (int k0 = 0; k0 < env_groups; k0 += kWave) {
            const int ek = (k0 + lane) * 8 + xcd;
            const int c = (k0 + lane < env_groups && ek < n) ? (int)min((int64_t)ray_count[ek], ray_cap) : 0;
            const int items = (c + width - 1) / width;
            const int incl = wave_inclusive_scan(items);
            const int chunk = __builtin_amdgcn_readlane(incl, kWave - 1);
            if (j < before + chunk) {
                const int kk = __ffsll((unsigned long long)__ballot(before + incl > j)) - 1;
                e = (k0 + kk) * 8 + xcd;
                sl = j - before - (__builtin_amdgcn_readlane(incl, kk) - __builtin_amdgcn_readlane(items, kk));
                cnt = __builtin_amdgcn_readlane(c, kk);
                break;
            }
            before += chunk;
        }
        if (e < 0) {  // past the XCD's last item
#ifdef PHASE_TIMING
            if (tid == 0 && first_item) {
                dbg[0] = (int32_t)(rt0 & 0x7fffffff); dbg[6] = (int32_t)(wall_clock64() - rt0); dbg[5] = 2;
                dbg[2] = (int32_t)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4);   // HW_ID
                dbg[1] = (int32_t)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 20);  // XCC_ID
            }
#endif
            return;
        }
#ifdef PHASE_TIMING
        const uint64_t rt_cnt = wall_clock64();
#endif
        // every request of the item's prologue at once: the pose, the frame, the lane's first list entry
        const int32_t *list = ray_list + (size_t)e * ray_cap;
        const float *pp = poses_xyz + (size_t)e * pose_stride;
        const int r_first = sl * width + tid;
        const int32_t lin_first = list[min(r_first, cnt - 1)];
        const float pose0 = pp[0], pose1 = pp[1], pose2 = pp[2];
        const float rmin0 = range_gt[e * 6 + 1], rmin1 = range_gt[e * 6 + 3], rmin2 = range_gt[e * 6 + 5];
        const float vox0 = voxel_size[e * 3 + 0], vox1 = voxel_size[e * 3 + 1], vox2 = voxel_size[e * 3 + 2];
        {
            uint4 *s4 = reinterpret_cast<uint4 *>(s_path);  // (words is a multiple of 64)
            for (int i = tid; i < words / 4; i += kListThreads) s4[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        const int src[3] = {pose_axis_to_idx(pose0, rmin0, vox0), pose_axis_to_idx(pose1, rmin1, vox1), pose_axis_to_idx(pose2, rmin2, vox2)};
        __syncthreads();
        const bool src_in = (unsigned)src[0] < (unsigned)g && (unsigned)src[1] < (unsigned)g && (unsigned)src[2] < (unsigned)g;  // (per env: uniform)
#ifdef PHASE_TIMING
        const uint64_t rt_src = src_in ? wall_clock64() : 0;  // (the pose loads have arrived)
#endif
        const int r_end = min(cnt, (sl + 1) * width);
        if (src_in && g <= 181) {
            const int l_src = src[0] * gg + src[1] * g + src[2];
            if (tid == 0) atomicOr(&s_path[l_src >> 5], 1u << (l_src & 31));  // every ray's first voxel
            for (int r = r_first; r < r_end; r += kListThreads)
                walk_packed(src[0], src[1], src[2], l_src, r == r_first ? lin_first : list[r], g, gg, inv_g, inv_gg, s_path);
        } else if (src_in) {
            walk_slice<true>(src, list, r_end, r_first, kListThreads, g, gg, s_path);
        } else {
            walk_slice<false>(src, list, r_end, r_first, kListThreads, g, gg, s_path);
        }
#ifdef PHASE_TIMING
        const uint64_t rt1w = wall_clock64();  // this wave's walk
#endif
        __syncthreads();
        uint32_t *gp = path_mask + (size_t)e * words;
        {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(s_path);
            for (int i = tid; i < words / 4; i += kListThreads) {
                const uint4 v = s4[i];
                if (v.x | v.y | v.z | v.w) {
                    if (v.x) atomicOr(&gp[4 * i + 0], v.x);
                    if (v.y) atomicOr(&gp[4 * i + 1], v.y);
                    if (v.z) atomicOr(&gp[4 * i + 2], v.z);
                    if (v.w) atomicOr(&gp[4 * i + 3], v.w);
                }
            }
        }
        __syncthreads();  // (the mask is cleared again for the next item)
#ifdef PHASE_TIMING
        if (tid == 0 && first_item) {
            dbg[0] = (int32_t)(rt0 & 0x7fffffff);
            dbg[1] = (int32_t)(rt1w - rt0);
            dbg[2] = (int32_t)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4);   // HW_ID
            dbg[3] = (int32_t)(wall_clock64() - rt0);
            dbg[4] = (int32_t)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 20);  // XCC_ID
            dbg[5] = 1;
            dbg[6] = (int32_t)(rt_cnt - rt0);
            dbg[7] = (int32_t)(rt_src - rt0);
        }
        first_item = false;
#endif
    }
}

// ===========================================================================
// large grids (G > 104: a 128^3 bitmask is 256 KiB, no workgroup can hold it next to its lists), launches 1 + 2 (round 3)
//
// k_hit_atomic: the phase A of k_hit_list -- same stream, same predictor, same queue of undecided pixels -- WITHOUT a mask in LDS: a
// pixel's voxel bit is ORed into the env's hit mask in global memory (all workgroups of an env run on one XCD: its L2 serves them),
// and the atomic's RETURN value says whether this pixel was the first to set it -- exactly then the voxel is appended to the env's
// ray list (wave-aggregated slot allocation).  Set semantics are exact (one list entry per distinct voxel, as the reference's
// torch.unique gives, utils.py:230-270); the four atomics of a lane's pixel group are issued before the first result is used.
// (The windowed round-1 k_hit_mask ran the canonical chain over every pixel once per 128 KiB window: 0.79 ms at 512 x 128^3.)
//
// k_ray_slab: workgroup (e, slab, slice) owns the x-planes [X0, X1) of env e's path mask (<= 32 KiB of LDS at 128^3: 16 planes) and
// walks, of every ray of its slice, ONLY the steps whose voxel lies in that slab.  The reference's integer Bresenham (utils.py:48-167)
// has a closed form -- point j of a ray is a_j = a_0 + s_a j on the dominant axis and b_j = b_0 + s_b floor((2 d_b j + d_a) / (2 d_a))
// on a minor axis, error term p_j = 2 d_b (j + 1) - d_a - 2 d_a nb_j (induction on p_j in [2 d_b - 2 d_a, 2 d_b)) -- so the first and
// last step inside a slab and the walker's state there follow from four small integer divisions; the steps in between are the
// sequential walk (RayWalk::step), bit for bit.  No step is walked twice (the windowed k_raycast walked every ray once per window),
// and the mask of a workgroup is as small as at 64^3.
// ===========================================================================
__device__ __forceinline__ int udiv_small(int n, int d)  // floor(n / d), 0 <= n < 2^22, d > 0: reciprocal estimate + two-sided correction
{
    int q = (int)(__fmul_rn((float)n, __builtin_amdgcn_rcpf((float)d)));
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) ++q;
    return q;
}

template <bool KFAST>
__global__ __launch_bounds__(kFusedThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_hit_atomic(
    const float *__restrict__ depth_raw, const float *__restrict__ seg_raw, const float *__restrict__ c2w, Intrinsics K,
    const float *__restrict__ range_gt, const float *__restrict__ voxel_size, int n, int h, int w, int g, float sense_dist,
    int chunks, int words, uint32_t *__restrict__ hit_mask, int32_t *__restrict__ ray_count, int32_t *__restrict__ ray_list,
    int64_t ray_cap, int32_t *__restrict__ coverage_zero)
{
    __shared__ int s_qcnt, s_cnt;
    __shared__ int s_queue[kQueueCapPx];
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int e = (slot / chunks) * 8 + xcd;
    const int c = slot % chunks;
    if (e >= n) return;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const bool own_env = chunks == 1;  // this workgroup lists the whole env: its ray count lives in LDS until the end
    if (tid == 0) { s_qcnt = 0; s_cnt = 0; }
    if (coverage_zero != nullptr && c == 0 && tid == 0) coverage_zero[e] = 0;
    const int hw = h * w;
    int ppc = (hw + chunks - 1) / chunks;
    ppc = (ppc + 3) & ~3;
    const int px0 = c * ppc, px1 = min(hw, px0 + ppc);
    const float *dptr = depth_raw + (size_t)e * hw;
    const float *sptr = seg_raw + (size_t)e * hw;
    const float inv_w = __frcp_rn((float)w);
    uint32_t *gh = hit_mask + (size_t)e * words;
    int32_t *list = ray_list + (size_t)e * ray_cap;
    int32_t *cnt_e = ray_count + e;
    // OR the bits of up to four voxels (l < 0: none) into the env's mask; the voxels this call set first go to the ray list.
    // Called by converged code paths and by loop tails alike: ballots see the active lanes only, the leader is one of them.
    auto mark4 = [&](int l0, int l1, int l2, int l3) {
        const int ls[4] = {l0, l1, l2, l3};
        uint32_t old[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) old[k] = ls[k] >= 0 ? atomicOr(&gh[ls[k] >> 5], 1u << (ls[k] & 31)) : 0xffffffffu;
        // ONE slot reservation per wave and call: the four ballots first, then the leader's add (an env owned by one workgroup counts in
        // LDS; the workgroups of a shared env add to its global counter -- the same address for all of them, so every add saved counts)
        bool first[4];
        unsigned long long m[4];
        int total = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            first[k] = ls[k] >= 0 && !((old[k] >> (ls[k] & 31)) & 1u);
            m[k] = __ballot(first[k]);
            total += __popcll(m[k]);
        }
        if (total) {
            const unsigned long long act = __ballot(true);
            const int leader = __ffsll((long long)act) - 1;
            int base = 0;
            if (lane == leader) base = own_env ? atomicAdd(&s_cnt, total) : atomicAdd(cnt_e, total);
            base = __builtin_amdgcn_readlane(base, leader);
            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = base + __popcll(m[k] & below);
                if (first[k] && o < ray_cap) list[o] = ls[k];
                base += __popcll(m[k]);
            }
        }
    };
    __syncthreads();
    bool need_full_exact = false;
    if (KFAST && (w & 3) == 0 && hw < (1 << 23) && kUsePredictor) {
        const VoxPredict vp = make_predictor(c2w + (size_t)e * 16, K, range_gt + e * 6, voxel_size + e * 3, g, h, w);
        const unsigned ug = (unsigned)g;
        constexpr int kTile = kFusedThreads * 4, kAhead = 2;  // (two tiles in flight: the atomics, not the stream, pace this kernel; 64 VGPRs)
        v4f_t dq[kAhead], sq[kAhead];
        const int plast = px1 - 4;
        const int ntiles = (px1 - px0 + kTile - 1) / kTile;
        auto tile_px = [&](int t) { return px0 + t * kTile + tid * 4; };
#pragma unroll
        for (int q = 0; q < kAhead; ++q) {
            const int pq = max(min(tile_px(max(min(q, ntiles - 1), 0)), plast), 0);
            ld4_stream_async(dq[q], dptr + pq);
            ld4_stream_async(sq[q], sptr + pq);
        }
        const int tdy = kTile / w, tdx = kTile - tdy * w;
        int py = floor_div_small(max(min(tile_px(0), plast), 0), w, inv_w), pxc = max(min(tile_px(0), plast), 0) - py * w;
        for (int t0 = 0; t0 < ntiles; t0 += kAhead) {
#pragma unroll
            for (int q = 0; q < kAhead; ++q) {
                const int t = t0 + q;
                const int p = t < ntiles ? tile_px(t) : px1;
                wait_vm_keep<2 * (kAhead - 1)>(dq[q], sq[q]);
                const v4f_t d4 = dq[q], s4 = sq[q];
                if (p < px1 && __any((s4.x > 50.0f) | (s4.y > 50.0f) | (s4.z > 50.0f) | (s4.w > 50.0f))) {
                    const float fy = (float)py, fx = (float)pxc;
                    const float row[3] = {__fmaf_rn(vp.be[0], fy, vp.ga[0]), __fmaf_rn(vp.be[1], fy, vp.ga[1]), __fmaf_rn(vp.be[2], fy, vp.ga[2])};
                    bool u0, u1, u2, u3;
                    const int l0 = pixel_predict(vp, row, fx, d4.x, s4.x, sense_dist, ug, u0);
                    const int l1 = pixel_predict(vp, row, fx + 1.0f, d4.y, s4.y, sense_dist, ug, u1);
                    const int l2 = pixel_predict(vp, row, fx + 2.0f, d4.z, s4.z, sense_dist, ug, u2);
                    const int l3 = pixel_predict(vp, row, fx + 3.0f, d4.w, s4.w, sense_dist, ug, u3);
                    const int pl = __builtin_amdgcn_update_dpp(-2, l3, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
                    mark4(l0 != pl ? l0 : -1, l1 != l0 ? l1 : -1, l2 != l1 ? l2 : -1, l3 != l2 ? l3 : -1);
                    if (__any(u0 | u1 | u2 | u3)) {
                        const int cnt = (int)u0 + (int)u1 + (int)u2 + (int)u3;
                        if (cnt) {
                            int qs = atomicAdd(&s_qcnt, cnt);
                            if (u0) { if (qs < kQueueCapPx) s_queue[qs] = p; ++qs; }
                            if (u1) { if (qs < kQueueCapPx) s_queue[qs] = p + 1; ++qs; }
                            if (u2) { if (qs < kQueueCapPx) s_queue[qs] = p + 2; ++qs; }
                            if (u3) { if (qs < kQueueCapPx) s_queue[qs] = p + 3; }
                        }
                    }
                }
                pxc += tdx; py += tdy;
                if (pxc >= w) { pxc -= w; ++py; }
                // refill THIS slot behind its last use (see k_hit_list)
                const int pn = max(min(tile_px(max(min(t + kAhead, ntiles - 1), 0)), plast), 0);
                ld4_stream_async(dq[q], dptr + pn);
                ld4_stream_async(sq[q], sptr + pn);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nq = s_qcnt;
        if (nq > kQueueCapPx) {
            need_full_exact = true;  // (re-marking a voxel the stream already set is harmless: the atomic reports "not first")
        } else if (nq > 0) {
            PixelFrame pf;
            load_pixel_frame(pf, c2w, range_gt, voxel_size, e, g, sense_dist);
            for (int j = tid; j < nq; j += kFusedThreads) {
                const int p = s_queue[j];
                const int y = p / w;
                mark4(pixel_to_lin<KFAST>(pf, K, (float)(p - y * w), (float)y, dptr[p], sptr[p]), -1, -1, -1);
            }
        }
    } else {
        need_full_exact = true;
    }
    if (need_full_exact) {
        PixelFrame pf;
        load_pixel_frame(pf, c2w, range_gt, voxel_size, e, g, sense_dist);
        for (int p = px0 + tid; p < px1; p += kFusedThreads) {
            const int y = p / w;
            mark4(pixel_to_lin<KFAST>(pf, K, (float)(p - y * w), (float)y, dptr[p], sptr[p]), -1, -1, -1);
        }
    }
    if (own_env) {
        __syncthreads();
        if (tid == 0) *cnt_e = s_cnt;
    }
}

// One ray restricted to the x-planes [X0, X1): RayWalk's state at the first point inside, `left` = points inside (see the header).
__device__ __forceinline__ int udiv_rcp(int n, int d, float rcp_d)  // the same with the reciprocal estimate of d at hand
{
    int q = (int)(__fmul_rn((float)n, rcp_d));
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) ++q;
    return q;
}
// per-workgroup constants of the slab walk: reciprocals of g^2 and g (grids below 2^22 voxels: the target's coordinates by two
// multiplications instead of two integer divisions) and of the ray count (the lanes' permutation of the list in 32-bit arithmetic)
struct SlabConsts {
    float rgg, rg;
    bool small_grid, perm32;
    uint32_t cnt_magic, P;
};
__device__ __forceinline__ SlabConsts slab_consts(int g, int gg, int cnt)
{
    SlabConsts k;
    k.rgg = __builtin_amdgcn_rcpf((float)gg);
    k.rg = __builtin_amdgcn_rcpf((float)g);
    k.small_grid = (int64_t)gg * g < (1 << 22);
    k.P = (cnt % 7919) ? 7919u : 7907u;  // (as walk_slice: neighbouring list entries are neighbouring voxels)
    k.perm32 = (uint64_t)cnt * k.P < (1ull << 32);
    k.cnt_magic = cnt > 0 ? 0xffffffffu / (uint32_t)cnt : 0u;
    return k;
}
// (r * P) mod cnt
__device__ __forceinline__ int perm_index(int r, int cnt, const SlabConsts &k)
{
    if (k.perm32) {
        const uint32_t x = (uint32_t)r * k.P;
        uint32_t rem = x - __umulhi(x, k.cnt_magic) * (uint32_t)cnt;  // the estimate is the quotient or one below it
        if (rem >= (uint32_t)cnt) rem -= (uint32_t)cnt;
        return (int)rem;
    }
    return (int)(((int64_t)r * (int64_t)k.P) % cnt);
}

template <bool INB>
__device__ __forceinline__ void init_ray_slab(RayWalk<INB> &rw, const int (&src)[3], int lin_t, int g, int gg, int X0, int X1, const SlabConsts &k)
{
    int tgt[3];
    tgt[0] = k.small_grid ? udiv_rcp(lin_t, gg, k.rgg) : lin_t / gg;
    rw.left = 0;
    if (max(src[0], tgt[0]) < X0 || min(src[0], tgt[0]) >= X1) return;  // the ray's x-range misses the slab
    const int rem = lin_t - tgt[0] * gg;
    tgt[1] = k.small_grid ? udiv_rcp(rem, g, k.rg) : rem / g;
    tgt[2] = rem - tgt[1] * g;
    const int d0 = abs(tgt[0] - src[0]), d1 = abs(tgt[1] - src[1]), d2 = abs(tgt[2] - src[2]);
    const int dm = max(max(d0, d1), d2);
    const bool ax = dm == d0, ay = !ax && dm == d1;  // dominant axis tested x, y, z (utils.py:69,102,133)
    const int pa0 = ax ? src[0] : (ay ? src[1] : src[2]), pb0 = ax ? src[1] : src[0], pc0 = (ax || ay) ? src[2] : src[1];
    const int ta = ax ? tgt[0] : (ay ? tgt[1] : tgt[2]), tb = ax ? tgt[1] : tgt[0], tc = (ax || ay) ? tgt[2] : tgt[1];
    const int da = dm, db = ax ? d1 : d0, dc = (ax || ay) ? d2 : d1;
    const int st_a = ax ? gg : (ay ? g : 1), st_b = ax ? g : gg, st_c = (ax || ay) ? 1 : g;
    rw.sa = pa0 < ta ? 1 : -1; rw.sb = pb0 < tb ? 1 : -1; rw.sc = pc0 < tc ? 1 : -1;
    // x advances by k in [klo, khi] x-steps while it is inside the slab (x is the dominant axis, or the FIRST minor axis)
    const int sx = src[0] < tgt[0] ? 1 : -1;
    const int klo = max(sx > 0 ? X0 - src[0] : src[0] - (X1 - 1), 0), khi = min(sx > 0 ? (X1 - 1) - src[0] : src[0] - X0, d0);
    int jlo = 0, jhi = da;
    if (ax) {
        jlo = klo; jhi = khi;
    } else if (d0 > 0) {
        // nx(j) = floor((2 dx j + da) / (2 da)); the first j with nx(j) >= k (k >= 1) is ceil((2 da k - da) / (2 dx))
        if (klo >= 1) jlo = udiv_small(2 * da * klo - da + 2 * d0 - 1, 2 * d0);
        jhi = min(da, udiv_small(2 * da * (khi + 1) - da + 2 * d0 - 1, 2 * d0) - 1);
    }
    if (jlo > jhi) return;
    const int nb = da > 0 ? udiv_small(2 * db * jlo + da, 2 * da) : 0, nc = da > 0 ? udiv_small(2 * dc * jlo + da, 2 * da) : 0;
    rw.pa = pa0 + rw.sa * jlo; rw.pb = pb0 + rw.sb * nb; rw.pc = pc0 + rw.sc * nc;
    rw.two_db = 2 * db; rw.two_dc = 2 * dc;
    rw.dl1 = rw.two_db - 2 * da; rw.dl2 = rw.two_dc - 2 * da;
    rw.p1 = rw.two_db * (jlo + 1) - da - 2 * da * nb; rw.p2 = rw.two_dc * (jlo + 1) - da - 2 * da * nc;
    rw.l = rw.pa * st_a + rw.pb * st_b + rw.pc * st_c - X0 * gg;  // bit index inside the slab's mask
    rw.la = rw.sa * st_a; rw.lb = rw.sb * st_b; rw.lc = rw.sc * st_c;
    rw.left = jhi - jlo + 1;
}

template <bool INB>
__device__ __forceinline__ void walk_slab(const int (&src)[3], const int32_t *__restrict__ list, int cnt, int first, int stride, int g, int gg,
                                          int X0, int X1, uint32_t *s_path)
{
    const SlabConsts k = slab_consts(g, gg, cnt);
    const unsigned ug = (unsigned)g;
    for (int r = first; r < cnt; r += stride) {
        RayWalk<INB> rw;
        init_ray_slab<INB>(rw, src, list[perm_index(r, cnt, k)], g, gg, X0, X1, k);
        for (int i = rw.left; i > 0; --i) rw.step(ug, s_path);
    }
}

__global__ __launch_bounds__(kSlabThreads) void k_ray_slab(
    const int32_t *__restrict__ ray_count, const int32_t *__restrict__ ray_list, int64_t ray_cap, const float *__restrict__ poses_xyz,
    int64_t pose_stride, const float *__restrict__ range_gt, const float *__restrict__ voxel_size, int n, int g, int words, int slabs,
    int slab_planes, int slices, uint32_t *__restrict__ path_mask)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_path[];
    // block -> (env, slab, slice); all workgroups of env e run on XCD e % 8.  (`slices` per slab: a workgroup pays ~2 us for clearing
    // and flushing its mask whatever it walks -- 16 slices x 8 slabs x 512 envs were 65 536 workgroups, a quarter of the launch.)
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int per_env = slabs * slices;
    const int e = (slot / per_env) * 8 + xcd;
    const int rem = slot % per_env, slab = rem / slices, sl = rem % slices;
    if (e >= n) return;
    const int cnt = (int)min((int64_t)ray_count[e], ray_cap);
    if (sl * kSlabThreads >= cnt) return;
    const int tid = threadIdx.x, gg = g * g;
    const int X0 = slab * slab_planes, X1 = min(g, X0 + slab_planes);
    const int w0 = (int)(((int64_t)X0 * gg) >> 5);  // (X0 * gg % 32 == 0: the host picks slab_planes that way)
    const int nw = min(words, (int)(((int64_t)X1 * gg + 31) >> 5)) - w0;
    for (int i = tid; i < nw; i += kSlabThreads) s_path[i] = 0u;
    const float *pp = poses_xyz + (size_t)e * pose_stride;
    const int src[3] = {pose_axis_to_idx(pp[0], range_gt[e * 6 + 1], voxel_size[e * 3 + 0]),
                        pose_axis_to_idx(pp[1], range_gt[e * 6 + 3], voxel_size[e * 3 + 1]),
                        pose_axis_to_idx(pp[2], range_gt[e * 6 + 5], voxel_size[e * 3 + 2])};
    const int32_t *list = ray_list + (size_t)e * ray_cap;
    __syncthreads();
    const bool src_in = (unsigned)src[0] < (unsigned)g && (unsigned)src[1] < (unsigned)g && (unsigned)src[2] < (unsigned)g;
    if (src_in)
        walk_slab<true>(src, list, cnt, sl * kSlabThreads + tid, slices * kSlabThreads, g, gg, X0, X1, s_path);
    else
        walk_slab<false>(src, list, cnt, sl * kSlabThreads + tid, slices * kSlabThreads, g, gg, X0, X1, s_path);
    __syncthreads();
    uint32_t *gp = path_mask + (size_t)e * words + w0;
    for (int i = tid; i < nw; i += kSlabThreads) {
        const uint32_t v = s_path[i];
        if (v) atomicOr(&gp[i], v);
    }
}

// ===========================================================================
// fused path, launch 3: streaming grid update (A6 tail + A7 + coverage count)
// ===========================================================================
constexpr int kGridThreads = 256;

__device__ __forceinline__ void update_voxel(float &prob, float &scan, float gt, bool hit, bool path, float &tri, int &cov)
{
    float pr = prob;
    if (path) pr = __fsub_rn(pr, 0.05f);
    if (hit) pr = 1.0f;
    prob = pr;
    tri = (pr > 0.5f ? 1.0f : 0.0f) - (pr < 0.0f ? 1.0f : 0.0f);
    float sc = __fadd_rn(scan, __fmul_rn(hit ? 1.0f : 0.0f, gt));
    sc = sc < 0.0f ? 0.0f : (sc > 1.0f ? 1.0f : sc);  // torch.clip(min=0,max=1); NaN propagates like torch
    scan = sc;
    cov += (sc != 0.0f) ? 1 : 0;
}

template <bool VEC4>
__global__ __launch_bounds__(kGridThreads) void k_grid_update(
    const uint32_t *__restrict__ hit_mask, const uint32_t *__restrict__ path_mask, const float *__restrict__ grid_gt,
    const uint8_t *__restrict__ reset_mask, int n, int g3, int words, float *__restrict__ prob_grid,
    float *__restrict__ scanned, float *__restrict__ tri_out, int64_t tri_stride, int32_t *__restrict__ coverage)
{
    const int e = blockIdx.y;
    const bool reset = reset_mask != nullptr && reset_mask[e] != 0;
    const uint32_t *hm = hit_mask + (size_t)e * words, *pm = path_mask + (size_t)e * words;
    float *prob = prob_grid + (size_t)e * g3, *scan = scanned + (size_t)e * g3;
    const float *gt = grid_gt + (size_t)e * g3;
    float *tri = tri_out + (size_t)e * tri_stride;
    int cov = 0;
    if (VEC4) {
        const int nv = g3 >> 2;
        for (int i = blockIdx.x * kGridThreads + threadIdx.x; i < nv; i += gridDim.x * kGridThreads) {
            const int v0 = i << 2;
            const uint32_t hb = (hm[v0 >> 5] >> (v0 & 31)) & 0xFu, pb = (pm[v0 >> 5] >> (v0 & 31)) & 0xFu;
            float4 p4 = reset ? make_float4(0, 0, 0, 0) : reinterpret_cast<const float4 *>(prob)[i];
            float4 s4 = reset ? make_float4(0, 0, 0, 0) : reinterpret_cast<const float4 *>(scan)[i];
            const float4 g4 = reinterpret_cast<const float4 *>(gt)[i];
            float4 t4;
            update_voxel(p4.x, s4.x, g4.x, hb & 1u, pb & 1u, t4.x, cov);
            update_voxel(p4.y, s4.y, g4.y, hb & 2u, pb & 2u, t4.y, cov);
            update_voxel(p4.z, s4.z, g4.z, hb & 4u, pb & 4u, t4.z, cov);
            update_voxel(p4.w, s4.w, g4.w, hb & 8u, pb & 8u, t4.w, cov);
            reinterpret_cast<float4 *>(prob)[i] = p4;
            reinterpret_cast<float4 *>(scan)[i] = s4;
            reinterpret_cast<float4 *>(tri)[i] = t4;
        }
    } else {
        for (int v = blockIdx.x * kGridThreads + threadIdx.x; v < g3; v += gridDim.x * kGridThreads) {
            const bool hb = (hm[v >> 5] >> (v & 31)) & 1u, pb = (pm[v >> 5] >> (v & 31)) & 1u;
            float p = reset ? 0.0f : prob[v], s = reset ? 0.0f : scan[v], t;
            update_voxel(p, s, gt[v], hb, pb, t, cov);
            prob[v] = p;
            scan[v] = s;
            tri[v] = t;
        }
    }
    __shared__ int s_cov[kGridThreads / kWave];
    cov = wave_reduce_sum(cov);
    if ((threadIdx.x & (kWave - 1)) == 0) s_cov[threadIdx.x / kWave] = cov;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int i = 0; i < kGridThreads / kWave; ++i) t += s_cov[i];
        if (t) atomicAdd(&coverage[e], t);
    }
}

// ---------------------------------------------------------------------------
// Packed variant: the ground truth and the scanned set are BITMASKS ([N, words] u32).  For a
// binary grid_gt (the reference's GT is an occupancy indicator, env_train_gennbv.py:64-66)
//   scanned = clip(scanned + occ*gt, 0, 1)   <=>   scanned_bits |= hit_bits & gt_bits
// exactly, and sum(scanned) is a popcount.  Same results, half the HBM traffic of the streaming
// pass: prob (R+W) + tri (W) = 12 B/voxel instead of 24.
// ---------------------------------------------------------------------------
template <bool VEC4>
__global__ __launch_bounds__(kGridThreads) void k_grid_update_packed(
    const uint32_t *__restrict__ hit_mask, const uint32_t *__restrict__ path_mask, const uint32_t *__restrict__ gt_bits,
    const uint8_t *__restrict__ reset_mask, int n, int g3, int words, int words_gt, float *__restrict__ prob_grid,
    uint32_t *__restrict__ scanned_bits, float *__restrict__ tri_out, int64_t tri_stride, int32_t *__restrict__ coverage)
{
    const int e = blockIdx.y;
    const bool reset = reset_mask != nullptr && reset_mask[e] != 0;
    const uint32_t *hm = hit_mask + (size_t)e * words, *pm = path_mask + (size_t)e * words;
    const uint32_t *gb = gt_bits + (size_t)e * words_gt;
    uint32_t *sb = scanned_bits + (size_t)e * words_gt;
    float *prob = prob_grid + (size_t)e * g3;
    float *tri = tri_out + (size_t)e * tri_stride;
    int cov = 0;
    auto voxel = [](float &pr, bool hit, bool path, float &t) {
        if (path) pr = __fsub_rn(pr, 0.05f);
        if (hit) pr = 1.0f;
        t = (pr > 0.5f ? 1.0f : 0.0f) - (pr < 0.0f ? 1.0f : 0.0f);
    };
    if (VEC4) {
        const int nv = g3 >> 2;
        for (int i = blockIdx.x * kGridThreads + threadIdx.x; i < nv; i += gridDim.x * kGridThreads) {
            const int v0 = i << 2, wd = v0 >> 5;
            const uint32_t hw = hm[wd], pw = pm[wd];
            const uint32_t hb = (hw >> (v0 & 31)) & 0xFu, pb = (pw >> (v0 & 31)) & 0xFu;
            float4 p4 = reset ? make_float4(0, 0, 0, 0) : reinterpret_cast<const float4 *>(prob)[i];
            float4 t4;
            voxel(p4.x, hb & 1u, pb & 1u, t4.x);
            voxel(p4.y, hb & 2u, pb & 2u, t4.y);
            voxel(p4.z, hb & 4u, pb & 4u, t4.z);
            voxel(p4.w, hb & 8u, pb & 8u, t4.w);
            reinterpret_cast<float4 *>(prob)[i] = p4;
            reinterpret_cast<float4 *>(tri)[i] = t4;
            if ((v0 & 31) == 0) {  // one lane in eight owns the 32-voxel word of the scanned set
                const uint32_t sw = (reset ? 0u : sb[wd]) | (hw & gb[wd]);
                sb[wd] = sw;
                cov += __popc(sw);
            }
        }
    } else {
        for (int v = blockIdx.x * kGridThreads + threadIdx.x; v < g3; v += gridDim.x * kGridThreads) {
            const int wd = v >> 5;
            const uint32_t hw = hm[wd];
            const bool hb = (hw >> (v & 31)) & 1u, pb = (pm[wd] >> (v & 31)) & 1u;
            float p = reset ? 0.0f : prob[v], t;
            voxel(p, hb, pb, t);
            prob[v] = p;
            tri[v] = t;
            if ((v & 31) == 0) {
                const uint32_t sw = (reset ? 0u : sb[wd]) | (hw & gb[wd]);
                sb[wd] = sw;
                cov += __popc(sw);
            }
        }
    }
    __shared__ int s_cov[kGridThreads / kWave];
    cov = wave_reduce_sum(cov);
    if ((threadIdx.x & (kWave - 1)) == 0) s_cov[threadIdx.x / kWave] = cov;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int i = 0; i < kGridThreads / kWave; ++i) t += s_cov[i];
        if (t) atomicAdd(&coverage[e], t);
    }
}

// ---------------------------------------------------------------------------
// Coded variant: prob_grid as ONE BYTE per voxel.  Between two resets a voxel's probability is the result of
// "set to 1.0 on a hit" and "x <- fl32(x - 0.05) on a path step" (env_train_gennbv.py:305-312), i.e. a pure
// function of (base, k): base = 1 after a hit / 0 since the reset, k = path steps since then.
//   code = base << 7 | k   (k <= 127: an episode is at most max_episode_length <= 127 steps; saturation sets
//   *overflow).  value = prob_lut[code] (exact fp32 iteration, gnbv_prob_code_tables), tri = tri_lut[code].
// Traffic per voxel: code R+W (2 B) + tri W (4 B) + mask bits, instead of 12 B: the grid update is HBM-bound.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t step_code(uint32_t c, bool hit, bool path, int &ovf)
{
    if (path) {
        const uint32_t k = c & 127u;
        if (k == 127u) ovf = 1;
        c = (c & 128u) | (k < 127u ? k + 1u : 127u);
    }
    if (hit) c = 128u;
    return c;
}

// VPL voxels per lane and trip: 1 (unaligned rows), 4 (one 4-byte code word in, one 16-byte fp32 tri vector out) or
// 16 (int8-only observations, tri_out == NULL: 16-byte code vector in/out, 16-byte int8 tri vector out).  Every
// request is coalesced.  Either tri_out (fp32 observation rows) or tri_i8 (compact rows) may be NULL, not both.
template <int VPL>
__global__ __launch_bounds__(kGridThreads) void k_grid_update_coded(
    uint32_t *hit_mask, uint32_t *path_mask /*(not restrict: cleared in passing when `clean`)*/, int clean, int32_t *ray_count,
    const uint32_t *__restrict__ gt_bits,
    const uint8_t *__restrict__ reset_mask, int n, int g3, int words, int words_gt, uint8_t *__restrict__ prob_code,
    const float *__restrict__ tri_lut, uint32_t *__restrict__ scanned_bits, float *__restrict__ tri_out, int64_t tri_stride,
    int8_t *__restrict__ tri_i8, int64_t tri_i8_stride, int32_t *__restrict__ coverage, int32_t *__restrict__ overflow)
{
    __shared__ float lut[256];
    __shared__ int s_cov[kGridThreads / kWave];
    __shared__ unsigned long long s_cls[3][4];  // codes with tri-class +1 / 0 / -1, 64 codes per entry
    static_assert(kGridThreads == 256, "one thread per code below");
    {
        const float t = tri_lut[threadIdx.x];
        lut[threadIdx.x] = t;
        const unsigned long long b1 = __ballot(t == 1.0f), b0 = __ballot(t == 0.0f), bm = __ballot(t == -1.0f);
        if ((threadIdx.x & (kWave - 1)) == 0) {
            s_cls[0][threadIdx.x / kWave] = b1; s_cls[1][threadIdx.x / kWave] = b0; s_cls[2][threadIdx.x / kWave] = bm;
        }
    }
    __syncthreads();
    // Byte-parallel tri-class (VPL = 16): the table of gnbv_prob_code_tables is, per base, +1 for the first K1 step counts, 0 up to K2,
    // -1 from there (a probability that only falls), so tri(code) = (k < K1[base]) + (k < K2[base]) - 1 and four codes of a dword are
    // classified with ~10 integer instructions instead of four table reads, conversions and merges (the update is as much VALU- as
    // HBM-bound: ~14 instructions per voxel before, profiles/r05_notes.md).  Any other table takes the table path.
    int K1[2], K2[2];
#ifdef GRID_NO_SWAR
    bool swar = false;
#else
    bool swar = true;
#endif
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const unsigned long long p0 = s_cls[0][2 * b], p1 = s_cls[0][2 * b + 1], z0 = s_cls[1][2 * b], z1 = s_cls[1][2 * b + 1];
        const unsigned long long m0 = s_cls[2][2 * b], m1 = s_cls[2][2 * b + 1];
        K1[b] = __popcll(p0) + __popcll(p1);
        K2[b] = K1[b] + __popcll(z0) + __popcll(z1);
        auto prefix = [](unsigned long long lo, unsigned long long hi, int k) {  // bits [0, k) of the 128
            const unsigned long long elo = k >= 64 ? ~0ull : ((1ull << k) - 1ull), ehi = k <= 64 ? 0ull : (k >= 128 ? ~0ull : ((1ull << (k - 64)) - 1ull));
            return lo == elo && hi == ehi;
        };
        swar = swar && prefix(p0, p1, K1[b]) && prefix(p0 | z0, p1 | z1, K2[b]) && (p0 | z0 | m0) == ~0ull && (p1 | z1 | m1) == ~0ull;
    }
    const int e = blockIdx.y;
    const bool reset = reset_mask != nullptr && reset_mask[e] != 0;
    uint32_t *hm = hit_mask + (size_t)e * words, *pm = path_mask + (size_t)e * words;
    // `clean`: the masks and the ray count are left ZERO for the next call (the word's owner lane clears it after every lane
    // that shares the word -- neighbours in the same wave, same instruction -- has loaded it): saves the 16 MB fill launch
    if (clean && ray_count != nullptr && blockIdx.x == 0 && threadIdx.x == 0) ray_count[e] = 0;
    const uint32_t *gb = gt_bits + (size_t)e * words_gt;
    uint32_t *sb = scanned_bits + (size_t)e * words_gt;
    uint8_t *code = prob_code + (size_t)e * g3;
    float *tri = tri_out ? tri_out + (size_t)e * tri_stride : nullptr;
    int8_t *tri8 = tri_i8 ? tri_i8 + (size_t)e * tri_i8_stride : nullptr;  // compact rows for the conv1 kernels
    int cov = 0, ovf = 0;
    if constexpr (VPL > 1) {
        constexpr int NW = VPL / 4;  // 4-voxel words per lane
        constexpr uint32_t kBits = (1u << VPL) - 1u;
        const int nv = g3 / VPL;
        for (int i = blockIdx.x * kGridThreads + threadIdx.x; i < nv; i += gridDim.x * kGridThreads) {
            const int v0 = i * VPL, wd = v0 >> 5, sh = v0 & 31;
            const uint32_t hw = hm[wd], pw = pm[wd];
            const uint32_t hb = (hw >> sh) & kBits, pb = (pw >> sh) & kBits;
            uint32_t cw[NW], out[NW], t8[NW];
            if constexpr (NW == 4) {
                const uint4 c4 = reset ? make_uint4(0u, 0u, 0u, 0u) : reinterpret_cast<const uint4 *>(code)[i];
                cw[0] = c4.x; cw[1] = c4.y; cw[2] = c4.z; cw[3] = c4.w;
            } else {
                cw[0] = reset ? 0u : reinterpret_cast<const uint32_t *>(code)[i];
            }
            if (NW == 4 && swar) {
                constexpr uint32_t k01 = 0x01010101u, k80 = 0x80808080u, k7f = 0x7f7f7f7fu;
                const uint32_t a1 = (uint32_t)K1[0] * k01, b1 = (uint32_t)K1[1] * k01, a2 = (uint32_t)K2[0] * k01, b2 = (uint32_t)K2[1] * k01;
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    // step_code on four bytes: path -> k + 1 (saturating at 127), hit -> 0x80.  (x * 0x00204081) & 0x01010101 spreads four
                    // bits to the low bits of four bytes; no byte carries into its neighbour: k + 1 <= 0x80, 0x80 | k - K >= 0.
                    const uint32_t inc = (((pb >> (4 * q)) & 15u) * 0x00204081u) & k01;
                    const uint32_t hb1 = (((hb >> (4 * q)) & 15u) * 0x00204081u) & k01;
                    const uint32_t hmask = (hb1 << 8) - hb1;  // 0xff per hit byte
                    uint32_t k = (cw[q] & k7f) + inc;
                    const uint32_t sat = k & k80;
                    ovf |= sat != 0u;
                    k -= sat >> 7;
                    const uint32_t o = ((cw[q] & k80) | k) & ~hmask | (k80 & hmask);
                    out[q] = o;
                    // tri-class bytes: 1 where k < K1[base], 0 where k < K2[base], 0xff otherwise
                    const uint32_t bs = o & k80, bmask = (bs - (bs >> 7)) | bs;  // 0xff per byte with base 1
                    const uint32_t t1 = (a1 & ~bmask) | (b1 & bmask), t2 = (a2 & ~bmask) | (b2 & bmask);
                    const uint32_t kk = (o & k7f) | k80;
                    const uint32_t ge1 = ((kk - t1) >> 7) & k01, g2 = (kk - t2) & k80;
                    t8[q] = (ge1 ^ k01) | ((g2 - (g2 >> 7)) | g2);
                }
            } else
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                float4 t4;
                float *tp = &t4.x;
                out[q] = 0;
                t8[q] = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t c = step_code((cw[q] >> (8 * b)) & 255u, (hb >> (4 * q + b)) & 1u, (pb >> (4 * q + b)) & 1u, ovf);
                    out[q] |= c << (8 * b);
                    tp[b] = lut[c];
                    t8[q] |= ((uint32_t)(int)tp[b] & 255u) << (8 * b);
                }
                if constexpr (NW == 1)
                    if (tri) reinterpret_cast<float4 *>(tri)[i] = t4;
            }
            // (a step touches ~3 % of the voxels: the codes of a lane change only where a hit / path bit is set or the env resets)
            // (the stores below are conditional: -2.9 us per update at 256 x 64^3, same-box A/B, profiles/r05_notes.md)
            const bool touched = reset || (hb | pb) != 0u;
            if constexpr (NW == 4) {
                if (touched) reinterpret_cast<uint4 *>(code)[i] = make_uint4(out[0], out[1], out[2], out[3]);
                // (non-temporal here: +5 us per update back to back, round 4; +1.6 us INSIDE the rollout, round 5 -- the policy's conv kernel
                // reads this row 20 us later: profiles/r05_notes.md)
                reinterpret_cast<uint4 *>(tri8)[i] = make_uint4(t8[0], t8[1], t8[2], t8[3]);
            } else {
                if (touched) reinterpret_cast<uint32_t *>(code)[i] = out[0];
                if (tri8) reinterpret_cast<uint32_t *>(tri8)[i] = t8[0];
            }
            if (sh == 0) {  // one lane in 32 / VPL owns the 32-voxel word of the scanned set
                const uint32_t s0 = sb[wd];
                const uint32_t sw = (reset ? 0u : s0) | (hw & gb[wd]);
                if (sw != s0) sb[wd] = sw;
                cov += __popc(sw);
                if (clean) {
                    if (hw) hm[wd] = 0u;
                    if (pw) pm[wd] = 0u;
                }
            }
        }
    } else {
        for (int v = blockIdx.x * kGridThreads + threadIdx.x; v < g3; v += gridDim.x * kGridThreads) {
            const int wd = v >> 5;
            const uint32_t hw = hm[wd];
            const bool hb = (hw >> (v & 31)) & 1u, pb = (pm[wd] >> (v & 31)) & 1u;
            const uint32_t c = step_code(reset ? 0u : code[v], hb, pb, ovf);
            code[v] = (uint8_t)c;
            if (tri) tri[v] = lut[c];
            if (tri8) tri8[v] = (int8_t)(int)lut[c];
            if ((v & 31) == 0) {
                const uint32_t sw = (reset ? 0u : sb[wd]) | (hw & gb[wd]);
                sb[wd] = sw;
                cov += __popc(sw);
                if (clean) { hm[wd] = 0u; pm[wd] = 0u; }
            }
        }
    }
    if (ovf && overflow != nullptr) *overflow = 1;
    cov = wave_reduce_sum(cov);
    if ((threadIdx.x & (kWave - 1)) == 0) s_cov[threadIdx.x / kWave] = cov;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int i = 0; i < kGridThreads / kWave; ++i) t += s_cov[i];
        if (t) atomicAdd(&coverage[e], t);
    }
}

// (Round 4 tried the ray walk and this update as ONE persistent launch -- workgroups pulling walk tasks and, behind per-env completion
// counters, update parts: 220 us with a dynamic scheduler and agent-scope release / acquire fences, 196 us without the fences, 115 us
// with a static schedule, against 27 + 41 us for the two launches.  Returning device-scope atomics cost 2-5 us each under the launch's
// own memory traffic, and 4 workgroups per CU (the 32 KiB path mask in LDS) are half the waves the HBM-bound update needs to hide its
// latency.  Even the refactoring it needed -- this kernel's arguments as one struct, its body as a function over an item range with a
// run-time stride -- cost 5.6 us per update (104.7 -> 110.4, same-box A/B) and was reverted with it.  profiles/r04_notes.md)

__global__ void k_decode_prob(const uint8_t *__restrict__ code, int64_t count, const float *__restrict__ prob_lut, float *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) out[i] = prob_lut[code[i]];
}

// f32 grid [N, G^3] -> bitmask [N, words] (bit = value != 0); *not_binary is set when a value is
// neither 0 nor 1.  One wave-ballot per 64 voxels.
__global__ void k_pack_bits(const float *__restrict__ grid, int n, int g3, int words, uint32_t *__restrict__ bits, int *__restrict__ not_binary)
{
    const int e = blockIdx.y;
    const float *src = grid + (size_t)e * g3;
    uint32_t *dst = bits + (size_t)e * words;
    const int lane = threadIdx.x & 63;
    const int g3p = (g3 + 63) & ~63;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < g3p; v += gridDim.x * blockDim.x) {
        const float x = v < g3 ? src[v] : 0.0f;
        if (x != 0.0f && x != 1.0f && not_binary) *not_binary = 1;
        const unsigned long long m = __ballot(x != 0.0f);
        if (lane == 0) dst[v >> 5] = (uint32_t)m;
        if (lane == 32 && (v >> 5) < words) dst[v >> 5] = (uint32_t)(m >> 32);
    }
}

__global__ void k_unpack_bits_f32(const uint32_t *__restrict__ bits, int n, int64_t g3, int words, float *__restrict__ out)
{
    const int64_t total = (int64_t)n * g3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i / g3);
        const int v = (int)(i - (int64_t)e * g3);
        out[i] = ((bits[(size_t)e * words + (v >> 5)] >> (v & 31)) & 1u) ? 1.0f : 0.0f;
    }
}

// ===========================================================================
// standalone operators (function-level drop-ins, parity at every stage)
// ===========================================================================
__global__ void k_post_process_depth(const float *__restrict__ d, const float *__restrict__ s, int64_t count, float sense,
                                     float *__restrict__ dout, float *__restrict__ sout)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        dout[i] = process_depth(d[i], sense);
        sout[i] = nan_to_num_neginf0(s[i]);
    }
}

__global__ void k_rgb_to_gray(const uint8_t *__restrict__ rgba, int n, int h, int w, int oh, int ow, float *__restrict__ gray,
                              int64_t row_stride)
{
    const int total = n * oh * ow;
    const float sh = (float)h / (float)oh, sw = (float)w / (float)ow;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int e = i / (oh * ow), r = i - e * oh * ow, y = r / ow, x = r - y * ow;
        int sy = (int)floorf(__fmul_rn((float)y, sh)), sx = (int)floorf(__fmul_rn((float)x, sw));
        sy = min(sy, h - 1);
        sx = min(sx, w - 1);
        const uint8_t *p = rgba + (((size_t)e * h + sy) * w + sx) * 4;
        float v = __fmul_rn(0.2989f, (float)p[0]);
        v = __fadd_rn(v, __fmul_rn(0.587f, (float)p[1]));
        v = __fadd_rn(v, __fmul_rn(0.114f, (float)p[2]));
        gray[(size_t)e * row_stride + r] = (float)(uint8_t)v;
    }
}

__global__ void k_back_projection(const float *__restrict__ depth, const float *__restrict__ seg, const float *__restrict__ c2w,
                                  Intrinsics K, int n, int h, int w, float *__restrict__ world, uint8_t *__restrict__ fg)
{
    const int hw = h * w;
    const int64_t total = (int64_t)n * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i / hw), p = (int)(i - (int64_t)e * hw), y = p / w, x = p - y * w;
        const bool is_fg = seg[i] > 50.0f;
        float M[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) M[k] = c2w[(size_t)e * 16 + k];
        float wp[3];
        pixel_to_world(is_fg ? depth[i] : 0.0f, (float)x, (float)y, K, M, wp);
        world[i * 3 + 0] = wp[0];
        world[i * 3 + 1] = wp[1];
        world[i * 3 + 2] = wp[2];
        fg[i] = is_fg ? 1 : 0;
    }
}

__global__ void k_points_to_idx(const float *__restrict__ world, const uint8_t *__restrict__ fg, const float *__restrict__ range_gt,
                                const float *__restrict__ voxel_size, int n, int64_t hw, int g, int32_t *__restrict__ idx)
{
    const int64_t total = (int64_t)n * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i / hw);
        const VoxelFrame f = load_frame(range_gt + e * 6, voxel_size + e * 3);
        const float p[3] = {world[i * 3], world[i * 3 + 1], world[i * 3 + 2]};
        int ix[3];
        const int lin = fg[i] ? point_to_voxel(p, f, g, ix) : -1;
        idx[i * 3 + 0] = lin >= 0 ? ix[0] : -1;
        idx[i * 3 + 1] = lin >= 0 ? ix[1] : -1;
        idx[i * 3 + 2] = lin >= 0 ? ix[2] : -1;
    }
}

__global__ void k_pose_to_idx(const float *__restrict__ poses, const float *__restrict__ range_gt, const float *__restrict__ voxel_size,
                              int n, int64_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 3) return;
    const int e = i / 3, a = i - e * 3;
    const float v = voxel_size[e * 3 + a];
    const float vmin = __fsub_rn(range_gt[e * 6 + 2 * a + 1], __fmul_rn(0.5f, v));
    out[i] = (long long)floorf(__fdiv_rn(__fsub_rn(poses[i], vmin), v));
}

// one lane per ray, same launch geometry as the reference (block 256)
__global__ __launch_bounds__(256) void k_bresenham3d(const int32_t *__restrict__ src, const int32_t *__restrict__ tgt, int num_rays,
                                                     int g, int32_t *__restrict__ traj, int32_t *__restrict__ lens)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= num_rays) return;
    const int max_pts = 3 * g;
    int32_t *out = traj + (size_t)r * max_pts * 3;
    int k = 0;
    lens[r] = bresenham_walk(src[0], src[1], src[2], tgt[r * 3], tgt[r * 3 + 1], tgt[r * 3 + 2], g, max_pts,
                             [&](int x, int y, int z) { out[k * 3] = x; out[k * 3 + 1] = y; out[k * 3 + 2] = z; ++k; });
}

__global__ void k_tri_cls(const float *__restrict__ p, int64_t count, float t_occ, float t_free, float *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = p[i];
        out[i] = (v > t_occ ? 1.0f : 0.0f) - (v < t_free ? 1.0f : 0.0f);
    }
}

__global__ void k_unpack(const uint32_t *__restrict__ mask, int64_t bits_per_env, int words, int n, uint8_t *__restrict__ out)
{
    const int64_t total = (int64_t)n * bits_per_env;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i / bits_per_env);
        const int v = (int)(i - (int64_t)e * bits_per_env);
        out[i] = (mask[(size_t)e * words + (v >> 5)] >> (v & 31)) & 1u;
    }
}

// ===========================================================================
// C-ABI
// ===========================================================================
static inline int grid_for(int64_t count, int block, int cap = 256 * 8)
{
    int64_t b = (count + block - 1) / block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

static inline int mask_words(int g) { return (int)(((int64_t)g * g * g + 31) / 32); }
// words rounded so that every env's mask starts 256-byte aligned
static inline int mask_words_padded(int g) { return (mask_words(g) + 63) & ~63; }

struct VoxelWorkspace {
    uint32_t *hit, *path;
    int words;
    int32_t *ray_count;  // [n] (padded to 64 ints), directly behind the masks: one fill zeroes masks + counts
    int32_t *ray_list;   // [n][ray_cap] target voxels, or NULL (mask-only workspace: the two-launch mask kernels run)
    int64_t ray_cap;
};

GNBV_API size_t gnbv_voxel_workspace_bytes_hw(int n, int g, int h, int w);
static inline size_t ray_count_ints(int n) { return ((size_t)n + 63) & ~(size_t)63; }
// an env lists one ray per distinct voxel and chunk: never more than its pixels
static inline int64_t ray_list_cap(int g, int h, int w) { return ((int64_t)h * w + 63) & ~(int64_t)63; }

static inline VoxelWorkspace carve(void *ws, size_t ws_bytes, int n, int g, int h, int w)
{
    VoxelWorkspace v;
    v.words = mask_words_padded(g);
    v.hit = (uint32_t *)ws;
    v.path = v.hit + (size_t)n * v.words;
    v.ray_count = nullptr; v.ray_list = nullptr; v.ray_cap = 0;
    if (h > 0 && w > 0 && ws_bytes >= gnbv_voxel_workspace_bytes_hw(n, g, h, w)) {
        v.ray_count = (int32_t *)(v.path + (size_t)n * v.words);
        v.ray_list = v.ray_count + ray_count_ints(n);
        v.ray_cap = ray_list_cap(g, h, w);
    }
    return v;
}

GNBV_API size_t gnbv_voxel_workspace_bytes(int n, int g)
{
    if (n <= 0 || g <= 0) return 0;
    return (size_t)2 * n * mask_words_padded(g) * sizeof(uint32_t);
}

GNBV_API size_t gnbv_voxel_workspace_bytes_hw(int n, int g, int h, int w)
{
    if (n <= 0 || g <= 0 || h <= 0 || w <= 0) return 0;
    return gnbv_voxel_workspace_bytes(n, g) + (ray_count_ints(n) + (size_t)n * (size_t)ray_list_cap(g, h, w)) * sizeof(int32_t);
}

// inv_intri is a HOST pointer to 9 floats: the matrix is a constant of the task
// (env_train_gennbv.py:168-169) and travels to the kernels by value.
static inline int fetch_intrinsics(const float *p, hipStream_t, Intrinsics *K)
{
    for (int i = 0; i < 9; ++i) K->k[i] = p[i];
    return 0;
}

GNBV_API int gnbv_abi_version(void) { return GNBV_ABI_VERSION; }
GNBV_API const char *gnbv_build_arch(void) { return "gfx950"; }

GNBV_API int gnbv_post_process_depth(const float *depth_raw, const float *seg_raw, int64_t count, float depth_sense_dist,
                                     float *depth_out, float *seg_out, void *stream)
{
    GNBV_CHECK_ARG(depth_raw && seg_raw && depth_out && seg_out && count >= 0);
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_post_process_depth, dim3(grid_for(count, 256)), dim3(256), 0, gnbv_stream(stream), depth_raw,
                       seg_raw, count, depth_sense_dist, depth_out, seg_out);
    return gnbv_launch_status();
}

GNBV_API int gnbv_rgb_to_gray(const uint8_t *rgba, int n, int h, int w, int oh, int ow, float *gray, int64_t gray_row_stride,
                              void *stream)
{
    GNBV_CHECK_ARG(rgba && gray && n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && gray_row_stride >= (int64_t)oh * ow);
    hipLaunchKernelGGL(k_rgb_to_gray, dim3(grid_for((int64_t)n * oh * ow, 256)), dim3(256), 0, gnbv_stream(stream), rgba, n,
                       h, w, oh, ow, gray, gray_row_stride);
    return gnbv_launch_status();
}

GNBV_API int gnbv_back_projection(const float *depth, const float *seg, const float *c2w, const float *inv_intri, int n, int h,
                                  int w, float *world, uint8_t *fg, void *stream)
{
    GNBV_CHECK_ARG(depth && seg && c2w && inv_intri && world && fg && n > 0 && h > 0 && w > 0);
    Intrinsics K;
    int err = fetch_intrinsics(inv_intri, gnbv_stream(stream), &K);
    if (err) return err;
    hipLaunchKernelGGL(k_back_projection, dim3(grid_for((int64_t)n * h * w, 256)), dim3(256), 0, gnbv_stream(stream), depth,
                       seg, c2w, K, n, h, w, world, fg);
    return gnbv_launch_status();
}

GNBV_API int gnbv_points_to_idx(const float *world, const uint8_t *fg, const float *range_gt, const float *voxel_size, int n,
                                int64_t hw, int g, int32_t *idx, void *stream)
{
    GNBV_CHECK_ARG(world && fg && range_gt && voxel_size && idx && n > 0 && hw > 0 && g > 0);
    hipLaunchKernelGGL(k_points_to_idx, dim3(grid_for(n * hw, 256)), dim3(256), 0, gnbv_stream(stream), world, fg, range_gt,
                       voxel_size, n, hw, g, idx);
    return gnbv_launch_status();
}

GNBV_API int gnbv_pose_to_idx(const float *poses_xyz, const float *range_gt, const float *voxel_size, int n, int64_t *pose_idx,
                              void *stream)
{
    GNBV_CHECK_ARG(poses_xyz && range_gt && voxel_size && pose_idx && n > 0);
    hipLaunchKernelGGL(k_pose_to_idx, dim3((n * 3 + 255) / 256), dim3(256), 0, gnbv_stream(stream), poses_xyz, range_gt,
                       voxel_size, n, pose_idx);
    return gnbv_launch_status();
}

GNBV_API int gnbv_bresenham3d(const int32_t *source_pts, const int32_t *target_pts, int num_rays, int map_size,
                              int32_t *trajectory_pts, int32_t *trajectory_lengths, void *stream)
{
    GNBV_CHECK_ARG(num_rays >= 0 && map_size > 0);
    if (num_rays == 0) return 0;  // empty target list: nothing to write (pointers may be NULL)
    GNBV_CHECK_ARG(source_pts && trajectory_lengths && target_pts && trajectory_pts);
    hipLaunchKernelGGL(k_bresenham3d, dim3((num_rays + 255) / 256), dim3(256), 0, gnbv_stream(stream), source_pts, target_pts,
                       num_rays, map_size, trajectory_pts, trajectory_lengths);
    return gnbv_launch_status();
}

GNBV_API int gnbv_grid_tri_cls(const float *grid_prob, int64_t count, float threshold_occu, float threshold_free,
                               float *grid_tri_cls, void *stream)
{
    GNBV_CHECK_ARG(grid_prob && grid_tri_cls && count >= 0);
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_tri_cls, dim3(grid_for(count, 256)), dim3(256), 0, gnbv_stream(stream), grid_prob, count,
                       threshold_occu, threshold_free, grid_tri_cls);
    return gnbv_launch_status();
}


// GENNBV_VOXEL_LARGE: 1 = the large-grid kernels (k_hit_atomic + k_ray_slab) at every grid size, 0 = never (the round-1 kernels above
// G = 104), unset = by size.  Read per call (tests switch it between calls).
static int voxel_large_mode()
{
    const char *v = getenv("GENNBV_VOXEL_LARGE");
    return (v && (v[0] == '0' || v[0] == '1')) ? v[0] - '0' : -1;
}

// launches 1 + 2 (and the memsets): hit mask and path mask of every env into the workspace
static int launch_masks(const float *depth_raw, const float *seg_raw, const float *c2w, const float *inv_intri,
                        const float *poses_xyz, int64_t poses_row_stride, const float *range_gt, const float *voxel_size, int n,
                        int h, int w, int g, float depth_sense_dist, int32_t *coverage_count, const VoxelWorkspace &ws,
                        hipStream_t st, bool masks_are_zero = false, bool *used_lists = nullptr)
{
    if (used_lists) *used_lists = false;
    Intrinsics K;
    int err = fetch_intrinsics(inv_intri, st, &K);
    if (err) return err;
    const int words = ws.words;
    const size_t mask_bytes = (size_t)words * sizeof(uint32_t);
    // LDS staging of the masks: a gfx950 workgroup may use all 160 KiB.  Larger masks (G > 104) are split into windows
    // of <= 128 KiB, one workgroup per window.
    const size_t kLdsMax = 160 * 1024, ray_fixed = (kQueueCap + 32) * sizeof(uint32_t);
    const int hit_windows = mask_bytes <= kLdsMax ? 1 : (int)((mask_bytes + 128 * 1024 - 1) / (128 * 1024));
    const int path_windows = mask_bytes + ray_fixed <= kLdsMax ? 1 : (int)((mask_bytes + 128 * 1024 - 1) / (128 * 1024));
    const int hit_nw = (words + hit_windows - 1) / hit_windows, path_nw = (words + path_windows - 1) / path_windows;
    const size_t hit_lds = (size_t)hit_nw * sizeof(uint32_t), path_lds = ray_fixed + (size_t)path_nw * sizeof(uint32_t);
    // ray-cast workgroups per env: ~4 per CU in total, so that an env with many hit voxels
    // (measured 4x the mean) is spread over several CUs (profiles/r01_notes.md)
    int splits = (1024 + n - 1) / n;
    splits = splits < 1 ? 1 : (splits > 8 ? 8 : splits);
    // inv_intri of a pinhole camera is [[a,0,c],[0,b,d],[0,0,1]]: lets the kernels drop exact-zero terms
    const bool kfast = K.k[1] == 0.0f && K.k[3] == 0.0f && K.k[6] == 0.0f && K.k[7] == 0.0f && K.k[8] == 1.0f;
    const int env_groups = (n + 7) / 8;
    // hit mask + ray list, then the load-balanced ray cast over the lists (needs the h/w-sized workspace)
    const size_t list_lds = mask_bytes + 64 * sizeof(uint32_t) + (size_t)((words + 1) & ~1) * sizeof(uint16_t) + kQueueCapPx * sizeof(int32_t);
    if (ws.ray_list != nullptr && list_lds <= kLdsMax && words <= 65536 && voxel_large_mode() != 1) {
        // workgroups per env: two per CU in total (1024 threads each = 32 waves per CU).  (The kernel is capped at 80 SGPRs:
        // with the 96 it wanted, the SIMD's 800-entry SGPR file held 7 waves and a second 16-wave workgroup never became
        // resident beside the first -- 512 workgroups ran as two rounds, profiles/r02_notes.md.)
        int fchunks = (512 + n - 1) / n;
        fchunks = fchunks < 1 ? 1 : (fchunks > 16 ? 16 : fchunks);
        if (used_lists) *used_lists = true;
        // one fill: [hit (only when OR-accumulated) | path | ray counts] are adjacent for the full env range
        if (masks_are_zero) {
            // (the previous call's grid-update launch left masks and counts zero: GNBV_VOXEL_WS_CLEAN)
        } else if (ws.path == ws.hit + (size_t)n * words && (void *)ws.ray_count == (void *)(ws.path + (size_t)n * words)) {
            uint32_t *z0 = fchunks > 1 ? ws.hit : ws.path;
            const size_t zb = (size_t)((char *)(ws.ray_count + ray_count_ints(n)) - (char *)z0);
            if ((err = (int)hipMemsetAsync(z0, 0, zb, st))) return err;
        } else {
            if (fchunks > 1 && (err = (int)hipMemsetAsync(ws.hit, 0, (size_t)n * mask_bytes, st))) return err;
            if ((err = (int)hipMemsetAsync(ws.path, 0, (size_t)n * mask_bytes, st))) return err;
            if ((err = (int)hipMemsetAsync(ws.ray_count, 0, (size_t)n * sizeof(int32_t), st))) return err;
        }
        const int fgrid = env_groups * 8 * fchunks;
#define GNBV_HITLIST(KF)                                                                                                             \
    do {                                                                                                                             \
        if (list_lds > 64 * 1024 &&                                                                                                  \
            hipFuncSetAttribute((const void *)k_hit_list<KF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)list_lds) != hipSuccess) \
            return (int)hipGetLastError();                                                                                           \
        hipLaunchKernelGGL((k_hit_list<KF>), dim3(fgrid), dim3(kFusedThreads), list_lds, st, depth_raw, seg_raw, c2w, K, range_gt,   \
                           voxel_size, n, h, w, g, depth_sense_dist, fchunks, words, ws.hit, ws.ray_count, ws.ray_list, ws.ray_cap,   \
                           coverage_count);                                                                                          \
    } while (0)
        if (kfast) GNBV_HITLIST(true);
        else GNBV_HITLIST(false);
#undef GNBV_HITLIST
        if ((err = gnbv_launch_status())) return err;
        const size_t ray_lds = mask_bytes;
        if (ray_lds > 64 * 1024 &&
            hipFuncSetAttribute((const void *)k_ray_list, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ray_lds) != hipSuccess)
            return (int)hipGetLastError();
        // grid: kListGridSlices workgroups per env on average (an env takes as many as it has slices of kListThreads rays; 256 x 240x320 x
        // 64^3: ~3.5); past that the workgroups take several items
        hipLaunchKernelGGL(k_ray_list, dim3(env_groups * 8 * kListGridSlices), dim3(kListThreads), ray_lds, st, ws.ray_count, ws.ray_list,
                           ws.ray_cap, poses_xyz, poses_row_stride, range_gt, voxel_size, n, g, words, ws.path);
        return gnbv_launch_status();
    }
    // large grids (the mask does not fit a workgroup's LDS next to its lists: G > 104) with the list workspace: k_hit_atomic +
    // k_ray_slab.  (GENNBV_VOXEL_LARGE=1 takes this path at every grid size: tests.)
    // (GENNBV_VOXEL_LARGE=0: the round-1 kernels below; GENNBV_VOXEL_SLAB_PLANES=k: planes per slab, rounded up to the alignment -- tests.)
    if (ws.ray_list != nullptr && voxel_large_mode() != 0) {
        // slabs of whole x-planes whose first bit is word-aligned, <= 32 KiB of mask each where the grid allows it
        const int64_t gg = (int64_t)g * g;
        int align = 32;
        while (align > 1 && (gg * (align / 2)) % 32 == 0) align /= 2;
        int64_t per = (8192 * 32) / (gg * align);
        int slab_planes = (int)(align * (per < 1 ? 1 : per));
        if (const char *sp = getenv("GENNBV_VOXEL_SLAB_PLANES")) {
            const int k = atoi(sp);
            if (k > 0) slab_planes = ((k + align - 1) / align) * align;
        }
        slab_planes = slab_planes > g ? ((g + align - 1) / align) * align : slab_planes;
        const int slabs = (g + slab_planes - 1) / slab_planes;
        const size_t slab_lds = (size_t)((slab_planes * gg + 31) / 32) * sizeof(uint32_t);
        if (slab_lds <= kLdsMax) {
            if (used_lists) *used_lists = true;
            if (masks_are_zero) {
            } else if (ws.path == ws.hit + (size_t)n * words && (void *)ws.ray_count == (void *)(ws.path + (size_t)n * words)) {
                const size_t zb = (size_t)((char *)(ws.ray_count + ray_count_ints(n)) - (char *)ws.hit);
                if ((err = (int)hipMemsetAsync(ws.hit, 0, zb, st))) return err;
            } else {
                if ((err = (int)hipMemsetAsync(ws.hit, 0, (size_t)n * mask_bytes, st))) return err;
                if ((err = (int)hipMemsetAsync(ws.path, 0, (size_t)n * mask_bytes, st))) return err;
                if ((err = (int)hipMemsetAsync(ws.ray_count, 0, (size_t)n * sizeof(int32_t), st))) return err;
            }
            int fchunks = (512 + n - 1) / n;  // two 1024-thread workgroups per CU in total
            fchunks = fchunks < 1 ? 1 : (fchunks > 16 ? 16 : fchunks);
            const int fgrid = env_groups * 8 * fchunks;
            if (kfast)
                hipLaunchKernelGGL((k_hit_atomic<true>), dim3(fgrid), dim3(kFusedThreads), 0, st, depth_raw, seg_raw, c2w, K, range_gt, voxel_size, n, h, w, g,
                                   depth_sense_dist, fchunks, words, ws.hit, ws.ray_count, ws.ray_list, ws.ray_cap, coverage_count);
            else
                hipLaunchKernelGGL((k_hit_atomic<false>), dim3(fgrid), dim3(kFusedThreads), 0, st, depth_raw, seg_raw, c2w, K, range_gt, voxel_size, n, h, w, g,
                                   depth_sense_dist, fchunks, words, ws.hit, ws.ray_count, ws.ray_list, ws.ray_cap, coverage_count);
            if ((err = gnbv_launch_status())) return err;
            if (slab_lds > 64 * 1024 &&
                hipFuncSetAttribute((const void *)k_ray_slab, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slab_lds) != hipSuccess)
                return (int)hipGetLastError();
            int slab_slices = 32 / slabs;  // ~32 workgroups per env, at least two per slab
            slab_slices = slab_slices < 2 ? 2 : (slab_slices > kListSlices ? kListSlices : slab_slices);
            hipLaunchKernelGGL(k_ray_slab, dim3(env_groups * 8 * slabs * slab_slices), dim3(kSlabThreads), slab_lds, st, ws.ray_count, ws.ray_list, ws.ray_cap,
                               poses_xyz, poses_row_stride, range_gt, voxel_size, n, g, words, slabs, slab_planes, slab_slices, ws.path);
            return gnbv_launch_status();
        }
    }
    // zero the hit masks (OR-accumulated by atomics); the path masks only when they are
    // OR-accumulated too (several splits, or no LDS staging)
    const bool path_needs_zero = splits > 1;  // (windows write disjoint word ranges)
    err = (int)hipMemsetAsync(ws.hit, 0, (size_t)n * mask_bytes, st);
    if (err) return err;
    if (path_needs_zero) {
        err = (int)hipMemsetAsync(ws.path, 0, (size_t)n * mask_bytes, st);  // (hit / path of an env RANGE are not adjacent)
        if (err) return err;
    }
    err = (int)hipMemsetAsync(coverage_count, 0, (size_t)n * sizeof(int32_t), st);
    if (err) return err;
    // launch 1: hit mask.  chunks: enough workgroups to cover the chip several times.
    int chunks = (8 * 256 + n - 1) / n;
    chunks = chunks < 1 ? 1 : (chunks > 16 ? 16 : chunks);
    const int hit_grid = env_groups * 8 * chunks * hit_windows;
#define GNBV_HIT(KF, WIN)                                                                                                            \
    do {                                                                                                                             \
        if (hit_lds > 64 * 1024 &&                                                                                                   \
            hipFuncSetAttribute((const void *)k_hit_mask<KF, WIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hit_lds) != hipSuccess) \
            return (int)hipGetLastError();                                                                                           \
        hipLaunchKernelGGL((k_hit_mask<KF, WIN>), dim3(hit_grid), dim3(kHitThreads), hit_lds, st, depth_raw, seg_raw, c2w, K, range_gt, \
                           voxel_size, n, h, w, g, depth_sense_dist, chunks, words, ws.hit, hit_windows, hit_nw);                    \
    } while (0)
    if (hit_windows > 1) {
        if (kfast) GNBV_HIT(true, true);
        else GNBV_HIT(false, true);
    } else {
        if (kfast) GNBV_HIT(true, false);
        else GNBV_HIT(false, false);
    }
#undef GNBV_HIT
    err = gnbv_launch_status();
    if (err) return err;
    // launch 2: ray cast, N x splits (x windows) workgroups
    const int ray_grid = env_groups * 8 * splits * path_windows;
#define GNBV_RAY(WIN)                                                                                                                \
    do {                                                                                                                             \
        if (path_lds > 64 * 1024 &&                                                                                                  \
            hipFuncSetAttribute((const void *)k_raycast<true, WIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)path_lds) != hipSuccess) \
            return (int)hipGetLastError();                                                                                           \
        hipLaunchKernelGGL((k_raycast<true, WIN>), dim3(ray_grid), dim3(kRayThreads), path_lds, st, ws.hit, poses_xyz, poses_row_stride, \
                           range_gt, voxel_size, n, g, words, splits, ws.path, path_windows, path_nw);                               \
    } while (0)
    if (path_windows > 1) {
        GNBV_RAY(true);
    } else {
        GNBV_RAY(false);
    }
#undef GNBV_RAY
    return gnbv_launch_status();
}

static inline int grid_update_blocks(int64_t items, int n)
{
    int bx = (int)((items + kGridThreads - 1) / kGridThreads);
    const int cap = (4096 + n - 1) / n;  // ~16 workgroups per CU in total, grid-stride beyond
    return bx > cap ? cap : bx;
}

GNBV_API int gnbv_update_occ_grid(const float *depth_raw, const float *seg_raw, const float *c2w, const float *inv_intri,
                                  const float *poses_xyz, int64_t poses_row_stride, const float *range_gt,
                                  const float *voxel_size, const float *grid_gt, const uint8_t *reset_mask, int n, int h,
                                  int w, int g, float depth_sense_dist, float *prob_grid, float *scanned_gt_grid,
                                  float *tri_out, int64_t tri_row_stride, int32_t *coverage_count, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    GNBV_CHECK_ARG(depth_raw && seg_raw && c2w && inv_intri && poses_xyz && range_gt && voxel_size && grid_gt);
    GNBV_CHECK_ARG(prob_grid && scanned_gt_grid && tri_out && coverage_count && workspace);
    GNBV_CHECK_ARG(n > 0 && h > 0 && w > 0 && g > 1 && g <= 1024 && poses_row_stride >= 3);
    const int64_t g3 = (int64_t)g * g * g;
    GNBV_CHECK_ARG(g3 < (1ll << 31) && tri_row_stride >= g3 && (int64_t)h * w < (1ll << 31));
    GNBV_CHECK_ARG(workspace_bytes >= gnbv_voxel_workspace_bytes(n, g) && ((uintptr_t)workspace & 255) == 0);
    hipStream_t st = gnbv_stream(stream);
    VoxelWorkspace ws = carve(workspace, workspace_bytes, n, g, h, w);
    int err = launch_masks(depth_raw, seg_raw, c2w, inv_intri, poses_xyz, poses_row_stride, range_gt, voxel_size, n, h, w, g,
                           depth_sense_dist, coverage_count, ws, st);
    if (err) return err;
    // launch 3: grid update
    const bool vec4 = (g3 % 4 == 0) && (tri_row_stride % 4 == 0) && (((uintptr_t)tri_out & 15) == 0) &&
                      (((uintptr_t)prob_grid & 15) == 0) && (((uintptr_t)scanned_gt_grid & 15) == 0) &&
                      (((uintptr_t)grid_gt & 15) == 0);
    const int bx = grid_update_blocks(vec4 ? g3 / 4 : g3, n);
    if (vec4)
        hipLaunchKernelGGL(k_grid_update<true>, dim3(bx, n), dim3(kGridThreads), 0, st, ws.hit, ws.path, grid_gt, reset_mask,
                           n, (int)g3, ws.words, prob_grid, scanned_gt_grid, tri_out, tri_row_stride, coverage_count);
    else
        hipLaunchKernelGGL(k_grid_update<false>, dim3(bx, n), dim3(kGridThreads), 0, st, ws.hit, ws.path, grid_gt,
                           reset_mask, n, (int)g3, ws.words, prob_grid, scanned_gt_grid, tri_out, tri_row_stride,
                           coverage_count);
    return gnbv_launch_status();
}

GNBV_API int gnbv_grid_bit_words(int g) { return g > 0 ? mask_words_padded(g) : 0; }

GNBV_API int gnbv_pack_grid_bits(const float *grid, int n, int g, uint32_t *bits, int *not_binary, void *stream)
{
    GNBV_CHECK_ARG(grid && bits && n > 0 && g > 0);
    const int64_t g3 = (int64_t)g * g * g;
    const int words = mask_words_padded(g);
    hipStream_t st = gnbv_stream(stream);
    int err = (int)hipMemsetAsync(bits, 0, (size_t)n * words * sizeof(uint32_t), st);
    if (err) return err;
    if (not_binary && (err = (int)hipMemsetAsync(not_binary, 0, sizeof(int), st))) return err;
    int bx = (int)((g3 + 255) / 256);
    bx = bx > 1024 ? 1024 : bx;
    hipLaunchKernelGGL(k_pack_bits, dim3(bx, n), dim3(256), 0, st, grid, n, (int)g3, words, bits, not_binary);
    return gnbv_launch_status();
}

GNBV_API int gnbv_unpack_grid_bits(const uint32_t *bits, int n, int g, float *grid, void *stream)
{
    GNBV_CHECK_ARG(grid && bits && n > 0 && g > 0);
    const int64_t g3 = (int64_t)g * g * g;
    hipLaunchKernelGGL(k_unpack_bits_f32, dim3(grid_for(n * g3, 256)), dim3(256), 0, gnbv_stream(stream), bits, n, g3,
                       mask_words_padded(g), grid);
    return gnbv_launch_status();
}

// one chain (hit mask -> ray cast -> grid update) over the env range [e0, e0 + n) of the batch
static int update_packed_range(const float *depth_raw, const float *seg_raw, const float *c2w, const float *inv_intri,
                               const float *poses_xyz, int64_t poses_row_stride, const float *range_gt, const float *voxel_size,
                               const uint32_t *gt_bits, const uint8_t *reset_mask, int e0, int n, int h, int w, int g,
                               float depth_sense_dist, float *prob_grid, uint32_t *scanned_bits, float *tri_out,
                               int64_t tri_row_stride, int32_t *coverage_count, const VoxelWorkspace &full, hipStream_t st)
{
    const int64_t g3 = (int64_t)g * g * g, hw = (int64_t)h * w;
    VoxelWorkspace ws = full;
    ws.hit = full.hit + (size_t)e0 * full.words;
    ws.path = full.path + (size_t)e0 * full.words;
    if (full.ray_list) { ws.ray_count = full.ray_count + e0; ws.ray_list = full.ray_list + (size_t)e0 * full.ray_cap; }
    depth_raw += e0 * hw; seg_raw += e0 * hw; c2w += (size_t)e0 * 16; poses_xyz += (size_t)e0 * poses_row_stride;
    range_gt += (size_t)e0 * 6; voxel_size += (size_t)e0 * 3; gt_bits += (size_t)e0 * full.words;
    if (reset_mask) reset_mask += e0;
    prob_grid += (size_t)e0 * g3; scanned_bits += (size_t)e0 * full.words; tri_out += (size_t)e0 * tri_row_stride;
    coverage_count += e0;
    int err = launch_masks(depth_raw, seg_raw, c2w, inv_intri, poses_xyz, poses_row_stride, range_gt, voxel_size, n, h, w, g,
                           depth_sense_dist, coverage_count, ws, st);
    if (err) return err;
    const bool vec4 = (g3 % 4 == 0) && (tri_row_stride % 4 == 0) && (((uintptr_t)tri_out & 15) == 0) &&
                      (((uintptr_t)prob_grid & 15) == 0);
    const int bx = grid_update_blocks(vec4 ? g3 / 4 : g3, n);
    if (vec4)
        hipLaunchKernelGGL(k_grid_update_packed<true>, dim3(bx, n), dim3(kGridThreads), 0, st, ws.hit, ws.path, gt_bits,
                           reset_mask, n, (int)g3, ws.words, ws.words, prob_grid, scanned_bits, tri_out, tri_row_stride,
                           coverage_count);
    else
        hipLaunchKernelGGL(k_grid_update_packed<false>, dim3(bx, n), dim3(kGridThreads), 0, st, ws.hit, ws.path, gt_bits,
                           reset_mask, n, (int)g3, ws.words, ws.words, prob_grid, scanned_bits, tri_out, tri_row_stride,
                           coverage_count);
    return gnbv_launch_status();
}

GNBV_API int gnbv_update_occ_grid_packed(const float *depth_raw, const float *seg_raw, const float *c2w, const float *inv_intri,
                                         const float *poses_xyz, int64_t poses_row_stride, const float *range_gt,
                                         const float *voxel_size, const uint32_t *gt_bits, const uint8_t *reset_mask, int n,
                                         int h, int w, int g, float depth_sense_dist, float *prob_grid, uint32_t *scanned_bits,
                                         float *tri_out, int64_t tri_row_stride, int32_t *coverage_count, void *workspace,
                                         size_t workspace_bytes, void *stream)
{
    GNBV_CHECK_ARG(depth_raw && seg_raw && c2w && inv_intri && poses_xyz && range_gt && voxel_size && gt_bits);
    GNBV_CHECK_ARG(prob_grid && scanned_bits && tri_out && coverage_count && workspace);
    GNBV_CHECK_ARG(n > 0 && h > 0 && w > 0 && g > 1 && g <= 1024 && poses_row_stride >= 3);
    const int64_t g3 = (int64_t)g * g * g;
    GNBV_CHECK_ARG(g3 < (1ll << 31) && tri_row_stride >= g3 && (int64_t)h * w < (1ll << 31));
    GNBV_CHECK_ARG(workspace_bytes >= gnbv_voxel_workspace_bytes(n, g) && ((uintptr_t)workspace & 255) == 0);
    hipStream_t st = gnbv_stream(stream);
    VoxelWorkspace ws = carve(workspace, workspace_bytes, n, g, h, w);
    return update_packed_range(depth_raw, seg_raw, c2w, inv_intri, poses_xyz, poses_row_stride, range_gt, voxel_size, gt_bits, reset_mask, 0,
                               n, h, w, g, depth_sense_dist, prob_grid, scanned_bits, tri_out, tri_row_stride, coverage_count, ws, st);
}

// host helper: the two 256-entry tables of the coded probability grid, by the exact fp32 iteration of the reference
// update (x <- x - 0.05f from base 0 / 1): prob_lut[code] = value, tri_lut[code] = (value > 0.5) - (value < 0)
GNBV_API void gnbv_prob_code_tables(float *prob_lut, float *tri_lut)
{
    for (int base = 0; base < 2; ++base) {
        volatile float x = (float)base;
        for (int k = 0; k < 128; ++k) {
            const float v = x;
            if (prob_lut) prob_lut[base * 128 + k] = v;
            if (tri_lut) tri_lut[base * 128 + k] = (v > 0.5f ? 1.0f : 0.0f) - (v < 0.0f ? 1.0f : 0.0f);
            x = v - 0.05f;
        }
    }
}

GNBV_API int gnbv_decode_prob_grid(const uint8_t *prob_code, int64_t count, const float *prob_lut, float *prob_out, void *stream)
{
    GNBV_CHECK_ARG(prob_code && prob_lut && prob_out && count >= 0);
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_decode_prob, dim3(grid_for(count, 256)), dim3(256), 0, gnbv_stream(stream), prob_code, count, prob_lut, prob_out);
    return gnbv_launch_status();
}

GNBV_API int gnbv_update_occ_grid_coded(const float *depth_raw, const float *seg_raw, const float *c2w, const float *inv_intri,
                                        const float *poses_xyz, int64_t poses_row_stride, const float *range_gt,
                                        const float *voxel_size, const uint32_t *gt_bits, const uint8_t *reset_mask, int n,
                                        int h, int w, int g, float depth_sense_dist, uint8_t *prob_code, const float *tri_lut,
                                        uint32_t *scanned_bits, float *tri_out, int64_t tri_row_stride, int8_t *tri_i8,
                                        int64_t tri_i8_row_stride, int32_t *coverage_count, int32_t *overflow, void *workspace,
                                        size_t workspace_bytes, int workspace_flags, void *stream)
{
    GNBV_CHECK_ARG(depth_raw && seg_raw && c2w && inv_intri && poses_xyz && range_gt && voxel_size && gt_bits);
    GNBV_CHECK_ARG(prob_code && tri_lut && scanned_bits && (tri_out || tri_i8) && coverage_count && workspace);
    GNBV_CHECK_ARG(n > 0 && h > 0 && w > 0 && g > 1 && g <= 1024 && poses_row_stride >= 3);
    const int64_t g3 = (int64_t)g * g * g;
    GNBV_CHECK_ARG(g3 < (1ll << 31) && (tri_out == nullptr || tri_row_stride >= g3) && (int64_t)h * w < (1ll << 31));
    GNBV_CHECK_ARG(workspace_bytes >= gnbv_voxel_workspace_bytes(n, g) && ((uintptr_t)workspace & 255) == 0);
    hipStream_t st = gnbv_stream(stream);
    VoxelWorkspace ws = carve(workspace, workspace_bytes, n, g, h, w);
    const bool clean = (workspace_flags & GNBV_VOXEL_WS_CLEAN) != 0;
    bool lists = false;
    int err = launch_masks(depth_raw, seg_raw, c2w, inv_intri, poses_xyz, poses_row_stride, range_gt, voxel_size, n, h, w, g,
                           depth_sense_dist, coverage_count, ws, st, clean, &lists);
    if (err) return err;
    const int leave_clean = (clean && lists) ? 1 : 0;  // (the two-launch mask kernels zero what they need themselves)
    GNBV_CHECK_ARG(tri_i8 == nullptr || tri_i8_row_stride >= g3);
    const bool vec4 = (g3 % 4 == 0) && (tri_out == nullptr || ((tri_row_stride % 4 == 0) && (((uintptr_t)tri_out & 15) == 0))) &&
                      (((uintptr_t)prob_code & 3) == 0) && (tri_i8 == nullptr || ((((uintptr_t)tri_i8 | (uintptr_t)tri_i8_row_stride) & 3) == 0));
    const bool vec16 = tri_out == nullptr && (g3 % 16 == 0) && (((uintptr_t)prob_code & 15) == 0) &&
                       ((((uintptr_t)tri_i8 | (uintptr_t)tri_i8_row_stride) & 15) == 0);
    const int vpl = vec16 ? 16 : vec4 ? 4 : 1;
    const int bx = grid_update_blocks(g3 / vpl, n);
#define GNBV_LAUNCH_CODED(V)                                                                                                          \
    hipLaunchKernelGGL(k_grid_update_coded<V>, dim3(bx, n), dim3(kGridThreads), 0, st, ws.hit, ws.path, leave_clean, ws.ray_count, gt_bits, reset_mask, n, (int)g3, \
                       ws.words, ws.words, prob_code, tri_lut, scanned_bits, tri_out, tri_row_stride, tri_i8, tri_i8_row_stride,        \
                       coverage_count, overflow)
    if (vpl == 16) GNBV_LAUNCH_CODED(16);
    else if (vpl == 4) GNBV_LAUNCH_CODED(4);
    else GNBV_LAUNCH_CODED(1);
#undef GNBV_LAUNCH_CODED
    return gnbv_launch_status();
}

GNBV_API int gnbv_unpack_masks(const void *workspace, int n, int g, uint8_t *hit_u8, uint8_t *path_u8, void *stream)
{
    GNBV_CHECK_ARG(workspace && n > 0 && g > 0);
    VoxelWorkspace ws = carve(const_cast<void *>(workspace), 0, n, g, 0, 0);
    const int64_t g3 = (int64_t)g * g * g;
    if (hit_u8)
        hipLaunchKernelGGL(k_unpack, dim3(grid_for(n * g3, 256)), dim3(256), 0, gnbv_stream(stream), ws.hit, g3, ws.words, n,
                           hit_u8);
    if (path_u8)
        hipLaunchKernelGGL(k_unpack, dim3(grid_for(n * g3, 256)), dim3(256), 0, gnbv_stream(stream), ws.path, g3, ws.words, n,
                           path_u8);
    return gnbv_launch_status();
}
