/*
 * oracle.c -- CPU restatement of the GenNBV state-encoding + GAE hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg may load this library, and only as the
 * checker / the timed CPU baseline.  It is never on the product path: the
 * product (gennbv_amd) calls the HIP library through the C-ABI declared in
 * include/gennbv_hip.h and fails loudly when that library is missing.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference root, zjwzcx/GenNBV).  The restatement is scalar, single
 * threaded C with a *canonical fp32 operation order* (DESIGN.md "canonical
 * order"): dot products are k-ordered fmaf chains whose first product is
 * rounded on its own, divisions are IEEE, nothing is contracted
 * (-ffp-contract=off).  That order reproduces torch-CPU einsum/bmm bit for bit
 * on the fixtures generated from the reference (tests/golden, F2/F3).
 *
 * Pinned against: the reference's Python run in the build container
 * (oracle/gen_golden.py -> tests/golden/<name>.npz) and the reference's CUDA-C
 * ray-cast kernel text compiled as host C++ (oracle/_ref/libref_bresenham.so).
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* A1  post_process_camera_tensor, depth/seg branch                          */
/*     gennbv/env/env_train_base.py:521-534                                   */
/*     nan_to_num(neginf=0): NaN->0, +inf->FLT_MAX, -inf->0 ; clamp(min=-50); */
/*     abs.  seg: nan_to_num(neginf=0) only.                                   */
/* ------------------------------------------------------------------------ */
static inline float orc_nan_to_num_neginf0(float x)
{
    if (isnan(x)) return 0.0f;
    if (isinf(x)) return x > 0 ? FLT_MAX : 0.0f;
    return x;
}

ORC_API void orc_post_process_depth(const float *depth_raw, const float *seg_raw, int64_t n,
                                    float depth_sense_dist, float *depth_out, float *seg_out)
{
    for (int64_t i = 0; i < n; ++i) {
        float d = orc_nan_to_num_neginf0(depth_raw[i]);
        d = d < depth_sense_dist ? depth_sense_dist : d; /* clamp(min=-50) */
        depth_out[i] = fabsf(d);
        seg_out[i] = orc_nan_to_num_neginf0(seg_raw[i]);
    }
}

/* ------------------------------------------------------------------------ */
/* A1  rgb branch: gennbv/env/env_train_base.py:517-520                       */
/*     RGBA u8 [N,H,W,4] -> drop alpha -> nearest interpolate to 64x64 ->     */
/*     rgb_to_grayscale -> f32 [N,1,64,64].                                    */
/*     PARITY UNPINNED: torchvision is absent here.  Restates the published   */
/*     torchvision formula for integer tensors: (0.2989 R + 0.587 G +         */
/*     0.114 B) computed in fp32 then truncated to uint8, then .to(float32);   */
/*     nearest index = floor(dst * scale), scale = in/out (torch 'nearest').   */
/* ------------------------------------------------------------------------ */
ORC_API void orc_rgb_to_gray64(const uint8_t *rgba, int n, int h, int w, int oh, int ow, float *gray)
{
    const float sh = (float)h / (float)oh, sw = (float)w / (float)ow;
    for (int e = 0; e < n; ++e)
        for (int y = 0; y < oh; ++y) {
            int sy = (int)floorf((float)y * sh);
            if (sy > h - 1) sy = h - 1;
            for (int x = 0; x < ow; ++x) {
                int sx = (int)floorf((float)x * sw);
                if (sx > w - 1) sx = w - 1;
                const uint8_t *p = rgba + (((int64_t)e * h + sy) * w + sx) * 4;
                float v = 0.2989f * (float)p[0];
                v = v + 0.587f * (float)p[1];
                v = v + 0.114f * (float)p[2];
                gray[((int64_t)e * oh + y) * ow + x] = (float)(uint8_t)v;
            }
        }
}

/* ------------------------------------------------------------------------ */
/* A2  back_projection_fg: gennbv/env/env_train_gennbv.py:494-533             */
/*   fg = seg > 50 (:504); depth[~fg] = 0 (:509)                               */
/*   coords_pixel = d*(u, v, 1)                       (:519, no contraction)   */
/*   coords_cam   = inv_intri @ coords_pixel          (:522, 3-term chain)     */
/*   coords_world = c2w @ (coords_cam, 1)             (:526, 4-term chain)     */
/*   c2w is an INPUT here: inv(view^T) @ blender2opencv, translation minus     */
/*   env_origins (:512-514) is host-side plumbing (torch.linalg.inv).          */
/*   Output: world point for every pixel [N,HW,3] and the fg mask [N,HW].      */
/* ------------------------------------------------------------------------ */
static inline void orc_pixel_to_world(float d, float u, float v, const float *Ki /*3x3*/,
                                      const float *M /*4x4 row-major c2w*/, float *out3)
{
    const float pu = d * u, pv = d * v, pw = d * 1.0f;
    float cam[3];
    for (int i = 0; i < 3; ++i) {
        float acc = Ki[i * 3 + 0] * pu;
        acc = fmaf(Ki[i * 3 + 1], pv, acc);
        acc = fmaf(Ki[i * 3 + 2], pw, acc);
        cam[i] = acc;
    }
    for (int i = 0; i < 3; ++i) {
        float acc = M[i * 4 + 0] * cam[0];
        acc = fmaf(M[i * 4 + 1], cam[1], acc);
        acc = fmaf(M[i * 4 + 2], cam[2], acc);
        acc = fmaf(M[i * 4 + 3], 1.0f, acc);
        out3[i] = acc;
    }
}

ORC_API void orc_back_projection(const float *depth /*[N,H,W] processed*/, const float *seg /*[N,H,W]*/,
                                 const float *c2w /*[N,4,4]*/, const float *inv_intri /*[3,3]*/,
                                 int n, int h, int w, float *world /*[N,HW,3]*/, uint8_t *fg /*[N,HW]*/)
{
    const int64_t hw = (int64_t)h * w;
    for (int e = 0; e < n; ++e)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const int64_t p = (int64_t)e * hw + (int64_t)y * w + x;
                const int is_fg = seg[p] > 50.0f;
                const float d = is_fg ? depth[p] : 0.0f;
                fg[p] = (uint8_t)is_fg;
                orc_pixel_to_world(d, (float)x, (float)y, inv_intri, c2w + (int64_t)e * 16, world + p * 3);
            }
}

/* ------------------------------------------------------------------------ */
/* A3/A4  voxel index math: gennbv/utils.py:230-270 and :273-306              */
/*   xyz_max_voxel = range[0,2,4] + 0.5*v ; xyz_min_voxel = range[1,3,5] - 0.5*v */
/*   idx = floor((p - min_voxel) / v)  (IEEE division, :251-253)                */
/*   keep iff max_voxel > p && p > min_voxel on all axes (:256-257)            */
/*   clamp idx to [0, G-1] after unique (:267)                                  */
/*   pose idx: same formula, NO clamp, no bound test (:297-301)                */
/* ------------------------------------------------------------------------ */
static inline int orc_point_to_voxel(const float *p, const float *range6, const float *vox3, int g, int *ix3)
{
    int keep = 1;
    for (int a = 0; a < 3; ++a) {
        const float half = 0.5f * vox3[a];
        const float vmax = range6[2 * a] + half;
        const float vmin = range6[2 * a + 1] - half;
        const float q = (p[a] - vmin) / vox3[a];
        const float f = floorf(q);
        keep &= (vmax > p[a]) && (p[a] > vmin);
        long long i = isfinite(f) ? (long long)f : 0;
        if (i < 0) i = 0;
        if (i > g - 1) i = g - 1;
        ix3[a] = (int)i;
    }
    return keep;
}

/* per-point (pre-unique, pre-clamp is irrelevant for kept points except the
 * top face) indices; idx[p] = -1 for dropped points.  Used to pin A3 point by
 * point against the reference before the set reduction. */
ORC_API void orc_points_to_idx(const float *world /*[N,HW,3]*/, const uint8_t *fg /*[N,HW]*/,
                               const float *range_gt /*[N,6]*/, const float *voxel_size /*[N,3]*/,
                               int n, int64_t hw, int g, int32_t *idx /*[N,HW,3]*/)
{
    for (int e = 0; e < n; ++e)
        for (int64_t p = 0; p < hw; ++p) {
            const int64_t q = (int64_t)e * hw + p;
            int ix[3];
            int keep = fg[q] && orc_point_to_voxel(world + q * 3, range_gt + e * 6, voxel_size + e * 3, g, ix);
            idx[q * 3 + 0] = keep ? ix[0] : -1;
            idx[q * 3 + 1] = keep ? ix[1] : -1;
            idx[q * 3 + 2] = keep ? ix[2] : -1;
        }
}

ORC_API void orc_pose_to_idx(const float *poses_xyz /*[N,3]*/, const float *range_gt, const float *voxel_size,
                             int n, int64_t *pose_idx /*[N,3]*/)
{
    for (int e = 0; e < n; ++e)
        for (int a = 0; a < 3; ++a) {
            const float v = voxel_size[e * 3 + a];
            const float vmin = range_gt[e * 6 + 2 * a + 1] - 0.5f * v;
            pose_idx[e * 3 + a] = (int64_t)floorf((poses_xyz[e * 3 + a] - vmin) / v);
        }
}

/* ------------------------------------------------------------------------ */
/* A5  ray casting: gennbv/utils.py:48-167 (kernel text), SURVEY appendix A.1  */
/*   Integer 3-D Bresenham, dominant axis tested x, y, z; start voxel emitted  */
/*   if in bounds; loop i in [0, d_a) while emitted < max_pts; endpoint        */
/*   included; only in-bound voxels are emitted.  `visit` is called per        */
/*   emitted voxel.  Returns the emitted count (= trajectory_lengths[ray]).    */
/* ------------------------------------------------------------------------ */
typedef void (*orc_visit_fn)(void *ctx, int x, int y, int z);

static inline int orc_inb(int x, int y, int z, int g)
{
    return x >= 0 && x < g && y >= 0 && y < g && z >= 0 && z < g;
}

static int orc_bresenham_walk(int x0, int y0, int z0, int x1, int y1, int z1, int g, int max_pts,
                              orc_visit_fn visit, void *ctx)
{
    const int d[3] = {abs(x1 - x0), abs(y1 - y0), abs(z1 - z0)};
    const int s[3] = {x0 < x1 ? 1 : -1, y0 < y1 ? 1 : -1, z0 < z1 ? 1 : -1};
    int dm = d[0] > d[1] ? d[0] : d[1];
    dm = dm > d[2] ? dm : d[2];
    /* dominant axis a, minors b, c in the reference's order */
    int a, b, c;
    if (dm == d[0]) { a = 0; b = 1; c = 2; }
    else if (dm == d[1]) { a = 1; b = 0; c = 2; }
    else { a = 2; b = 0; c = 1; }
    int pos[3] = {x0, y0, z0};
    int p1 = 2 * d[b] - d[a];
    int p2 = 2 * d[c] - d[a];
    int emitted = 0;
    if (orc_inb(pos[0], pos[1], pos[2], g)) {
        if (visit) visit(ctx, pos[0], pos[1], pos[2]);
        emitted++;
    }
    for (int i = 0; i < d[a] && emitted < max_pts; ++i) {
        if (p1 >= 0) { pos[b] += s[b]; p1 -= 2 * d[a]; }
        if (p2 >= 0) { pos[c] += s[c]; p2 -= 2 * d[a]; }
        pos[a] += s[a];
        p1 += 2 * d[b];
        p2 += 2 * d[c];
        if (orc_inb(pos[0], pos[1], pos[2], g)) {
            if (visit) visit(ctx, pos[0], pos[1], pos[2]);
            emitted++;
        }
    }
    return emitted;
}

struct orc_traj_ctx { int32_t *out; int n; };
static void orc_traj_visit(void *c, int x, int y, int z)
{
    struct orc_traj_ctx *t = (struct orc_traj_ctx *)c;
    t->out[t->n * 3 + 0] = x; t->out[t->n * 3 + 1] = y; t->out[t->n * 3 + 2] = z;
    t->n++;
}

/* Same I/O contract as the reference kernel launch (utils.py:170-196,204-220):
 * trajectory_pts [num_rays, 3G, 3] int32 (zero filled by the caller),
 * trajectory_lengths [num_rays] int32. */
ORC_API void orc_bresenham3d(const int32_t *source3, const int32_t *targets /*[R,3]*/, int num_rays, int g,
                             int32_t *trajectory_pts, int32_t *trajectory_lengths)
{
    const int max_pts = 3 * g;
    for (int r = 0; r < num_rays; ++r) {
        struct orc_traj_ctx ctx = {trajectory_pts + (int64_t)r * max_pts * 3, 0};
        trajectory_lengths[r] = orc_bresenham_walk(source3[0], source3[1], source3[2], targets[r * 3],
                                                   targets[r * 3 + 1], targets[r * 3 + 2], g, max_pts,
                                                   orc_traj_visit, &ctx);
    }
}

/* ------------------------------------------------------------------------ */
/* A6/A7  update_occ_grid + grid_occupancy_tri_cls                             */
/*   gennbv/env/env_train_gennbv.py:277-326 ; gennbv/utils.py:309-325          */
/*   Set semantics: per env                        */
/*     hit  = unique clamped voxel indices of kept fg points                   */
/*     path = union of ray voxels pose_idx -> hit (endpoint included)          */
/*     prob[path] = prob[path] - 0.05 (once per voxel); prob[hit] = 1.0        */
/*   all envs: tri = (prob > 0.5) - (prob < 0.0);                              */
/*             scanned = clip(scanned + occ*gt, 0, 1), occ = 1 on hit.         */
/*   Envs with no kept point skip the per-env part (:298-299).                 */
/*   `reset_mask[e]` != 0 zeroes prob/scanned of env e BEFORE the update       */
/*   (reset_idx :416-420 does it after the previous step's obs; folding the    */
/*   zeroing into the next update is observationally identical).               */
/* ------------------------------------------------------------------------ */
struct orc_mask_ctx { uint8_t *mask; int g; };
static void orc_mask_visit(void *c, int x, int y, int z)
{
    struct orc_mask_ctx *m = (struct orc_mask_ctx *)c;
    m->mask[((int64_t)x * m->g + y) * m->g + z] = 1;
}

ORC_API void orc_update_occ_grid(const float *depth /*[N,H,W] processed*/, const float *seg,
                                 const float *c2w, const float *inv_intri, const float *poses_xyz /*[N,3]*/,
                                 const float *range_gt, const float *voxel_size, const float *grid_gt /*[N,G^3]*/,
                                 const uint8_t *reset_mask /*[N] or NULL*/, int n, int h, int w, int g,
                                 float *prob_grid, float *scanned_gt_grid, float *tri_cls /*[N,G^3]*/,
                                 uint8_t *hit_mask_out /*[N,G^3] or NULL*/, uint8_t *path_mask_out /*or NULL*/,
                                 int32_t *coverage_count /*[N] or NULL: sum(scanned) for binary gt*/)
{
    const int64_t g3 = (int64_t)g * g * g, hw = (int64_t)h * w;
    /* envs are independent (no cross-env term anywhere in update_occ_grid): one env per OpenMP thread when built with
     * -fopenmp (bench.py's cpu_baseline uses every host core); the per-env arithmetic and its order are unchanged */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int e = 0; e < n; ++e) {
        uint8_t *hit = (uint8_t *)malloc(g3), *path = (uint8_t *)malloc(g3);
        float *prob = prob_grid + e * g3, *scan = scanned_gt_grid + e * g3, *tri = tri_cls + e * g3;
        const float *gt = grid_gt + e * g3;
        if (reset_mask && reset_mask[e]) {
            memset(prob, 0, g3 * sizeof(float));
            memset(scan, 0, g3 * sizeof(float));
        }
        memset(hit, 0, g3);
        memset(path, 0, g3);
        int64_t nhit = 0;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const int64_t p = (int64_t)e * hw + (int64_t)y * w + x;
                if (!(seg[p] > 50.0f)) continue;
                float wp[3];
                int ix[3];
                orc_pixel_to_world(depth[p], (float)x, (float)y, inv_intri, c2w + (int64_t)e * 16, wp);
                if (!orc_point_to_voxel(wp, range_gt + e * 6, voxel_size + e * 3, g, ix)) continue;
                uint8_t *cell = hit + ((int64_t)ix[0] * g + ix[1]) * g + ix[2];
                nhit += !*cell;
                *cell = 1;
            }
        if (nhit) {
            int64_t src[3];
            orc_pose_to_idx(poses_xyz + e * 3, range_gt + e * 6, voxel_size + e * 3, 1, src);
            struct orc_mask_ctx ctx = {path, g};
            for (int x = 0; x < g; ++x)
                for (int y = 0; y < g; ++y)
                    for (int z = 0; z < g; ++z)
                        if (hit[((int64_t)x * g + y) * g + z])
                            /* int32 truncation mirrors pts_source.int() utils.py:32 */
                            orc_bresenham_walk((int32_t)src[0], (int32_t)src[1], (int32_t)src[2], x, y, z, g, 3 * g,
                                               orc_mask_visit, &ctx);
        }
        int32_t cov = 0;
        for (int64_t v = 0; v < g3; ++v) {
            float pr = prob[v];
            if (path[v]) pr = pr - 0.05f;
            if (hit[v]) pr = 1.0f;
            prob[v] = pr;
            tri[v] = (pr > 0.5f ? 1.0f : 0.0f) - (pr < 0.0f ? 1.0f : 0.0f);
            float sc = scan[v] + (hit[v] ? 1.0f : 0.0f) * gt[v];
            sc = sc < 0.0f ? 0.0f : (sc > 1.0f ? 1.0f : sc);
            scan[v] = sc;
            cov += sc != 0.0f;
        }
        if (coverage_count) coverage_count[e] = cov;
        if (hit_mask_out) memcpy(hit_mask_out + e * g3, hit, g3);
        if (path_mask_out) memcpy(path_mask_out + e * g3, path, g3);
        free(hit);
        free(path);
    }
}

/* ------------------------------------------------------------------------ */
/* C2  TensorRolloutBuffer_Grid_Obs.compute_returns_and_advantage              */
/*     stable_baselines3/common/buffers.py:706-724                              */
/*   reverse scan; torch evaluates every op separately in fp32 (no FMA):       */
/*     nnt   = 1 - dones (last row) | 1 - episode_starts[t+1]                   */
/*     delta = (r + (gamma*nv)*nnt) - v                                         */
/*     last  = delta + ((gamma*lambda)*nnt)*last     (gamma*lambda in double)   */
/*     returns = advantages + values                                            */
/*   Layout [T,N] row-major (the reference's [T,N,1]).                          */
/* ------------------------------------------------------------------------ */
ORC_API void orc_gae_sb3(const float *rewards, const float *values, const uint8_t *episode_starts,
                         const float *last_values /*[N]*/, const uint8_t *dones /*[N]*/, int t_steps, int n,
                         double gamma, double gae_lambda, float *advantages, float *returns)
{
    const float g = (float)gamma, gl = (float)(gamma * gae_lambda);
    for (int e = 0; e < n; ++e) {
        float last = 0.0f;
        for (int t = t_steps - 1; t >= 0; --t) {
            float nnt, nv;
            if (t == t_steps - 1) { nnt = 1.0f - (float)dones[e]; nv = last_values[e]; }
            else { nnt = 1.0f - (float)episode_starts[(int64_t)(t + 1) * n + e]; nv = values[(int64_t)(t + 1) * n + e]; }
            const int64_t i = (int64_t)t * n + e;
            float t1 = g * nv;
            float t2 = t1 * nnt;
            float t3 = rewards[i] + t2;
            float delta = t3 - values[i];
            float u1 = gl * nnt;
            float u2 = u1 * last;
            last = delta + u2;
            advantages[i] = last;
            returns[i] = last + values[i];
        }
    }
}

/* C-alt  rsl_rl RolloutStorage.compute_returns: rsl_rl/storage/rollout_storage.py:130-144
 *   nnt = 1 - dones[t]; delta = (r + ((nnt*gamma)*nv)) - v;
 *   adv = delta + (((nnt*gamma)*lam)*adv); returns[t] = adv + v
 *   advantages = returns - values (un-normalised here; the whole-buffer
 *   normalisation :143-144 is a floating-point reduction done by the caller). */
ORC_API void orc_gae_rsl(const float *rewards, const float *values, const uint8_t *dones /*[T,N]*/,
                         const float *last_values, int t_steps, int n, double gamma, double lam,
                         float *returns, float *advantages)
{
    const float g = (float)gamma, l = (float)lam;
    for (int e = 0; e < n; ++e) {
        float adv = 0.0f;
        for (int t = t_steps - 1; t >= 0; --t) {
            const int64_t i = (int64_t)t * n + e;
            const float nv = (t == t_steps - 1) ? last_values[e] : values[i + n];
            const float nnt = 1.0f - (float)dones[i];
            float a1 = nnt * g;
            float a2 = a1 * nv;
            float a3 = rewards[i] + a2;
            float delta = a3 - values[i];
            float b1 = a1 * l;
            float b2 = b1 * adv;
            adv = delta + b2;
            returns[i] = adv + values[i];
            advantages[i] = returns[i] - values[i];
        }
    }
}

ORC_API int orc_abi_version(void) { return 1; }

/* number of OpenMP threads for orc_update_occ_grid (ignored without OpenMP) */
ORC_API void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* threads orc_update_occ_grid uses (1 without OpenMP) */
ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
