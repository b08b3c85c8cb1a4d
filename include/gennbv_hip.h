/*
 * gennbv_hip.h -- C-ABI of libgennbv_hip.so, the MI355X (gfx950) implementation of
 * GenNBV's state-encoding + PPO hot path.
 *
 * Conventions (SURVEY.md section 8b "Ownership / lifetime"):
 *   - every pointer is a CALLER-OWNED DEVICE pointer (HBM) unless marked [host];
 *     nothing is retained or freed; scratch comes from a caller-owned workspace;
 *   - `stream` is the caller's hipStream_t passed as void* (NULL = default stream);
 *     all work is enqueued on it, nothing synchronises the device;
 *   - return value: 0 on success, otherwise a hipError_t value
 *     (1 = hipErrorInvalidValue for bad arguments). Never throws.
 *   - tensors are dense row-major fp32 unless stated; grids are [N, G, G, G]
 *     C-order over (X, Y, Z) exactly like the reference's torch tensors.
 *
 * Each entry point names the reference interface it replaces (paths relative to
 * the zjwzcx/GenNBV root). INTEGRATION.md shows the ctypes binding a maintainer
 * would add on the reference side.
 */
#ifndef GENNBV_HIP_H
#define GENNBV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNBV_ABI_VERSION 5

int gnbv_abi_version(void);
/* Name of the device architecture the library was compiled for ("gfx950"). [host] */
const char *gnbv_build_arch(void);

/* ------------------------------------------------------------------------- */
/* A1  Env_Train_Base.post_process_camera_tensor, depth + seg branch          */
/*     gennbv/env/env_train_base.py:521-534                                    */
/* ------------------------------------------------------------------------- */
int gnbv_post_process_depth(const float *depth_raw, const float *seg_raw, int64_t count,
                            float depth_sense_dist, float *depth_out, float *seg_out, void *stream);

/* A1 rgb branch (env_train_base.py:517-520): RGBA u8 [N,H,W,4] -> nearest
 * resize to [oh,ow] -> grayscale f32 [N,1,oh,ow].  Parity unpinned (torchvision). */
int gnbv_rgb_to_gray(const uint8_t *rgba, int n, int h, int w, int oh, int ow, float *gray,
                     int64_t gray_row_stride /* floats between envs, >= oh*ow */, void *stream);

/* ------------------------------------------------------------------------- */
/* A2  Env_Train_GenNBV.back_projection_fg  (env_train_gennbv.py:494-533)      */
/*     depth/seg are the PROCESSED tensors; c2w [N,4,4] is inv(view^T) @        */
/*     blender2opencv with env_origins subtracted (host plumbing, :512-514).   */
/*     world [N,HW,3]; fg [N,HW] u8 (seg > 50).                                 */
/* ------------------------------------------------------------------------- */
int gnbv_back_projection(const float *depth, const float *seg, const float *c2w, const float *inv_intri /*[host] [3,3]*/,
                         int n, int h, int w, float *world, uint8_t *fg, void *stream);

/* A3  scanned_pts_to_idx_3D (gennbv/utils.py:230-270), per point, before the
 *     set reduction: idx [N,HW,3] int32, (-1,-1,-1) for dropped points. */
int gnbv_points_to_idx(const float *world, const uint8_t *fg, const float *range_gt /*[N,6]*/,
                       const float *voxel_size /*[N,3]*/, int n, int64_t hw, int g, int32_t *idx, void *stream);

/* A4  pose_coord_to_idx_3D (gennbv/utils.py:273-306, if_col=False): no clamp. */
int gnbv_pose_to_idx(const float *poses_xyz /*[N,3]*/, const float *range_gt, const float *voxel_size, int n,
                     int64_t *pose_idx /*[N,3]*/, void *stream);

/* A5  bresenham3D_pycuda kernel launch (gennbv/utils.py:170-220): one source,
 *     num_rays targets; trajectory_pts [num_rays, 3*map_size, 3] int32 and
 *     trajectory_lengths [num_rays] int32 must be zero-filled by the caller
 *     (as utils.py:40-41 does). */
int gnbv_bresenham3d(const int32_t *source_pts /*[3]*/, const int32_t *target_pts /*[R,3]*/, int num_rays, int map_size,
                     int32_t *trajectory_pts, int32_t *trajectory_lengths, void *stream);

/* A7  grid_occupancy_tri_cls (gennbv/utils.py:309-325), return_tri_cls_only. */
int gnbv_grid_tri_cls(const float *grid_prob, int64_t count, float threshold_occu, float threshold_free,
                      float *grid_tri_cls, void *stream);

/* ------------------------------------------------------------------------- */
/* A1-A7 fused: Env_Train_GenNBV.update_occ_grid (env_train_gennbv.py:277-326)  */
/*   One call = one environment step for all N envs, three launches:           */
/*     hit-mask scatter (LDS-staged bitmask) -> ray cast (LDS path bitmask)     */
/*     -> streaming grid update.                                                */
/*   depth_raw / seg_raw are the RAW camera tensors (A1 is fused).              */
/*   reset_mask [N] u8 or NULL: env rows whose prob/scanned grids are treated   */
/*   as zero before the update (reset_idx :416-420 folded into the next step).  */
/*   tri_out: row e starts at tri_out + e*tri_row_stride (floats) so the        */
/*   tri-class grid can land directly inside the flat observation row           */
/*   (wrapper key order state, grid, state_rgb).                                */
/*   coverage_count [N] int32 = number of non-zero scanned_gt voxels            */
/*   (== scanned_gt.sum() for the reference's binary GT, :537).                 */
/*   workspace: gnbv_voxel_workspace_bytes(n, g) bytes, 256-byte aligned.       */
/* ------------------------------------------------------------------------- */
size_t gnbv_voxel_workspace_bytes(int n, int g);
/* The same two bitmask arrays + per-env ray lists (one int32 per distinct hit voxel and image chunk, capacity h*w per
 * env): with a workspace of at least this size the update runs the hit-list + load-balanced ray-cast launches
 * (csrc/voxel.hip: k_hit_list, k_ray_list; grids whose bitmask fits no workgroup's LDS -- G > 104 -- k_hit_atomic, k_ray_slab); with
 * the smaller mask-only workspace it falls back to k_hit_mask + k_raycast.
 * Same results either way. */
size_t gnbv_voxel_workspace_bytes_hw(int n, int g, int h, int w);
/* gnbv_update_occ_grid_coded, workspace_flags: the caller guarantees that the two mask arrays and the ray counts of the
 * workspace are ZERO on entry -- freshly zero-initialised, or left so by the previous call made with this flag -- and the
 * call leaves them zero (its grid-update launch clears every word it consumes): the 16 MB fill launch per call goes away.
 * Do not set it on a workspace that another entry point (or a call without the flag) has used since. */
#define GNBV_VOXEL_WS_CLEAN 1

int gnbv_update_occ_grid(const float *depth_raw, const float *seg_raw, const float *c2w,
                         const float *inv_intri /*[host] [3,3]*/,
                         const float *poses_xyz /*[N,3]*/, int64_t poses_row_stride /*floats*/,
                         const float *range_gt, const float *voxel_size, const float *grid_gt,
                         const uint8_t *reset_mask, int n, int h, int w, int g, float depth_sense_dist,
                         float *prob_grid, float *scanned_gt_grid, float *tri_out, int64_t tri_row_stride,
                         int32_t *coverage_count, void *workspace, size_t workspace_bytes, void *stream);

/* Packed variant of gnbv_update_occ_grid for a BINARY ground truth (the reference's GT is an
 * occupancy indicator): grid_gt and scanned_gt_grid are bitmasks [N, gnbv_grid_bit_words(G)] u32
 * (bit v = voxel (x*G+y)*G+z). scanned = clip(scanned + occ*gt, 0, 1) == scanned | (hit & gt)
 * exactly, coverage = popcount. Identical results, half the HBM traffic of the streaming pass. */
int gnbv_grid_bit_words(int g);
int gnbv_pack_grid_bits(const float *grid /*[N,G^3]*/, int n, int g, uint32_t *bits, int *not_binary /*[1] or NULL*/,
                        void *stream);
int gnbv_unpack_grid_bits(const uint32_t *bits, int n, int g, float *grid /*[N,G^3] of {0,1}*/, void *stream);
int gnbv_update_occ_grid_packed(const float *depth_raw, const float *seg_raw, const float *c2w,
                                const float *inv_intri /*[host] [3,3]*/, const float *poses_xyz, int64_t poses_row_stride,
                                const float *range_gt, const float *voxel_size, const uint32_t *gt_bits,
                                const uint8_t *reset_mask, int n, int h, int w, int g, float depth_sense_dist,
                                float *prob_grid, uint32_t *scanned_bits, float *tri_out, int64_t tri_row_stride,
                                int32_t *coverage_count, void *workspace, size_t workspace_bytes, void *stream);

/* Debug/parity view of the workspace after gnbv_update_occ_grid: expands the
 * hit / path bitmasks to u8 [N,G^3] (either output may be NULL). */
int gnbv_unpack_masks(const void *workspace, int n, int g, uint8_t *hit_u8, uint8_t *path_u8, void *stream);

/* Coded probability grid (same A1-A7 step, 1 byte per voxel instead of an fp32 prob_grid): between two resets a voxel's
 * probability is a function of (base, k) -- base = 1 after a hit (prob = 1.0, env_train_gennbv.py:311) / 0 since the
 * reset, k = number of "prob -= 0.05" path steps since (:308) -- so code = base << 7 | k is exact for episodes of at
 * most 127 steps (*overflow is set to 1 if a counter saturates).  gnbv_prob_code_tables fills the 256-entry HOST
 * tables prob_lut[code] (the exact fp32 iteration) and tri_lut[code] (A7); pass DEVICE copies to the kernels. */
void gnbv_prob_code_tables(float *prob_lut /*[256] host or NULL*/, float *tri_lut /*[256] host or NULL*/);
int gnbv_decode_prob_grid(const uint8_t *prob_code, int64_t count, const float *prob_lut /*[256] device*/, float *prob_out,
                          void *stream);
int gnbv_update_occ_grid_coded(const float *depth_raw, const float *seg_raw, const float *c2w, const float *inv_intri /*[host]*/,
                               const float *poses_xyz, int64_t poses_row_stride, const float *range_gt, const float *voxel_size,
                               const uint32_t *gt_bits, const uint8_t *reset_mask, int n, int h, int w, int g,
                               float depth_sense_dist, uint8_t *prob_code /*[N,G^3]*/, const float *tri_lut /*[256] device*/,
                               uint32_t *scanned_bits, float *tri_out /*NULL: compact observations, int8 rows only*/,
                               int64_t tri_row_stride,
                               int8_t *tri_i8 /*NULL, or the tri-class grid as int8 rows (-1/0/1): row e at tri_i8 + e*stride;
                                                at least one of tri_out / tri_i8 must be given*/,
                               int64_t tri_i8_row_stride /*bytes*/, int32_t *coverage_count,
                               int32_t *overflow /*[1] device or NULL*/, void *workspace, size_t workspace_bytes,
                               int workspace_flags /* GNBV_VOXEL_WS_* */, void *stream);


/* ------------------------------------------------------------------------- */
/* A8/A9  environment-step bookkeeping (no simulator: recorded/synthetic feed)  */
/* ------------------------------------------------------------------------- */
/* Action lattice of the task (gennbv/env/config_gennbv_train.py:62-69). [host struct] */
typedef struct GnbvLattice {
    int64_t clip_low[6], clip_up[6], init_action[6];
    float action_unit[6], pose_low[6], init_pose[6];
} GnbvLattice;

/* Env_Train_GenNBV.step head (env_train_gennbv.py:246-255) + post_physics_step's
 * episode_length_buf += 1 (:337): clip actions, force init_action where
 * episode_length_buf == 0, poses = action*unit + low. actions int64 [N,6], poses f32 [N,6]. */
int gnbv_env_pre_step(const int64_t *actions_in, const GnbvLattice *lattice /*[host]*/, int64_t *episode_length_buf,
                      int n, int64_t *actions_out, float *poses_out, void *stream);

/* update_obs_buf pose deque (:273-275) + obs["state"] (:361): pose_hist [N,stack,6]
 * oldest->newest is shifted, the new pose appended, and the row is written to
 * obs + e*obs_row_stride. reset_mask [N] (may be NULL): history was refilled with
 * init_pose_buf by reset_idx (:397-400). */
int gnbv_env_obs_state(float *pose_hist, const float *poses, const uint8_t *reset_mask, const GnbvLattice *lattice /*[host]*/,
                       int n, int stack, float *obs, int64_t obs_row_stride, void *stream);

/* post_process_camera_tensor rgb branch (env_train_base.py:517-520) + rgb deque (k=2)
 * + obs["state_rgb"] (:363): writes [older | newest] gray frames (2*oh*ow floats) at
 * obs_rgb + e*obs_row_stride; gray_prev [N,oh*ow] is the persistent older frame. */
int gnbv_env_obs_rgb(const uint8_t *rgba, float *gray_prev, const uint8_t *reset_mask, int n, int h, int w, int oh, int ow,
                     float *obs_rgb, int64_t obs_row_stride, void *stream);

/* The three calls above as ONE launch (round 5; a new entry point of ABI 5, no layout changes): same arguments, same arithmetic, bit-identical outputs -- what a reference-side integration
 * would issue between `step(actions)` and `update_occ_grid` (env_train_gennbv.py:246-275).  obs_rgb = the row's state_rgb slice (same
 * row stride as obs). */
int gnbv_env_observe(const int64_t *actions_in, const GnbvLattice *lattice /*[host]*/, int64_t *episode_length_buf, int n, int64_t *actions_out,
                     float *poses_out, float *pose_hist, const uint8_t *reset_mask, int stack, float *obs, int64_t obs_row_stride,
                     const uint8_t *rgba, float *gray_prev, int h, int w, int oh, int ow, float *obs_rgb, void *stream);

/* compute_reward (env_train_base.py:377-398), _reward_* / check_termination /
 * reset_idx (env_train_gennbv.py:377-457,535-556), update_extra_episode_info
 * (env_train_base.py:629-639). All pointers device, arrays [N] unless noted. [host struct] */
typedef struct GnbvEnvPost {
    int n;
    int only_positive;              /* cfg.rewards.only_positive_rewards */
    int64_t max_episode_length;
    float scale_cov, scale_short, scale_term; /* reward scales * dt, rounded to fp32 */
    float coverage_threshold;       /* 0.99 */
    const int32_t *coverage_count;  /* from gnbv_update_occ_grid */
    const float *num_valid_voxel_gt;
    float *prev_ratio;              /* in/out: reward_ratio_buf[-1] */
    int64_t *episode_length_buf;    /* in/out */
    float *rewards;                 /* out */
    uint8_t *dones;                 /* out: reset_buf */
    uint8_t *reset_mask;            /* out: envs whose grids/history reset before the next step */
    uint8_t *step_time_out;         /* out: time_out_buf of this step */
    uint8_t *extras_time_outs;      /* in/out: infos["time_outs"] (refreshed only if any env reset) */
    float *coverage_ratio;          /* out */
    float *episode_sums;            /* in/out [3,N]: surface_coverage, short_path, termination */
    float *cur_reward_sum;          /* in/out */
    float *cur_episode_length;      /* in/out */
    float *ring_reward;             /* in/out [ring_len]: rewbuffer (deque maxlen 100) */
    float *ring_length;             /* in/out [ring_len]: lenbuffer */
    int64_t *ring_state;            /* in/out [1]: episodes finished so far */
    int ring_len;
    double *episode_info;           /* out [6] or NULL: extras["episode"] as it stands after THIS step, fp64:
                                       [0] generation = number of dict objects the reference has created so far (a new
                                           one on every step where some env resets, env_train_gennbv.py:424; steps in
                                           between mutate and re-emit the SAME dict, so earlier buffer entries alias it),
                                       [1] episode_reward, [2] episode_length = np.mean of the rewbuffer / lenbuffer
                                           deques (env_train_base.py:638-639; 0 when empty),
                                       [3..5] rew_<name> = mean(episode_sums[name][reset envs]) / max_episode_length_s
                                           of the dict's creation step (:425-427) */
    double *episode_state;          /* in/out [4], required with episode_info: generation, rew_<name> x 3 */
    float max_episode_length_s;     /* cfg.env.episode_length_s */
} GnbvEnvPost;

int gnbv_env_post_step(const GnbvEnvPost *args /*[host]*/, void *stream);

/* Tail of one rollout step in one launch: the time-out bootstrap `rewards += gamma * squeeze(terminal_value * time_outs)`
 * (on_policy_algorithm_grid_obs.py:205-208; same fp32 operation order; terminal_value_stride = 1: env i uses
 * terminal_value[i]; 0: every env uses terminal_value[0] -- what the reference computes, its `predict_values(new_obs)[0]`
 * takes ROW 0 of the [N,1] values and broadcasts it over the envs) and the five copies of
 * TensorRolloutBuffer_Grid_Obs.add (stable_baselines3/common/buffers.py:676-704) into row `step` of the buffer arrays
 * (caller passes the row pointers): actions int64 [N,A] -> f32, episode_starts bool/u8 [N] -> u8, rewards / values /
 * log_probs f32 [N].  time_outs: bool/u8 [N]. */
int gnbv_rollout_add(int n, int action_dim, const int64_t *actions, const float *rewards, const uint8_t *time_outs,
                     const float *terminal_value, int terminal_value_stride, float gamma, const uint8_t *episode_starts,
                     const float *values, const float *log_probs, float *buf_actions, float *buf_rewards, uint8_t *buf_episode_starts,
                     float *buf_values, float *buf_log_probs, void *stream);

/* ------------------------------------------------------------------------- */
/* B1  Hybrid_Encoder grid branch (gennbv/network/hybrid_encoder.py:38-45,90-94): */
/*     Conv3d(1,16,3,s2) BN ReLU Conv3d(16,16,3,s2) BN ReLU -> flatten             */
/*     fp32, forward + backward, hand-written MFMA kernels (csrc/encoder.hip).     */
/* ------------------------------------------------------------------------- */
/* Device pointers to the parameters / buffers of `naive_encoder_grid`
 * (torch layouts: w1 [16,1,3,3,3], w2 [16,16,3,3,3], vectors [16]). [host struct] */
typedef struct GnbvEncoderParams {
    const float *w1, *b1, *bn1_w, *bn1_b;
    float *bn1_rm, *bn1_rv;       /* running_mean / running_var (updated when training) */
    int64_t *bn1_nbt;             /* num_batches_tracked [1] or NULL */
    const float *w2, *b2, *bn2_w, *bn2_b;
    float *bn2_rm, *bn2_rv;
    int64_t *bn2_nbt;
    float eps, momentum;          /* 1e-5, 0.1 (torch.nn.BatchNorm3d defaults) */
    const int8_t *grid_i8;        /* NULL, or an int8 copy of the grid slices (values -1/0/1, as written by
                                     gnbv_update_occ_grid_coded): sample b reads grid_i8 + (rows ? rows[b] : b) *
                                     grid_i8_row_stride bytes instead of obs_grid (a quarter of the input traffic);
                                     used by the LDS-staged kernels when grid % 16 == 0, otherwise obs_grid is read --
                                     unless obs_grid == NULL (compact observations: the grid exists only as these
                                     rows; any grid size, fp32 activations) */
    int64_t grid_i8_row_stride;
    const int32_t *autocorr;      /* NULL, or per-sample input autocorrelation rows (gnbv_input_autocorr over the same
                                     int8 rows, same row numbering): the fused backward sums rows[b]'s rows instead of
                                     recomputing the minibatch's autocorrelation */
    int64_t autocorr_row_stride;  /* ints */
    /* ---- data-parallel replicas: BatchNorm over the GLOBAL minibatch.  world <= 1 / sync_sum == NULL: a single
     * replica, nothing below is read.  Otherwise the training forward / backward call sync_sum three times -- BN2 batch sums
     * (forward), BN2-backward sums and BN1-backward sums (backward) -- each time on 32 doubles of `sync_buf`, and the
     * statistics of BatchNorm-1 come from `autocorr_global`.  Needs the fused backward (aligned int8 rows + autocorr rows). */
    int world;                    /* number of replicas */
    int (*sync_sum)(void *ctx, int offset, int n, void *stream);  /* [host callback] sum sync_buf[offset, offset + n) over the
                                     replicas in place, ordered on `stream` (e.g. an all-reduce enqueued there); returns 0 */
    void *sync_ctx;
    double *sync_buf;             /* device [96] */
    const int32_t *autocorr_global; /* device [768]: sum of the autocorrelation rows of ALL replicas' minibatch rows */
    /* ---- operand ranges of the split-f16 kernels (ABI 2).  At G = 64 the conv stack runs on the f16 matrix pipe with every
     * fp32 operand written as hi + lo f16 halves under fixed power-of-two scalings: relu(bn1(y1)) <= 253.9, |W2| < 63,
     * fc_grid inputs <= 1015, |W_fc| < 15.8.  Outside those ranges the split kernels would clamp, so: */
    const int32_t *autocorr_total;/* NULL, or device [768]: the SUM of the minibatch's autocorrelation rows, computed by the caller (the
                                     permutation is fixed for a whole train() call: one table for all its minibatches) -- the training
                                     forward then skips its gather of `autocorr` rows (128 scattered 3 KiB rows, ~6 us of dependent
                                     round trips on the update's critical path); `autocorr` must still be set (it selects the path) */
    int force_fp32;               /* 1: never take the split-f16 kernels (the fp32-MFMA kernels have no range limits); the host
                                     mirror sets it when a parameter pre-check finds a weight outside its range */
    int32_t *range_flag;          /* NULL, or a device word the kernels OR bits into when an ACTIVATION bound is reached:
                                     2 = a BatchNorm-1 channel whose parameter bound |scale| sum|W1| + |scale b1 + shift| exceeds
                                     253 (conv1's input is tri-class, |x| <= 1); 4 = a feature (fc_grid input) above 1000.
                                     The caller reads it once per train() / rollout; a non-zero word means the results of the
                                     calls since the last check may be clamped: raise, or repeat with force_fp32 */
    int eval_prepared;            /* (ABI 5) inference only (training == 0), 0 by default.  1: the caller ran gnbv_encoder_eval_prepare()
                                     for THESE parameters, this (batch, grid), this `bn_state` and this `workspace` since the parameters
                                     or BatchNorm's running statistics last changed, and nothing else has written bn_state / used the
                                     workspace since: the forward then skips the launches that only depend on the parameters
                                     (BatchNorm scale / shift of both layers, the conv2 weight images) -- three small kernels on the
                                     critical path of every env step of a rollout (sb3/ppo_grid_obs.py collect_rollouts: the
                                     parameters are fixed between two train() calls).  Ignored where the inference path has no
                                     such launches to skip. */
} GnbvEncoderParams;

typedef struct GnbvEncoderGrads {  /* outputs, same shapes as the parameters */
    float *w1, *b1, *bn1_w, *bn1_b, *w2, *b2, *bn2_w, *bn2_b;
} GnbvEncoderGrads;

/* Input autocorrelation of the conv1 patches of n int8 grid rows (values -1/0/1): out row e (gnbv_input_autocorr_row_ints()
 * = 768 ints: the 16x16 tiles [0..15][0..15], [0..15][16..31], [16..31][16..31] of the symmetric R[32][32]) holds
 * R[t][t'] = sum over the O1^3 conv1 output positions of x[pos, t] * x[pos, t'], t = 0..26 the taps of
 * hybrid_encoder.py:39's Conv3d(1, 16, k3, s2), t = 27 a constant 1 (R[t][27] = sum x, R[27][27] = O1^3).  Exact (int8
 * MFMA).  BatchNorm-backward of that layer is linear in the input, so these rows replace every reduction over the
 * layer's activations that does not involve the incoming gradient (encoder.hip, k_conv2_dgrad_c1w).  G % 16 == 0,
 * 3 G^2 <= 64 KiB. */
int gnbv_input_autocorr_row_ints(void);
int gnbv_input_autocorr(const int8_t *grid_i8, int64_t grid_i8_row_stride, int n, int grid, int32_t *out, int64_t out_row_stride,
                        void *stream);

size_t gnbv_encoder_workspace_bytes(int batch, int grid);
/* The parameter-only part of an INFERENCE forward (see GnbvEncoderParams.eval_prepared): BatchNorm-1 / -2 scale, shift, mean, rstd
 * from the running statistics into bn_state, the conv2 weight images into the workspace.  Returns 0 when done, and
 * GNBV_ERR_NOT_APPLICABLE (-2) -- nothing launched, nothing to skip -- where the inference forward of this (params, grid) does
 * not take the one-launch conv1 + conv2 kernel (then call gnbv_encoder_grid_forward with eval_prepared = 0 as before). */
#define GNBV_ERR_NOT_APPLICABLE (-2)
int gnbv_encoder_eval_prepare(int batch, int grid, const GnbvEncoderParams *params, float *bn_state, void *workspace, size_t workspace_bytes,
                              void *stream);
/* number of fp32 ELEMENTS of the layer-1 activation buffers (y1, dz1_scratch) */
size_t gnbv_encoder_y1_elems(int batch, int grid);

/* obs_grid: pointer to the grid slice of row 0 of an observation matrix (NULL with params->grid_i8 set: compact
 * observations); sample b reads
 * obs_grid + (rows ? rows[b] : b) * row_stride floats (the minibatch gather of
 * buffers.py:753-762 is fused into the read).  training != 0: BatchNorm uses batch statistics
 * and updates the running stats (unless *skip_flag != 0), else the running stats.
 * Saved for backward: y1 (gnbv_encoder_y1_elems floats: channels-last, x-parity-split, pre-BN),
 * y2 [B,16,O2^3] (pre-BN),
 * bn_state: 128 floats [2][4][16] (scale, shift, mean, rstd per layer) + 768 ints (the minibatch total of the input
 * autocorrelation, written when the forward derived BatchNorm-1's statistics from it and read back by the backward):
 * 896 four-byte words.
 * features [B, 16*O2^3] is the reference's `naive_encoder_grid(x).reshape(num_env, -1)`; NULL: not written (the BatchNorm-2 + ReLU pass is
 * left to gnbv_linear_forward_fold / gnbv_linear_bwd_dw_fold, which form the activations from y2 and bn_state in registers). */
int gnbv_encoder_grid_forward(const float *obs_grid, const int64_t *rows, int64_t row_stride, int batch, int grid,
                              const GnbvEncoderParams *params /*[host]*/, int training, const int *skip_flag, void *y1,
                              float *y2, float *bn_state, float *features, void *workspace, size_t workspace_bytes,
                              void *stream);

/* Gradients of all eight parameter tensors given d_features [B, 16*O2^3].  Must follow a gnbv_encoder_grid_forward call
 * with training != 0 and the same obs / rows / params arguments (it consumes that call's y1, y2 and bn_state, including the
 * autocorrelation total when the forward left one there).
 * dy2_scratch [B,O2^3,16] and dz1_scratch (gnbv_encoder_y1_elems floats) are caller-owned scratch. */
int gnbv_encoder_grid_backward(const float *obs_grid, const int64_t *rows, int64_t row_stride, int batch, int grid,
                               const GnbvEncoderParams *params /*[host]*/, const void *y1, const float *y2,
                               const float *bn_state, const float *d_features, float *dy2_scratch, void *dz1_scratch,
                               const GnbvEncoderGrads *grads /*[host]*/, void *workspace, size_t workspace_bytes,
                               void *stream);

/* B1  Hybrid_Encoder.output_layer_grid = Linear(16*o2^3, 256) + ReLU
 *     (gennbv/network/hybrid_encoder.py:39-42, applied at :87).  out[m][n] = act(bias[n] + sum_k x[m][k] w[n][k]),
 *     x [M][K], w [N][K] (torch.nn.Linear.weight layout), K % 4 == 0, N % 64 == 0, all pointers 16-byte aligned.
 *     Deterministic split-K (fixed summation order).  `relu` is a flag word: bit 0 fuses the ReLU; bit 1 forces the fp32-MFMA
 *     kernel (no operand-range limits) where the default is the split-f16 kernel, which CLAMPS |x| at 1015 and |w| at 15.8
 *     (the caller checks its ranges: GnbvEncoderParams.range_flag bit 4 for x, the weights on the host).
 *     workspace >= gnbv_linear_workspace_bytes(M, N, K). */
size_t gnbv_linear_workspace_bytes(int M, int N, int K);
int gnbv_linear_forward(const float *x, const float *w, const float *bias, int M, int N, int K, int relu, float *out,
                        void *workspace, size_t workspace_bytes, void *stream);

/*     Backward of the same layer (gennbv_amd/ops/encoder_ops.py::_LinearReluFn; the reference leaves it to autograd):
 *     g = d_out * (out > 0), dx = g w [M][K], dw = g^T x [N][K], db = sum_m g [N].  gnbv_linear_bwd_prep builds g's two operand
 *     images (row-scaled, split into f16 halves) in `workspace` and writes db (may be NULL); gnbv_linear_bwd_dx / _dw are the two
 *     products, each streaming its [.][K] operand once -- they may run on different streams once prep is done.
 *     M % 16 == 0, N % 16 == 0, N <= 256, K % 4 == 0, K >= 64; dx: M <= 128; dw: M <= 256.  fp32-accurate (split operands, fp32
 *     accumulation); |w| is clamped at 15.8 and |x| at 1015 like in the forward.  workspace 256-byte aligned. */
size_t gnbv_linear_bwd_workspace_bytes(int M, int N, int K);
int gnbv_linear_bwd_prep(const float *d_out, const float *out, int M, int N, float *db, void *workspace, size_t workspace_bytes,
                         void *stream);
int gnbv_linear_bwd_dx(const void *workspace, const float *w, int M, int N, int K, float *dx, void *stream);
int gnbv_linear_bwd_dw(const void *workspace, const float *x, int M, int N, int K, float *dw, void *stream);
/*     The same product, which also leaves sum(dw^2) as gnbv_linear_bwd_dw_sq_parts(K) fp64 partial sums (one per 64 columns) in
 *     `sq_partial`: the gradient-norm clip of the optimizer step (GnbvAdamStep.sq_partial) then needs no pass of its own over
 *     this -- by far the largest -- gradient. */
int gnbv_linear_bwd_dw_sq_parts(int K);
int gnbv_linear_bwd_dw_sq(const void *workspace, const float *x, int M, int N, int K, float *dw, double *sq_partial, void *stream);

/* B1  fc_grid with BatchNorm-2 + ReLU folded into its operand load (round 3; replaces Hybrid_Encoder.naive_encoder_grid[4:6] +
 *     output_layer_grid, gennbv/network/hybrid_encoder.py:40-49, as ONE product): x[m][k] = relu(scale[k / P] y[m][k] +
 *     shift[k / P]) is formed in registers from the conv stack's raw output y [M][K] (K = channels x P), the 4 K M bytes of
 *     normalised activations are never written or re-read.  Same arithmetic as gnbv_encoder_forward's k_bn_relu_apply followed by
 *     gnbv_linear_forward / gnbv_linear_bwd_dw_sq (one fma + one max in fp32, then the same split-f16 product): bit-identical
 *     results.  scale / shift: GnbvEncoderParams.bn_state + 4 x 16 and + 5 x 16 floats (layer 2) of the forward call that wrote
 *     y (features == NULL in gnbv_encoder_grid_forward skips that call's own BN2 + ReLU pass).  *range_flag (may be NULL): bit 4 when an
 *     operand passes 1000 (the f16 split clamps at 1015).  gnbv_linear_fold_ok: 1 when (M, N, K, P) can take this path
 *     (P >= 512, K % P == 0, split kernels on); otherwise materialise the activations (features != NULL) as before.
 *     d/dy of the product = the existing gnbv_linear_bwd_dx (d/dx) followed by gnbv_encoder_grid_backward (whose BN2 backward takes
 *     d/dx and y, never x). */
int gnbv_linear_fold_ok(int M, int N, int K, int P);
int gnbv_linear_forward_fold(const float *y, const float *scale, const float *shift, int P, int *range_flag, const float *w, const float *bias,
                             int M, int N, int K, int relu, float *out, void *workspace, size_t workspace_bytes, void *stream);
int gnbv_linear_bwd_dw_fold(const void *workspace, const float *y, const float *scale, const float *shift, int P, int M, int N, int K, float *dw,
                            double *sq_partial /*NULL: none*/, void *stream);


/* B1  pose-history input (gennbv/network/hybrid_encoder.py:63-74 positional_encoding with 2 frequency bands, :78-80): the state
 *     columns [0, 6 n_pose) of observation rows `rows` (NULL: rows 0 .. batch-1) of `base` (row stride in floats) ->
 *     out [batch][24 n_pose]: per pose cat(sin(p), cos(p)) of p = (x0, 2 x0, x1, 2 x1, ..., x5, 2 x5). */
int gnbv_pose_encode(const float *base, const int64_t *rows, int64_t row_stride, int batch, int n_pose, float *out, void *stream);

/* B1/B2  policy head, fused:  feat = relu([fa | fg] W_out^T + b_out)   Hybrid_Encoder.output_layer
 *                                                     (gennbv/network/hybrid_encoder.py:51-54, :89)
 *        logits = feat W_act^T + b_act, values = feat W_val^T + b_val      ActorCriticPolicy.action_net / value_net
 *                                                     (stable_baselines3/common/policies.py:975-979, :1011, :1024)
 *        fa [M][K1], fg [M][K2] (the two encoder branches, concatenated implicitly), W_out [F][K1+K2],
 *        W_act [A][F], W_val [1][F]; F, K1, K2 multiples of 16; torch.nn.Linear layouts.
 *        forward: feat [M][F] (kept for the backward), logits [M][A], values [M].
 *        backward: from d_logits [M][A], d_values [M]: d_fa, d_fg and the six parameter gradients
 *        (plain stores: pass the .grad slices for write-through).  dH_scratch: M*F floats. */
int gnbv_policy_head_forward(const float *fa, const float *fg, int M, int K1, int K2, const float *W_out, const float *b_out, int F,
                             const float *W_act, const float *b_act, int A, const float *W_val, const float *b_val, float *feat,
                             float *logits, float *values, void *stream);
int gnbv_policy_head_backward(const float *fa, const float *fg, int M, int K1, int K2, const float *feat, const float *d_logits,
                              const float *d_values, const float *W_out, int F, const float *W_act, int A, const float *W_val,
                              float *dH_scratch, float *d_fa, float *d_fg, float *gW_out, float *gb_out, float *gW_act,
                              float *gb_act, float *gW_val, float *gb_val, void *stream);

/* ------------------------------------------------------------------------- */
/* C2  TensorRolloutBuffer_Grid_Obs.compute_returns_and_advantage               */
/*     stable_baselines3/common/buffers.py:706-724.  All arrays [T,N] (the      */
/*     reference's [T,N,1]); episode_starts / dones are u8.                      */
/* ------------------------------------------------------------------------- */
int gnbv_gae_sb3(const float *rewards, const float *values, const uint8_t *episode_starts, const float *last_values,
                 const uint8_t *dones, int t_steps, int n, double gamma, double gae_lambda, float *advantages,
                 float *returns, void *stream);

/* C-alt  rsl_rl RolloutStorage.compute_returns (rsl_rl/storage/rollout_storage.py:130-142),
 *        advantages = returns - values, NOT yet normalised (:143-144 is the caller's reduction). */
int gnbv_gae_rsl(const float *rewards, const float *values, const uint8_t *dones, const float *last_values,
                 int t_steps, int n, double gamma, double lam, float *returns, float *advantages, void *stream);

/* ------------------------------------------------------------------------- */
/* C3/C4  minibatch gather + PPO loss + clip/Adam                               */
/*        stable_baselines3/common/buffers.py:753-762, ppo/ppo_grid_obs.py:196-275 */
/* ------------------------------------------------------------------------- */
/* rows [batch] int64 index the flattened [T*N] arrays (row = t*N + n). */
int gnbv_gather_minibatch(const int64_t *rows, int batch, int act_dim, const float *actions, const float *values,
                          const float *log_probs, const float *advantages, const float *returns, float *o_actions,
                          float *o_values, float *o_log_probs, float *o_adv, float *o_ret, void *stream);

/* One launch (one wave per sample; the workgroup that finishes last adds the per-sample terms in a fixed order):
 * advantage normalisation, MultiCategorical log-prob / entropy, clipped surrogate,
 * clipped value loss, entropy loss, loss = policy_scale*pg + ent_coef*ent + vf_coef*vl,
 * approx-KL, clip fraction, and d loss / d logits, d loss / d values.
 * stats row (8 floats) = pg, vl, ent, approx_kl, clip_fraction, loss, live, 0 is written at
 * stats[*stats_row] and *stats_row is incremented; *stop_flag becomes 1 (sticky) when
 * approx_kl > 1.5*target_kl (target_kl <= 0: never). [host struct, device pointers] */
typedef struct GnbvPpoLoss {
    int batch, n_logits, n_heads;
    int head_dims[8];
    int normalize_advantage;
    float clip_range, clip_range_vf /* <= 0: no value clipping */, ent_coef, vf_coef, policy_scale, target_kl;
    const float *logits;        /* [B, n_logits] */
    const float *values;        /* [B] */
    const float *actions;       /* [B, n_heads] stored as float like the reference buffer */
    const float *old_values, *old_log_prob, *advantages, *returns; /* [B] */
    float *d_logits;            /* out [B, n_logits] */
    float *d_values;            /* out [B] */
    float *head_entropy;        /* out [B, n_heads] or NULL */
    float *head_lse;            /* out [B, n_heads] or NULL */
    float *stats;               /* [rows, 8] */
    int64_t *stats_row;         /* in/out [1] */
    int *stop_flag;             /* in/out [1] or NULL */
    float *scratch;             /* [8*B + 64] device scratch, ZERO-initialised by the caller once (holds a completion
                                   counter that every call leaves at zero) */
    float *kl_out;              /* NULL, or [1]: receives approx_kl INSTEAD of setting stop_flag
                                   (data-parallel: decided on the global mean, gnbv_clip_adam_step) */
    const int64_t *rows;        /* NULL, or [B]: fused minibatch gather (buffers.py:753-762) -- actions, old_values,
                                   old_log_prob, advantages, returns then point at the whole [T*N] rollout arrays
                                   and sample i reads row rows[i] */
    const float *adv_norm;      /* NULL: the minibatch's own advantage mean / unbiased std (ppo_grid_obs.py:214-216);
                                   or [2] = (mean, 1 / (std + 1e-8)) of the GLOBAL minibatch (data-parallel replicas:
                                   the statistics of all ranks' rows, gennbv_amd/parallel.py) */
    int defer_stats;            /* 0: gnbv_ppo_loss also writes the statistics row and takes the KL stop decision (its last workgroup:
                                   a release fence + a ticket per workgroup on the update's critical path).  1: it only leaves the
                                   per-sample terms in `scratch`; the caller finishes them with gnbv_ppo_loss_finish, or -- no launch
                                   of its own -- inside gnbv_clip_adam_step_ex (GnbvAdamStep.loss_finish), before the update reads
                                   the stop flag.  With kl_out (data-parallel replicas: the rank's KL must exist before the all-reduce
                                   that carries it) only gnbv_ppo_loss_finish, launched in front of that exchange -- e.g. on a second
                                   stream beside the backward (sb3/ppo_grid_obs.py _dp_step_body). */
} GnbvPpoLoss;

int gnbv_ppo_loss(const GnbvPpoLoss *args /*[host]*/, void *stream);
int gnbv_ppo_loss_finish(const GnbvPpoLoss *args /*[host]; defer_stats == 1*/, void *stream);

/* C5  rollout side of MultiCategoricalDistribution (stable_baselines3/common/distributions.py:299-352 via
 *     ActorCriticPolicy.forward, policies.py:1024-1030): actions[b][h] ~ Categorical(softmax(logits_h[b])) by
 *     the inverse CDF at uniforms[b][h] in [0,1) (deterministic != 0: first arg-max = mode()), and
 *     log_prob[b] = sum_h log softmax(logits_h[b])[actions[b][h]].  head_dims [n_heads] is a HOST array,
 *     sum(head_dims) == n_logits, n_heads <= 8.  Same distribution as torch.multinomial, different random stream. */
int gnbv_multicategorical_sample(const float *logits, int batch, int n_logits, int n_heads, const int *head_dims /*[host]*/,
                                 const float *uniforms, int deterministic, int64_t *actions, float *log_prob, void *stream);

/* rsl_rl flavour of the PPO minibatch loss (rsl_rl/algorithms/ppo.py:160-180): scalar part of
 *   surrogate  = mean(max(-A r, -A clamp(r, 1 - c, 1 + c))),  r = exp(log_prob - old_log_prob)
 *   value_loss = mean(max((v - R)^2, (tv + clamp(v - tv, -c, c) - R)^2))   (use_clipped_value_loss) or mean((R - v)^2)
 *   loss       = surrogate + value_loss_coef * value_loss - entropy_coef * mean(entropy)
 * for ANY action distribution: the caller evaluates log_prob / value / entropy of the minibatch with its own modules and
 * back-propagates the three gradient vectors this call returns (ties of max() split evenly, like torch.max).
 * All arrays [batch] fp32, device.  sums [2] (device, caller-zeroed): += value_loss, += surrogate (the reference's
 * running `mean_value_loss` / `mean_surrogate_loss`, read once per update() instead of two .item() per minibatch). */
int gnbv_ppo_loss_rsl(int batch, const float *log_prob, const float *old_log_prob, const float *advantages, const float *values,
                      const float *target_values, const float *returns, float clip_param, float value_loss_coef,
                      float entropy_coef, int use_clipped_value_loss, float *d_log_prob, float *d_values, float *d_entropy,
                      float *sums, void *stream);

/* torch.nn.utils.clip_grad_norm_(max_grad_norm) + torch.optim.Adam step over ONE flat fp32
 * buffer of n parameters (max_grad_norm <= 0: no clipping). grads is the SUM over ranks,
 * grad_scale = 1/world turns it into the mean (1.0 on one GPU).  kl_slot (may be NULL): sum over
 * ranks of approx_kl; *stop_flag becomes 1 (sticky) when kl_slot*grad_scale > 1.5*target_kl.
 * *step is incremented and the update applied unless *stop_flag != 0.
 * norm_out[0] = norm of the mean gradient, [1] = factor applied to `grads`. */
size_t gnbv_adam_workspace_bytes(void);
int gnbv_clip_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float max_grad_norm,
                        float lr, float beta1, float beta2, float eps, int64_t *step, int *stop_flag, float grad_scale,
                        const float *kl_slot, float target_kl, float *norm_out, void *workspace, size_t workspace_bytes,
                        void *stream);
/* Same, for a minibatch update that is replayed as a hipGraph (the reference's inner loop, ppo_grid_obs.py:199-287, walks
 * `rollout_buffer.get(batch_size)`): the Adam launch -- the last of a minibatch -- also copies row (*counter + 1) % table_rows of
 * `table` [table_rows][row_len] (the row numbers of every minibatch of this train() call) into `out`, the buffer all kernels
 * of the graph read their row numbers from, and stores the new *counter: no copy and no host work between two replays. */
int gnbv_clip_adam_step_rotate(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float max_grad_norm,
                               float lr, float beta1, float beta2, float eps, int64_t *step, int *stop_flag, float grad_scale,
                               const float *kl_slot, float target_kl, float *norm_out, void *workspace, size_t workspace_bytes,
                               const int64_t *table, int table_rows, int row_len, int64_t *out, int *counter, void *stream);
/* Both of the above are this call with the optional parts left out.  `table` .. `counter`: the row rotation of
 * gnbv_clip_adam_step_rotate (table == NULL: none).  sq_lo .. sq_parts: the squared sum of the gradient slice [sq_lo, sq_hi) was
 * already taken by its producer (gnbv_linear_bwd_dw_sq) and is read from `sq_partial` instead of from the gradient itself;
 * sq_partial == NULL: the whole gradient is summed here.  Two launches: the squared-norm partial sums (+ step counter / KL
 * decision), then the update, every workgroup of which evaluates the clip coefficient from the partial sums in one fixed order. */
typedef struct GnbvAdamStep {
    float *params; const float *grads; float *exp_avg, *exp_avg_sq; int64_t n;
    float max_grad_norm, lr, beta1, beta2, eps;
    int64_t *step; int *stop_flag; float grad_scale; const float *kl_slot; float target_kl;
    float *norm_out; void *workspace; size_t workspace_bytes;
    const int64_t *table; int table_rows, row_len; int64_t *out; int *counter;
    int64_t sq_lo, sq_hi; const double *sq_partial; int sq_parts;
    const GnbvPpoLoss *loss_finish;     /* NULL, or [host] the loss arguments of this minibatch with defer_stats = 1: an extra workgroup
                                           of the norm launch adds up its per-sample terms (statistics row, KL stop decision, then the
                                           step counter) -- same stop_flag as this call's */
    int64_t upd_skip_lo, upd_skip_hi;   /* parameters [upd_skip_lo, upd_skip_hi) are NOT updated by this call (their gradient still
                                           counts for the norm through sq_partial): a slice whose update is sharded over the
                                           data-parallel replicas, gnbv_adam_shard_step */
} GnbvAdamStep;
int gnbv_clip_adam_step_ex(const GnbvAdamStep *a /*[host]*/, void *stream);
/* sum(grads[0 .. n)^2) as gnbv_sq_partials_count() fp64 partial sums in one fixed order -> partial [device].  The sharded data-parallel
 * update: a rank squares the shard of the reduced gradient it owns, the partial sums are summed over the ranks (an all-reduce of
 * 2 KB) and enter the clip factor of gnbv_clip_adam_step_ex through GnbvAdamStep.sq_partial / sq_parts. */
int gnbv_sq_partials_count(void);
int gnbv_sq_partials(const float *grads, int64_t n, double *partial /*[gnbv_sq_partials_count()]*/, void *stream);
/* Adam on a shard of n parameters with the clip factor norm_out[1] that gnbv_clip_adam_step_ex of the SAME optimizer step left
 * behind (same step counter and stop flag, neither is modified). */
int gnbv_adam_shard_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, const float *norm_out,
                         float lr, float beta1, float beta2, float eps, const int64_t *step, const int *stop_flag, void *stream);


/* ------------------------------------------------------------------------- */
/* 8f.3  evaluation metric: reconstruction accuracy of Env_Eval_GenNBV          */
/*       (gennbv/env/env_eval_gennbv.py:253-262): pytorch3d.loss.chamfer_distance */
/*       (x[None], y[None])[0] with pytorch3d 0.7.8 defaults = mean_i min_j |x_i-y_j|^2 */
/*       + mean_j min_i |x_i-y_j|^2 (squared distances).  x [n,3], y [m,3] fp32.  */
/* ------------------------------------------------------------------------- */
size_t gnbv_chamfer_workspace_bytes(int n, int m);
int gnbv_chamfer_distance(const float *x, int n, const float *y, int m, float *out /*[1]*/, void *workspace,
                          size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GENNBV_HIP_H */
