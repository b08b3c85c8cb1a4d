// envstep.hip -- per-environment bookkeeping of one Env_Train_GenNBV step on MI355X.
//
// Replaces the Python / tiny-torch-op glue around the voxel update:
//   step()                 gennbv/env/env_train_gennbv.py:246-264  (clip, forced init action, poses)
//   update_obs_buf         :273-275   (pose / gray-frame history)
//   compute_reward         gennbv/env/env_train_base.py:377-398, _reward_* env_train_gennbv.py:535-556
//   check_termination      env_train_gennbv.py:438-457
//   reset_idx              :377-436   (history refill, counters; grid zeroing is folded into the
//                                      next gnbv_update_occ_grid through reset_mask)
//   get_step_return / flatten_observations  :359-366, wrapper :27-56 (obs written in place:
//                                      [state | grid | state_rgb], row stride = D_obs)
//   update_extra_episode_info  env_train_base.py:629-639 (episode reward/length ring buffers,
//                                      kept on the device: no .cpu() per step)
// None of it is heavy: N x a few hundred floats.  The point is zero host synchronisation
// and four launches instead of ~60 torch ops + N device->host reads per step.
#include "common.h"
#include "../../include/gennbv_hip.h"

// ---------------------------------------------------------------------------
// pre-step: actions -> clipped actions, poses; episode_length_buf += 1
// ---------------------------------------------------------------------------
// action a of env e after step()'s clip and forced init action (env_train_gennbv.py:249-253), and the pose it means:
// poses = action * action_unit + clip_pose_low (env_train_base.py:665-667): int64 -> fp32, two roundings
__device__ __forceinline__ int64_t env_action(const int64_t *__restrict__ actions_in, const GnbvLattice &lat, bool fresh, int e, int a)
{
    int64_t v = actions_in[(size_t)e * 6 + a];
    v = v < lat.clip_low[a] ? lat.clip_low[a] : (v > lat.clip_up[a] ? lat.clip_up[a] : v);
    return fresh ? lat.init_action[a] : v;
}
__device__ __forceinline__ float env_pose(const GnbvLattice &lat, int64_t v, int a) { return __fadd_rn(__fmul_rn((float)v, lat.action_unit[a]), lat.pose_low[a]); }

__global__ void k_env_pre_step(const int64_t *__restrict__ actions_in, GnbvLattice lat, int64_t *__restrict__ episode_length_buf,
                               int n, int64_t *__restrict__ actions_out, float *__restrict__ poses_out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const bool fresh = episode_length_buf[e] == 0;  // env_train_gennbv.py:249-253
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const int64_t v = env_action(actions_in, lat, fresh, e, a);
        actions_out[(size_t)e * 6 + a] = v;
        poses_out[(size_t)e * 6 + a] = env_pose(lat, v, a);
    }
    episode_length_buf[e] += 1;  // post_physics_step :337
}

// ---------------------------------------------------------------------------
// observation, state slice: pose history shift + append, written to the obs row.
// One 64-lane workgroup per env; history is [stack, 6] oldest -> newest.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_env_obs_state(float *__restrict__ pose_hist, const float *__restrict__ poses,
                                                      const uint8_t *__restrict__ reset_mask, GnbvLattice lat, int n, int stack,
                                                      float *__restrict__ obs, int64_t obs_row_stride)
{
    const int e = blockIdx.x;
    if (e >= n) return;
    const int len = stack * 6;
    float *hist = pose_hist + (size_t)e * len;
    float *out = obs + (size_t)e * obs_row_stride;
    const bool reset = reset_mask != nullptr && reset_mask[e] != 0;  // reset_idx refilled the deque with init_pose_buf
    // new[i] = old[i + 6] for i < len - 6 ; new[len-6 .. len) = pose
    constexpr int kMaxPerLane = 16;  // stack <= 170
    float reg[kMaxPerLane];
#pragma unroll
    for (int k = 0; k < kMaxPerLane; ++k) {
        const int i = threadIdx.x + k * 64;
        if (i < len) {
            const int src = i + 6;
            reg[k] = src < len ? (reset ? lat.init_pose[src % 6] : hist[src]) : poses[(size_t)e * 6 + (src - len)];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kMaxPerLane; ++k) {
        const int i = threadIdx.x + k * 64;
        if (i < len) {
            hist[i] = reg[k];
            out[i] = reg[k];
        }
    }
}

// ---------------------------------------------------------------------------
// observation, rgb slice: [older gray | newest gray]; newest = nearest-resized,
// grayscaled RGBA (env_train_base.py:517-520, parity unpinned -- torchvision).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void env_obs_rgb_body(const uint8_t *__restrict__ rgba, float *__restrict__ gray_prev, const uint8_t *__restrict__ reset_mask,
                                                 int n, int h, int w, int oh, int ow, float *__restrict__ obs_rgb, int64_t obs_row_stride, int first, int stride)
{
    const int per = oh * ow;
    const int total = n * per;
    const float sh = (float)h / (float)oh, sw = (float)w / (float)ow;
    for (int i = first; i < total; i += stride) {
        const int e = i / per, r = i - e * per, y = r / ow, x = r - y * ow;
        int sy = (int)floorf(__fmul_rn((float)y, sh)), sx = (int)floorf(__fmul_rn((float)x, sw));
        sy = min(sy, h - 1);
        sx = min(sx, w - 1);
        const uint8_t *p = rgba + (((size_t)e * h + sy) * w + sx) * 4;
        float v = __fmul_rn(0.2989f, (float)p[0]);
        v = __fadd_rn(v, __fmul_rn(0.587f, (float)p[1]));
        v = __fadd_rn(v, __fmul_rn(0.114f, (float)p[2]));
        v = (float)(uint8_t)v;
        const bool reset = reset_mask != nullptr && reset_mask[e] != 0;
        float *row = obs_rgb + (size_t)e * obs_row_stride;
        row[r] = reset ? 0.0f : gray_prev[i];
        row[per + r] = v;
        gray_prev[i] = v;
    }
}
__global__ void k_env_obs_rgb(const uint8_t *__restrict__ rgba, float *__restrict__ gray_prev, const uint8_t *__restrict__ reset_mask,
                              int n, int h, int w, int oh, int ow, float *__restrict__ obs_rgb, int64_t obs_row_stride)
{
    env_obs_rgb_body(rgba, gray_prev, reset_mask, n, h, w, oh, ow, obs_rgb, obs_row_stride, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// ---------------------------------------------------------------------------
// pre-step + both observation slices as ONE launch (round 5: three dependent launches of 7-10 us each in front of every voxel update).
// Blocks [0, n): env e -- lanes 0-5 clip / force the action and write it with its pose (lane 0 counts the step), every lane shifts the
// pose history with the new pose computed in place (the same two roundings); blocks [n, n + rgb_blocks): the gray frames.
// 256 threads per block: an env block uses its first 64.  Same arithmetic as the three kernels: bit-identical outputs.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_env_observe(const int64_t *__restrict__ actions_in, GnbvLattice lat, int64_t *__restrict__ episode_length_buf, int n,
                                                    int64_t *__restrict__ actions_out, float *__restrict__ poses_out, float *__restrict__ pose_hist,
                                                    const uint8_t *__restrict__ reset_mask, int stack, float *__restrict__ obs, int64_t obs_row_stride,
                                                    const uint8_t *__restrict__ rgba, float *__restrict__ gray_prev, int h, int w, int oh, int ow,
                                                    float *__restrict__ obs_rgb, int rgb_blocks)
{
    if ((int)blockIdx.x >= n) {
        env_obs_rgb_body(rgba, gray_prev, reset_mask, n, h, w, oh, ow, obs_rgb, obs_row_stride, ((int)blockIdx.x - n) * 256 + (int)threadIdx.x, rgb_blocks * 256);
        return;
    }
    if (threadIdx.x >= 64) return;
    const int e = blockIdx.x;
    const bool fresh = episode_length_buf[e] == 0;  // (read by every lane BEFORE lane 0 counts the step: same wave, program order)
    const int len = stack * 6;
    float *hist = pose_hist + (size_t)e * len;
    float *out = obs + (size_t)e * obs_row_stride;
    const bool reset = reset_mask != nullptr && reset_mask[e] != 0;
    constexpr int kMaxPerLane = 16;  // stack <= 170
    float reg[kMaxPerLane];
#pragma unroll
    for (int k = 0; k < kMaxPerLane; ++k) {
        const int i = threadIdx.x + k * 64;
        if (i < len) {
            const int src = i + 6;
            reg[k] = src < len ? (reset ? lat.init_pose[src % 6] : hist[src]) : env_pose(lat, env_action(actions_in, lat, fresh, e, src - len), src - len);
        }
    }
    if (threadIdx.x < 6) {
        const int64_t v = env_action(actions_in, lat, fresh, e, threadIdx.x);
        actions_out[(size_t)e * 6 + threadIdx.x] = v;
        poses_out[(size_t)e * 6 + threadIdx.x] = env_pose(lat, v, threadIdx.x);
    }
    __builtin_amdgcn_wave_barrier();
    if (threadIdx.x == 0) episode_length_buf[e] += 1;  // post_physics_step :337
#pragma unroll
    for (int k = 0; k < kMaxPerLane; ++k) {
        const int i = threadIdx.x + k * 64;
        if (i < len) {
            hist[i] = reg[k];
            out[i] = reg[k];
        }
    }
}

// ---------------------------------------------------------------------------
// post-step: rewards, termination, reset bookkeeping, episode statistics.
// ONE workgroup (cross-env "any reset" + ordered ring-buffer appends need a scan).
// ---------------------------------------------------------------------------
constexpr int kPostThreads = 1024;

__global__ __launch_bounds__(kPostThreads) void k_env_post_step(GnbvEnvPost a)
{
    __shared__ int s_wave[kPostThreads / kWave + 1];
    __shared__ int s_any;
    __shared__ double s_sum[3];  // sums of episode_sums[name] over the envs that reset at this step
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
    if (tid == 0) s_any = 0;
    if (tid < 3) s_sum[tid] = 0.0;
    __syncthreads();
    int base_count = 0;  // dones seen in earlier tiles (ring-buffer order = env order)
    for (int e0 = 0; e0 < a.n; e0 += kPostThreads) {
        const int e = e0 + tid;
        const bool live = e < a.n;
        bool reset = false, time_out = false;
        float rew = 0.0f, cur_sum = 0.0f, cur_len = 0.0f;
        if (live) {
            const int64_t len = a.episode_length_buf[e];
            // _reward_surface_coverage :535-539 -- count is exact in fp32 (< 2^24)
            const float ratio = __fdiv_rn((float)a.coverage_count[e], a.num_valid_voxel_gt[e]);
            const float r_cov = __fmul_rn(__fsub_rn(ratio, a.prev_ratio[e]), a.scale_cov);
            rew = __fadd_rn(0.0f, r_cov);
            // _reward_short_path :541-545
            int64_t extra = len - 30;
            extra = extra < 0 ? 0 : (extra > 2 ? 2 : extra);
            const float r_short = __fmul_rn((float)(-extra), a.scale_short);
            rew = __fadd_rn(rew, r_short);
            if (a.only_positive) rew = rew < 0.0f ? 0.0f : rew;  // torch.clip(min=0.)
            // check_termination :438-457 (no contact forces in a replay feed)
            time_out = len >= a.max_episode_length;
            reset = time_out || (ratio > a.coverage_threshold);
            const float r_term = __fmul_rn((reset && !time_out) ? 1.0f : 0.0f, a.scale_term);
            rew = __fadd_rn(rew, r_term);
            a.rewards[e] = rew;
            a.dones[e] = reset ? 1 : 0;
            a.coverage_ratio[e] = ratio;
            // episode_sums += rew (base:387,398); reset_idx logs their mean over the reset envs and zeroes them (:425-427)
            const float s0 = __fadd_rn(a.episode_sums[e], r_cov);
            const float s1 = __fadd_rn(a.episode_sums[a.n + e], r_short);
            const float s2 = __fadd_rn(a.episode_sums[2 * a.n + e], r_term);
            a.episode_sums[e] = reset ? 0.0f : s0;
            a.episode_sums[a.n + e] = reset ? 0.0f : s1;
            a.episode_sums[2 * a.n + e] = reset ? 0.0f : s2;
            if (reset) {
                atomicAdd(&s_sum[0], (double)s0);
                atomicAdd(&s_sum[1], (double)s1);
                atomicAdd(&s_sum[2], (double)s2);
            }
            // reset_idx :377-436
            a.prev_ratio[e] = reset ? 0.0f : ratio;
            a.reset_mask[e] = reset ? 1 : 0;
            if (reset) a.episode_length_buf[e] = 0;
            a.step_time_out[e] = time_out ? 1 : 0;
            // update_extra_episode_info (env_train_base.py:629-639)
            cur_sum = __fadd_rn(a.cur_reward_sum[e], rew);
            cur_len = __fadd_rn(a.cur_episode_length[e], 1.0f);
            a.cur_reward_sum[e] = reset ? 0.0f : cur_sum;
            a.cur_episode_length[e] = reset ? 0.0f : cur_len;
        }
        // ordered append of finished episodes to the 100-deep ring buffers
        const int flag = (live && reset) ? 1 : 0;
        const int incl = wave_inclusive_scan(flag);
        if (lane == kWave - 1) s_wave[wv] = incl;
        __syncthreads();
        if (wv == 0) {
            int v = lane < kPostThreads / kWave ? s_wave[lane] : 0;
            const int sc = wave_inclusive_scan(v);
            if (lane < kPostThreads / kWave) s_wave[lane] = sc - v;
            if (lane == kPostThreads / kWave - 1) s_wave[kPostThreads / kWave] = sc;
        }
        __syncthreads();
        const int tile_total = s_wave[kPostThreads / kWave];
        if (flag) {
            const int64_t pos = a.ring_state[0] + base_count + s_wave[wv] + incl - 1;
            a.ring_reward[pos % a.ring_len] = cur_sum;
            a.ring_length[pos % a.ring_len] = cur_len;
            atomicOr(&s_any, 1);
        }
        base_count += tile_total;
        __syncthreads();
    }
    // the ring's entries (this step's appends included: stored above, in front of the tile loop's last barrier) into LDS by ALL threads, in
    // deque order -- thread 0 then adds them in that order from LDS instead of walking up to ring_len pairs of dependent global loads by
    // itself (the same additions in the same order: the same fp64 bits; 100 entries were ~100 round trips of this single workgroup)
    __shared__ float s_ring[2][kPostThreads];
    const int64_t total_all = a.ring_state[0] + base_count;
    const int k_all = (int)(total_all < a.ring_len ? total_all : a.ring_len);
    const bool ring_lds = a.episode_info != nullptr && k_all <= kPostThreads;
    if (ring_lds && tid < k_all) {
        const int64_t pos = total_all - k_all + tid;
        s_ring[0][tid] = ((const volatile float *)a.ring_reward)[pos % a.ring_len];
        s_ring[1][tid] = ((const volatile float *)a.ring_length)[pos % a.ring_len];
    }
    __syncthreads();
    if (tid == 0) {
        const int64_t total = a.ring_state[0] + base_count;  // total finished episodes so far
        a.ring_state[0] = total;
        if (a.episode_info) {  // extras["episode"] as it stands after this step
            if (base_count > 0) {  // some env reset: the reference creates a NEW dict (reset_idx :424-427)
                a.episode_state[0] += 1.0;
                for (int k = 0; k < 3; ++k)
                    a.episode_state[1 + k] = (double)__fdiv_rn((float)(s_sum[k] / (double)base_count), a.max_episode_length_s);
            }
            const int k = (int)(total < a.ring_len ? total : a.ring_len);
            double sr = 0.0, sl = 0.0;
            if (ring_lds) {
                for (int i = 0; i < k; ++i) {  // deque order: oldest -> newest
                    sr += (double)s_ring[0][i];
                    sl += (double)s_ring[1][i];
                }
            } else {
                for (int i = 0; i < k; ++i) {
                    const int64_t pos = total - k + i;
                    sr += (double)((const volatile float *)a.ring_reward)[pos % a.ring_len];
                    sl += (double)((const volatile float *)a.ring_length)[pos % a.ring_len];
                }
            }
            a.episode_info[0] = a.episode_state[0];
            a.episode_info[1] = k ? sr / k : 0.0;
            a.episode_info[2] = k ? sl / k : 0.0;
            for (int j = 0; j < 3; ++j) a.episode_info[3 + j] = a.episode_state[1 + j];
        }
    }
    // infos["time_outs"]: refreshed only on steps where some env resets (reference quirk,
    // reset_idx returns early on an empty id list :390-391)
    const bool any = s_any != 0;
    for (int e = tid; e < a.n; e += kPostThreads)
        if (any) a.extras_time_outs[e] = a.step_time_out[e];
}

// ===========================================================================
// C-ABI
// ===========================================================================
GNBV_API int gnbv_env_pre_step(const int64_t *actions_in, const GnbvLattice *lattice, int64_t *episode_length_buf, int n,
                               int64_t *actions_out, float *poses_out, void *stream)
{
    GNBV_CHECK_ARG(actions_in && lattice && episode_length_buf && actions_out && poses_out && n > 0);
    hipLaunchKernelGGL(k_env_pre_step, dim3((n + 255) / 256), dim3(256), 0, gnbv_stream(stream), actions_in, *lattice,
                       episode_length_buf, n, actions_out, poses_out);
    return gnbv_launch_status();
}

GNBV_API int gnbv_env_obs_state(float *pose_hist, const float *poses, const uint8_t *reset_mask, const GnbvLattice *lattice, int n,
                                int stack, float *obs, int64_t obs_row_stride, void *stream)
{
    GNBV_CHECK_ARG(pose_hist && poses && lattice && obs && n > 0 && stack > 0 && stack * 6 <= 16 * 64);
    GNBV_CHECK_ARG(obs_row_stride >= (int64_t)stack * 6);
    hipLaunchKernelGGL(k_env_obs_state, dim3(n), dim3(64), 0, gnbv_stream(stream), pose_hist, poses, reset_mask, *lattice, n,
                       stack, obs, obs_row_stride);
    return gnbv_launch_status();
}

GNBV_API int gnbv_env_obs_rgb(const uint8_t *rgba, float *gray_prev, const uint8_t *reset_mask, int n, int h, int w, int oh, int ow,
                              float *obs_rgb, int64_t obs_row_stride, void *stream)
{
    GNBV_CHECK_ARG(rgba && gray_prev && obs_rgb && n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0);
    GNBV_CHECK_ARG(obs_row_stride >= (int64_t)2 * oh * ow);
    int grid = (n * oh * ow + 255) / 256;
    grid = grid > 2048 ? 2048 : grid;
    hipLaunchKernelGGL(k_env_obs_rgb, dim3(grid), dim3(256), 0, gnbv_stream(stream), rgba, gray_prev, reset_mask, n, h, w, oh, ow,
                       obs_rgb, obs_row_stride);
    return gnbv_launch_status();
}

GNBV_API int gnbv_env_observe(const int64_t *actions_in, const GnbvLattice *lattice, int64_t *episode_length_buf, int n, int64_t *actions_out,
                              float *poses_out, float *pose_hist, const uint8_t *reset_mask, int stack, float *obs, int64_t obs_row_stride,
                              const uint8_t *rgba, float *gray_prev, int h, int w, int oh, int ow, float *obs_rgb, void *stream)
{
    GNBV_CHECK_ARG(actions_in && lattice && episode_length_buf && actions_out && poses_out && n > 0);
    GNBV_CHECK_ARG(pose_hist && obs && stack > 0 && stack * 6 <= 16 * 64 && obs_row_stride >= (int64_t)stack * 6);
    GNBV_CHECK_ARG(rgba && gray_prev && obs_rgb && h > 0 && w > 0 && oh > 0 && ow > 0 && obs_row_stride >= (int64_t)2 * oh * ow);
    int rgb_blocks = (n * oh * ow + 255) / 256;
    rgb_blocks = rgb_blocks > 2048 ? 2048 : rgb_blocks;
    hipLaunchKernelGGL(k_env_observe, dim3(n + rgb_blocks), dim3(256), 0, gnbv_stream(stream), actions_in, *lattice, episode_length_buf, n, actions_out,
                       poses_out, pose_hist, reset_mask, stack, obs, obs_row_stride, rgba, gray_prev, h, w, oh, ow, obs_rgb, rgb_blocks);
    return gnbv_launch_status();
}

GNBV_API int gnbv_env_post_step(const GnbvEnvPost *args, void *stream)
{
    GNBV_CHECK_ARG(args && args->n > 0 && args->ring_len > 0);
    GNBV_CHECK_ARG(args->coverage_count && args->num_valid_voxel_gt && args->prev_ratio && args->episode_length_buf);
    GNBV_CHECK_ARG(args->rewards && args->dones && args->reset_mask && args->step_time_out && args->extras_time_outs);
    GNBV_CHECK_ARG(args->coverage_ratio && args->episode_sums && args->cur_reward_sum && args->cur_episode_length);
    GNBV_CHECK_ARG(args->ring_reward && args->ring_length && args->ring_state);
    GNBV_CHECK_ARG(!args->episode_info || (args->episode_state && args->max_episode_length_s > 0.0f));
    hipLaunchKernelGGL(k_env_post_step, dim3(1), dim3(kPostThreads), 0, gnbv_stream(stream), *args);
    return gnbv_launch_status();
}

// ---------------------------------------------------------------------------
// One launch for the tail of a rollout step (on_policy_algorithm_grid_obs.py:205-211 + buffers.py:676-704): the
// time-out bootstrap  rewards += gamma * squeeze(terminal_value * time_outs)  (same fp32 operation order: the product
// with the 0/1 mask, times gamma, plus the reward, each rounded) and rollout_buffer.add()'s five copies (actions int64 ->
// fp32, episode_starts bool -> u8) straight into row `step` of the buffer arrays.
// ---------------------------------------------------------------------------
__global__ void k_rollout_add(int n, int adim, const int64_t *__restrict__ actions, const float *__restrict__ rewards,
                              const uint8_t *__restrict__ time_outs, const float *__restrict__ terminal_value, int tv_stride, float gamma,
                              const uint8_t *__restrict__ episode_starts, const float *__restrict__ values, const float *__restrict__ log_probs,
                              float *__restrict__ buf_actions, float *__restrict__ buf_rewards, uint8_t *__restrict__ buf_starts,
                              float *__restrict__ buf_values, float *__restrict__ buf_log_probs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float boot = __fmul_rn(gamma, __fmul_rn(terminal_value[(size_t)i * tv_stride], time_outs[i] ? 1.0f : 0.0f));
    buf_rewards[i] = __fadd_rn(rewards[i], boot);
    buf_starts[i] = episode_starts[i] ? 1 : 0;
    buf_values[i] = values[i];
    buf_log_probs[i] = log_probs[i];
    for (int a = 0; a < adim; ++a) buf_actions[(size_t)i * adim + a] = (float)actions[(size_t)i * adim + a];
}

GNBV_API int gnbv_rollout_add(int n, int action_dim, const int64_t *actions, const float *rewards, const uint8_t *time_outs,
                              const float *terminal_value, int terminal_value_stride, float gamma, const uint8_t *episode_starts,
                              const float *values, const float *log_probs, float *buf_actions, float *buf_rewards,
                              uint8_t *buf_episode_starts, float *buf_values, float *buf_log_probs, void *stream)
{
    GNBV_CHECK_ARG(terminal_value_stride == 0 || terminal_value_stride == 1);
    GNBV_CHECK_ARG(n > 0 && action_dim > 0 && actions && rewards && time_outs && terminal_value && episode_starts && values && log_probs);
    GNBV_CHECK_ARG(buf_actions && buf_rewards && buf_episode_starts && buf_values && buf_log_probs);
    hipLaunchKernelGGL(k_rollout_add, dim3((n + 255) / 256), dim3(256), 0, gnbv_stream(stream), n, action_dim, actions, rewards, time_outs,
                       terminal_value, terminal_value_stride, gamma, episode_starts, values, log_probs, buf_actions, buf_rewards, buf_episode_starts, buf_values,
                       buf_log_probs);
    return gnbv_launch_status();
}

