// gae.hip -- reverse GAE scan over the rollout buffer on MI355X (gfx950).
//
// Replaces stable_baselines3/common/buffers.py:706-724
// (TensorRolloutBuffer_Grid_Obs.compute_returns_and_advantage: T sequential torch
// steps of ~8 tiny launches each) and rsl_rl/storage/rollout_storage.py:130-142
// with ONE launch: a workgroup stages a [time-chunk x 32-env] tile of
// rewards / values / masks in LDS with coalesced loads from all 256 lanes, 32
// lanes run the strictly sequential recurrence out of LDS, and all lanes write
// advantages / returns back coalesced.  The recurrence is independent per env,
// so envs shard across workgroups (and across GPUs) with no exchange.
//
// Bit-exactness: torch evaluates every elementwise op of the reference scan as
// its own fp32 kernel, so the op order below uses explicit _rn intrinsics (no FMA
// contraction) in exactly the reference's association order; gamma and
// gamma*lambda are rounded to fp32 the way a Python scalar is when it multiplies
// an fp32 tensor.
#include "common.h"
#include "../../include/gennbv_hip.h"

constexpr int kEnvTile = 32;
constexpr int kTimeChunk = 128;
constexpr int kGaeThreads = 256;

template <bool RSL>
__global__ __launch_bounds__(kGaeThreads) void k_gae(
    const float *__restrict__ rewards, const float *__restrict__ values, const uint8_t *__restrict__ masks,
    const float *__restrict__ last_values, const uint8_t *__restrict__ last_dones, int t_steps, int n, float gamma,
    float gamma_lambda /* SB3: fl(gamma*lambda); rsl: lambda */, float *__restrict__ out_a, float *__restrict__ out_b)
{
    // s_val has one extra row: values[t+1] of the chunk's last row
    __shared__ float s_rew[kTimeChunk][kEnvTile];
    __shared__ float s_val[kTimeChunk + 1][kEnvTile];
    __shared__ float s_nnt[kTimeChunk][kEnvTile];  // next-non-terminal factor of row t
    const int e0 = blockIdx.x * kEnvTile;
    const int ne = min(kEnvTile, n - e0);
    const int tid = threadIdx.x, col = tid & (kEnvTile - 1), row0 = tid / kEnvTile;
    constexpr int kRowsPerPass = kGaeThreads / kEnvTile;
    float carry = 0.0f;  // last_gae_lam / advantage of row t+1 (lanes < ne of wave 0)

    for (int t_hi = t_steps; t_hi > 0; t_hi -= kTimeChunk) {
        const int t_lo = max(0, t_hi - kTimeChunk);
        const int rows = t_hi - t_lo;
        __syncthreads();  // previous chunk fully consumed
        for (int r = row0; r <= rows; r += kRowsPerPass) {
            if (col >= ne) continue;
            const int t = t_lo + r;
            if (r < rows) {
                const size_t i = (size_t)t * n + e0 + col;
                s_rew[r][col] = rewards[i];
                s_val[r][col] = values[i];
                float nnt;
                if (RSL) {
                    nnt = __fsub_rn(1.0f, (float)masks[i]);  // 1 - dones[t]
                } else {
                    // 1 - episode_starts[t+1], or 1 - dones for the last row
                    const uint8_t m = (t == t_steps - 1) ? last_dones[e0 + col] : masks[i + n];
                    nnt = __fsub_rn(1.0f, (float)m);
                }
                s_nnt[r][col] = nnt;
            } else {
                s_val[r][col] = (t == t_steps) ? last_values[e0 + col] : values[(size_t)t * n + e0 + col];
            }
        }
        __syncthreads();
        if (tid < ne) {
            float last = carry;
            for (int r = rows - 1; r >= 0; --r) {
                const float nv = s_val[r + 1][tid], v = s_val[r][tid], nnt = s_nnt[r][tid], rew = s_rew[r][tid];
                float delta, nxt;
                if (RSL) {
                    const float a1 = __fmul_rn(nnt, gamma);
                    delta = __fsub_rn(__fadd_rn(rew, __fmul_rn(a1, nv)), v);
                    nxt = __fadd_rn(delta, __fmul_rn(__fmul_rn(a1, gamma_lambda), last));
                } else {
                    delta = __fsub_rn(__fadd_rn(rew, __fmul_rn(__fmul_rn(gamma, nv), nnt)), v);
                    nxt = __fadd_rn(delta, __fmul_rn(__fmul_rn(gamma_lambda, nnt), last));
                }
                last = nxt;
                s_rew[r][tid] = nxt;  // reuse the tile: advantage of row r
            }
            carry = last;
        }
        __syncthreads();
        for (int r = row0; r < rows; r += kRowsPerPass) {
            if (col >= ne) continue;
            const size_t i = (size_t)(t_lo + r) * n + e0 + col;
            const float adv = s_rew[r][col], v = s_val[r][col];
            if (RSL) {
                const float ret = __fadd_rn(adv, v);
                out_a[i] = ret;                   // returns
                out_b[i] = __fsub_rn(ret, v);     // advantages = returns - values
            } else {
                out_a[i] = adv;                   // advantages
                out_b[i] = __fadd_rn(adv, v);     // returns
            }
        }
    }
}

GNBV_API int gnbv_gae_sb3(const float *rewards, const float *values, const uint8_t *episode_starts, const float *last_values,
                          const uint8_t *dones, int t_steps, int n, double gamma, double gae_lambda, float *advantages,
                          float *returns, void *stream)
{
    GNBV_CHECK_ARG(rewards && values && episode_starts && last_values && dones && advantages && returns);
    GNBV_CHECK_ARG(t_steps > 0 && n > 0);
    hipLaunchKernelGGL(k_gae<false>, dim3((n + kEnvTile - 1) / kEnvTile), dim3(kGaeThreads), 0, gnbv_stream(stream), rewards,
                       values, episode_starts, last_values, dones, t_steps, n, (float)gamma, (float)(gamma * gae_lambda),
                       advantages, returns);
    return gnbv_launch_status();
}

GNBV_API int gnbv_gae_rsl(const float *rewards, const float *values, const uint8_t *dones, const float *last_values, int t_steps,
                          int n, double gamma, double lam, float *returns, float *advantages, void *stream)
{
    GNBV_CHECK_ARG(rewards && values && dones && last_values && returns && advantages);
    GNBV_CHECK_ARG(t_steps > 0 && n > 0);
    hipLaunchKernelGGL(k_gae<true>, dim3((n + kEnvTile - 1) / kEnvTile), dim3(kGaeThreads), 0, gnbv_stream(stream), rewards,
                       values, dones, last_values, (const uint8_t *)nullptr, t_steps, n, (float)gamma, (float)lam, returns,
                       advantages);
    return gnbv_launch_status();
}
