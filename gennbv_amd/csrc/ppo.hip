// ppo.hip -- the PPO minibatch update of stable_baselines3/ppo/ppo_grid_obs.py:196-275 on MI355X:
//   * k_gather_minibatch   the five per-sample gathers of buffers.py:753-762 in one launch
//   * k_ppo_fused          advantage normalisation (:214-216), MultiCategorical log-prob / entropy
//                          (distributions.py:321-332), clipped surrogate (:219-224), clipped value
//                          loss (:231-241), entropy loss (:245-249), loss = 10*pg + c_e*ent + c_v*vl
//                          (:253), approx-KL (:259-262) AND the analytic gradient of the loss with
//                          respect to the logits and the values -- one launch
//                          instead of ~100 tiny torch kernels forward + backward;
//                          sets the sticky early-stop flag (:264-268) on the device.
//   * k_grad_sqnorm / k_adam_flat   clip_grad_norm_(max_norm) + Adam(eps=1e-5) (:271-275) over ONE
//                          flat fp32 parameter / gradient buffer, masked by the stop flag.
// Nothing here synchronises with the host: the five .item()/.cpu() reads per minibatch of the
// reference (:227,228,242,251,261) become rows of a device-side statistics table.
#include "common.h"
#include "../../include/gennbv_hip.h"

constexpr int kLossThreads = 256;
constexpr int kMaxHeads = 8;

// ---------------------------------------------------------------------------
__global__ void k_gather_minibatch(const int64_t *__restrict__ rows, int batch, int act_dim, const float *__restrict__ actions,
                                   const float *__restrict__ values, const float *__restrict__ log_probs,
                                   const float *__restrict__ advantages, const float *__restrict__ returns,
                                   float *__restrict__ o_actions, float *__restrict__ o_values, float *__restrict__ o_log_probs,
                                   float *__restrict__ o_adv, float *__restrict__ o_ret)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const int64_t r = rows[i];
    for (int a = 0; a < act_dim; ++a) o_actions[(size_t)i * act_dim + a] = actions[(size_t)r * act_dim + a];
    o_values[i] = values[r];
    o_log_probs[i] = log_probs[r];
    o_adv[i] = advantages[r];
    o_ret[i] = returns[r];
}

// ---------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ float block_sum(float v, float *scratch /*[NT/64 + 1]*/)
{
    v = wave_reduce_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < NT / 64; ++i) t += scratch[i];
    return t;
}

// row of the rollout buffer behind minibatch sample i (fused gather, buffers.py:753-762) or i itself
__device__ __forceinline__ int64_t src_row(const GnbvPpoLoss &a, int i) { return a.rows ? a.rows[i] : (int64_t)i; }

// head statistics of one sample, computed by one wave (lanes stride over the categories)
struct HeadStats { float lse, ent; };
__device__ __forceinline__ HeadStats head_stats(const float *lg, int n, int lane)
{
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, lg[j]);
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    float se = 0.f;
    for (int j = lane; j < n; j += 64) se += expf(lg[j] - mx);
    for (int d = 32; d > 0; d >>= 1) se += __shfl_xor(se, d, 64);
    const float lse = mx + logf(se);
    float pe = 0.f;
    for (int j = lane; j < n; j += 64) {
        const float lp = lg[j] - lse;
        pe += expf(lp) * lp;
    }
    for (int d = 32; d > 0; d >>= 1) pe += __shfl_xor(pe, d, 64);
    return {lse, -pe};
}

// The same statistics from the head's logits in registers (x0 = logit lane, x1 = logit lane + 64; n <= 128): the same per-lane order of
// operations and the same shuffles as head_stats -- bit-identical --, without its three passes of loads per head (each pass of each of
// the six heads was a round trip of its own on the update's critical path: k_ppo_fused 16 us).
__device__ __forceinline__ HeadStats head_stats_regs(float x0, float x1, int n, int lane)
{
    const bool h0 = lane < n, h1 = lane + 64 < n;
    float mx = -INFINITY;
    if (h0) mx = fmaxf(mx, x0);
    if (h1) mx = fmaxf(mx, x1);
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    float se = 0.f;
    if (h0) se += expf(x0 - mx);
    if (h1) se += expf(x1 - mx);
    for (int d = 32; d > 0; d >>= 1) se += __shfl_xor(se, d, 64);
    const float lse = mx + logf(se);
    float pe = 0.f;
    if (h0) { const float lp = x0 - lse; pe += expf(lp) * lp; }
    if (h1) { const float lp = x1 - lse; pe += expf(lp) * lp; }
    for (int d = 32; d > 0; d >>= 1) pe += __shfl_xor(pe, d, 64);
    return {lse, -pe};
}

// The minibatch's logged scalars from the per-sample terms k_ppo_fused left in `terms` ([B][8]), added in a fixed order by ONE
// workgroup of kLossThreads threads; writes the statistics row, takes the KL early-stop decision (ppo_grid_obs.py:261-268) and --
// `step` != NULL: the caller is the optimizer's norm launch -- counts the optimizer step unless the update is masked.
__device__ void ppo_stats_finish(const GnbvPpoLoss &a, const float *__restrict__ terms, float *scratch /*LDS, kLossThreads / 64 + 1*/, int64_t *step)
{
    const int B = a.batch, tid = threadIdx.x;
    const float invB = 1.0f / (float)B;
    // (thread 0's two dependent words are requested before the sums, not after them: this workgroup is the longest of its launch)
    int stopped_before = 0;
    int64_t stats_row = 0;
    if (tid == 0) {
        stopped_before = a.stop_flag ? *a.stop_flag : 0;
        stats_row = *a.stats_row;
    }
    float pg = 0.f, vl = 0.f, en = 0.f, kl = 0.f, cf = 0.f;
    for (int j = tid; j < B; j += kLossThreads) {
        const volatile float *t = terms + (size_t)j * 8;
        pg += t[0]; vl += t[1]; en += t[2]; kl += t[3]; cf += t[4];
    }
    // the five block sums TOGETHER: the five wave reductions interleave and share one pair of barriers (five block_sum calls were ten
    // barriers and five exposed reduction chains in a row); per sum the same operations in the same order as block_sum: the same bits
    {
        __shared__ float five[5][kLossThreads / 64];
        pg = wave_reduce_sum(pg); vl = wave_reduce_sum(vl); en = wave_reduce_sum(en); kl = wave_reduce_sum(kl); cf = wave_reduce_sum(cf);
        __syncthreads();
        if ((tid & 63) == 0) {
            const int w = tid >> 6;
            five[0][w] = pg; five[1][w] = vl; five[2][w] = en; five[3][w] = kl; five[4][w] = cf;
        }
        __syncthreads();
        float t5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int i = 0; i < kLossThreads / 64; ++i) t5[q] += five[q][i];
        pg = t5[0] * invB; vl = t5[1] * invB; en = t5[2] * invB; kl = t5[3] * invB; cf = t5[4] * invB;
    }
    (void)scratch;
    if (tid == 0) {
        const float loss = pg * a.policy_scale + a.ent_coef * en + a.vf_coef * vl;
        float *row = a.stats + (size_t)stats_row * 8;
        row[0] = pg; row[1] = vl; row[2] = en; row[3] = kl; row[4] = cf; row[5] = loss;
        row[6] = stopped_before ? 0.f : 1.f;  // live row (the reference never ran this minibatch otherwise)
        row[7] = 0.f;
        *a.stats_row = stats_row + 1;
        if (a.kl_out) *a.kl_out = kl;  // data-parallel: the decision is taken on the global mean after the all-reduce
        else if (a.stop_flag && a.target_kl > 0.f && kl > 1.5f * a.target_kl) *a.stop_flag = 1;  // sticky (:264-268)
        if (step != nullptr && !(a.stop_flag != nullptr && *a.stop_flag != 0)) *step += 1;
    }
}

// ---------------------------------------------------------------------------
// ONE launch: a wave per sample computes the head statistics once and uses them for the log-prob, the entropy AND the
// logit gradient (three launches recomputed them twice and paid two extra launch latencies on the update's critical path:
// 32 us -> see profiles/r02_notes.md).  The only cross-sample quantities are the advantage mean / std -- functions of the
// INPUTS, which every wave evaluates for itself over the <= a few hundred advantages of the minibatch -- and the logged
// scalars, which the workgroup that finishes last adds up in a fixed order (per-sample terms in `scratch`).
// (A single workgroup walking all samples serially was measured at ~130 us.)
// ---------------------------------------------------------------------------
// NH > 0, REGS: the number of heads is a compile-time constant and every head has <= 128 categories (the reference's action lattice:
// six heads of 81, 81, 51, 1, 13, 13).  The per-head code then is ONE straight-line block -- no `h < n_heads` / `regs` / pointer branches
// between the heads --, so the scheduler interleaves the six heads' reductions (18 dependent lane exchanges each, ~60 cycles apiece: six
// chains one after the other were ~2.5 us of this launch, which sits on the critical path of every minibatch).  Same operations per
// head in the same order: the same bits.  NH = 0: heads and their sizes at run time (any lattice).
template <int NH, bool REGS>
__global__ __launch_bounds__(kLossThreads) void k_ppo_fused(GnbvPpoLoss a, float *__restrict__ terms /*[B][8]*/, int *__restrict__ counter)
{
    __shared__ float scratch[kLossThreads / 64 + 1];
    __shared__ int s_last;
    const int B = a.batch, tid = threadIdx.x, lane = tid & 63;
    const int i = blockIdx.x * (kLossThreads / 64) + (tid >> 6);
    const float invB = 1.0f / (float)B;
    const int n_heads = NH ? NH : a.n_heads;
    if (i < B) {
        const float *lg = a.logits + (size_t)i * a.n_logits;
        const int64_t r = src_row(a, i);
        // (round 5: the sample's scalars are requested HERE, with the logits, instead of after the head statistics -- one dependent
        // round trip less in a launch that is nothing but round trips)
        const float adv_raw = a.advantages[r], old_lp = a.old_log_prob[r], v = a.values[i], vo = a.old_values[r], ret = a.returns[r];
        const float norm_mean = (a.normalize_advantage && a.adv_norm) ? a.adv_norm[0] : 0.f;
        const float norm_inv = (a.normalize_advantage && a.adv_norm) ? a.adv_norm[1] : 1.f;
        // ---- head statistics, log-prob of the taken actions, entropy ----
        float lse[kMaxHeads], hent[kMaxHeads];
        int act[kMaxHeads];
        float logp = 0.f, ent = 0.f;
        // every operand first -- the sample's logits (two per lane and head: heads of <= 128 categories), its actions and the logits of
        // the taken actions' heads are then in flight together instead of head after head
        float x0[kMaxHeads], x1[kMaxHeads], xa[kMaxHeads];
        bool regs = true;  // (uniform)
        {
            int off = 0;
#pragma unroll
            for (int h = 0; h < kMaxHeads; ++h) {
                if (h < n_heads) {
                    const int n = a.head_dims[h];
                    regs = regs && n <= 128;
                    x0[h] = lg[off + min(lane, n - 1)];
                    x1[h] = lg[off + min(lane + 64, n - 1)];
                    act[h] = (int)a.actions[(size_t)r * n_heads + h];  // actions are stored as float (buffers.py:664)
                    off += n;
                }
            }
            if (REGS) regs = true;
            off = 0;
#pragma unroll
            for (int h = 0; h < kMaxHeads; ++h) {
                if (h < n_heads) {
                    // the taken action's logit: with the head in registers it is in lane act & 63 already (round 5: a lane exchange instead
                    // of a third dependent round trip rows -> actions -> logits[action])
                    if (REGS || regs) {
                        const float lo_ = __shfl(x0[h], act[h] & 63, 64), hi_ = __shfl(x1[h], act[h] & 63, 64);
                        xa[h] = act[h] < 64 ? lo_ : hi_;
                    } else {
                        xa[h] = lg[off + act[h]];
                    }
                    off += a.head_dims[h];
                }
            }
        }
        {
            int off = 0;
#pragma unroll
            for (int h = 0; h < kMaxHeads; ++h) {
                if (h < n_heads) {
                    const int n = a.head_dims[h];
                    const HeadStats hs = (REGS || regs) ? head_stats_regs(x0[h], x1[h], n, lane) : head_stats(lg + off, n, lane);
                    lse[h] = hs.lse; hent[h] = hs.ent;
                    logp += xa[h] - hs.lse;
                    ent += hs.ent;
                    off += n;
                }
            }
            // (the optional per-head outputs: stored behind the loop, so that no pointer test ends a head's basic block)
            if ((a.head_entropy || a.head_lse) && lane == 0) {
#pragma unroll
                for (int h = 0; h < kMaxHeads; ++h) {
                    if (h < n_heads) {
                        if (a.head_entropy) a.head_entropy[(size_t)i * n_heads + h] = hent[h];
                        if (a.head_lse) a.head_lse[(size_t)i * n_heads + h] = lse[h];
                    }
                }
            }
        }
        // ---- advantage normalisation: (A - mean) / (std_unbiased + 1e-8), the statistics of the (global) minibatch ----
        float mean = 0.f, inv_std = 1.f;
        if (a.normalize_advantage) {
            if (a.adv_norm) {
                mean = norm_mean; inv_std = norm_inv;
            } else {
                float s = 0.f;
                for (int j = lane; j < B; j += 64) s += a.advantages[src_row(a, j)];
                for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);  // (all lanes need the total)
                mean = s * invB;
                float q = 0.f;
                for (int j = lane; j < B; j += 64) {
                    const float d = a.advantages[src_row(a, j)] - mean;
                    q += d * d;
                }
                for (int d = 32; d > 0; d >>= 1) q += __shfl_xor(q, d, 64);
                inv_std = 1.0f / (sqrtf(q / (float)(B > 1 ? B - 1 : 1)) + 1e-8f);
            }
        }
        // ---- this sample's scalar terms (every lane computes them: no broadcast needed) ----
        const float adv = a.normalize_advantage ? (adv_raw - mean) * inv_std : adv_raw;
        const float log_ratio = logp - old_lp;
        const float ratio = expf(log_ratio);
        const float lo = 1.0f - a.clip_range, hi = 1.0f + a.clip_range;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float s1 = adv * ratio, s2 = adv * rc;
        // d min(s1, s2): the smaller operand takes the gradient, ties split evenly (torch.minimum)
        const float g1 = s1 < s2 ? 1.f : (s1 > s2 ? 0.f : 0.5f), g2 = 1.f - g1;
        const float inrange = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
        const float gl = -invB * a.policy_scale * adv * ratio * (g1 + g2 * inrange);
        float vp = v, dvp = 1.f;
        if (a.clip_range_vf > 0.f) {
            const float dv = v - vo;
            vp = vo + fminf(fmaxf(dv, -a.clip_range_vf), a.clip_range_vf);
            dvp = (dv >= -a.clip_range_vf && dv <= a.clip_range_vf) ? 1.f : 0.f;
        }
        const float err = vp - ret;
        if (lane == 0) {
            a.d_values[i] = a.vf_coef * 2.0f * invB * err * dvp;
            float *t = terms + (size_t)i * 8;
            t[0] = -fminf(s1, s2);                                  // policy-gradient term
            t[1] = err * err;                                       // value term
            t[2] = -ent;                                            // entropy term
            t[3] = (ratio - 1.0f) - log_ratio;                      // approx-KL term
            t[4] = fabsf(ratio - 1.0f) > a.clip_range ? 1.f : 0.f;  // clip fraction term
        }
        // ---- d logits = gl*(onehot - p) + (ent_coef/B) * p*(log p + H_head) ----
        float *dl = a.d_logits + (size_t)i * a.n_logits;
        int off = 0;
#pragma unroll
        for (int h = 0; h < kMaxHeads; ++h) {
            if (h < n_heads) {
                const int n = a.head_dims[h];
                if (REGS) {
                    // (two predicated steps instead of a loop over j = lane, lane + 64: straight-line across the heads)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int j = lane + 64 * half;
                        const float xv = half ? x1[h] : x0[h];
                        const float lp = xv - lse[h], p = expf(lp);
                        const float d = gl * ((j == act[h] ? 1.f : 0.f) - p) + a.ent_coef * invB * p * (lp + hent[h]);
                        if (j < n) dl[off + j] = d;
                    }
                } else {
                    for (int j = lane; j < n; j += 64) {
                        const float xv = regs ? (j == lane ? x0[h] : x1[h]) : lg[off + j];
                        const float lp = xv - lse[h], p = expf(lp);
                        dl[off + j] = gl * ((j == act[h] ? 1.f : 0.f) - p) + a.ent_coef * invB * p * (lp + hent[h]);
                    }
                }
                off += n;
            }
        }
    }
    if (a.defer_stats) return;  // (the caller adds the terms up later: ppo_stats_finish inside the optimizer's norm launch)
    // ---- the workgroup that finishes last adds the per-sample terms (fixed order: deterministic) ----
    __syncthreads();  // (every wave's stores have left the CU: s_waitcnt vmcnt(0) + barrier)
    if (tid == 0) {
        __threadfence();  // one release per workgroup, then the ticket
        s_last = atomicAdd(counter, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (tid == 0) *counter = 0;  // ready for the next call
    ppo_stats_finish(a, terms, scratch, nullptr);
}

// ---------------------------------------------------------------------------
// rsl_rl flavour (rsl_rl/algorithms/ppo.py:160-180): one workgroup, the minibatch's scalar loss and its gradient with
// respect to (log_prob, value, entropy) -- the distribution itself stays with the caller's modules.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kLossThreads) void k_ppo_loss_rsl(int B, const float *__restrict__ logp, const float *__restrict__ old_logp,
                                                              const float *__restrict__ adv, const float *__restrict__ v,
                                                              const float *__restrict__ tv, const float *__restrict__ ret, float clip,
                                                              float vcoef, float ecoef, int clipped_v, float *__restrict__ d_logp,
                                                              float *__restrict__ d_v, float *__restrict__ d_ent, float *__restrict__ sums)
{
    __shared__ float scratch[kLossThreads / 64 + 1];
    const float invB = 1.0f / (float)B;
    float sl = 0.f, vl = 0.f;
    for (int i = threadIdx.x; i < B; i += kLossThreads) {
        const float a = adv[i];
        const float ratio = expf(logp[i] - old_logp[i]);
        const float lo = 1.0f - clip, hi = 1.0f + clip;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float s1 = -a * ratio, s2 = -a * rc;
        sl += fmaxf(s1, s2);
        // d max(s1, s2): the larger operand takes the gradient, ties split evenly (torch.max of two tensors)
        const float g1 = s1 > s2 ? 1.f : (s1 < s2 ? 0.f : 0.5f), g2 = 1.f - g1;
        const float inrange = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
        d_logp[i] = invB * (-a) * ratio * (g1 + g2 * inrange);
        const float e1 = v[i] - ret[i];
        if (clipped_v) {
            const float dv = v[i] - tv[i];
            const float vc = tv[i] + fminf(fmaxf(dv, -clip), clip);
            const float e2 = vc - ret[i];
            const float l1 = e1 * e1, l2 = e2 * e2;
            vl += fmaxf(l1, l2);
            const float h1 = l1 > l2 ? 1.f : (l1 < l2 ? 0.f : 0.5f), h2 = 1.f - h1;
            const float vin = (dv >= -clip && dv <= clip) ? 1.f : 0.f;
            d_v[i] = vcoef * invB * 2.0f * (h1 * e1 + h2 * e2 * vin);
        } else {
            vl += e1 * e1;
            d_v[i] = vcoef * invB * 2.0f * e1;
        }
        d_ent[i] = -ecoef * invB;
    }
    sl = block_sum<kLossThreads>(sl, scratch) * invB;
    vl = block_sum<kLossThreads>(vl, scratch) * invB;
    if (threadIdx.x == 0 && sums) { sums[0] += vl; sums[1] += sl; }
}

// ---------------------------------------------------------------------------
// clip_grad_norm_ + Adam over a flat buffer
// ---------------------------------------------------------------------------
// sum g^2 over the flat gradient, WITHOUT the slice [lo, lo + gap) when gap > 0 (its producer took that sum: GnbvAdamStep.sq_*).
// Fixed grid -> fixed summation order.  Four 16-byte requests per lane are in flight before the first is consumed (the first
// version chained one request per iteration: 17 us for 58 MB).  Block 0 also carries the bookkeeping that used to sit in a
// single-workgroup finalize launch between this kernel and the update: the data-parallel KL decision and the optimizer step
// counter, both final before the Adam launch starts.
__global__ __launch_bounds__(256) void k_grad_sqnorm(const float *__restrict__ g, int64_t n_eff, int64_t lo, int64_t gap, double *__restrict__ partial,
                                                     float grad_scale, int64_t *step, int *stop_flag, const float *kl_slot, float target_kl,
                                                     int nblocks /*norm blocks; then, if launched: one block that folds `extra`, one that finishes the loss*/,
                                                     const double *__restrict__ extra, int nextra, int fin_block, GnbvPpoLoss fin,
                                                     float *__restrict__ hyper /*[2]: Adam's step size lr / (1 - beta1^t) and sqrt(1 - beta2^t)*/, float lr,
                                                     float beta1, float beta2)
{
    static_assert(kLossThreads == 256, "the finish block is one loss workgroup");
    // Adam's bias corrections (fp64 pow, like torch's scalar path) are functions of the step counter alone: evaluated ONCE, by the thread
    // that has just counted the step, instead of by every lane of the update launch (two fp64 pow per lane were a quarter of that
    // launch at 20^3, where it is 2 070 workgroups of prologue and one trip of work).  Same expressions: the same bits.
    auto bias_corrections = [&]() {
        const double t = (double)(*step);
        const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
        hyper[0] = (float)((double)lr / bc1);
        hyper[1] = (float)sqrt(bc2);
    };
    if (extra != nullptr && (int)blockIdx.x == nblocks) {
        // the producer's partial sums of the skipped slice (gnbv_linear_bwd_dw_sq: 844 of them) folded into ONE more partial here,
        // so that the update's workgroups re-add ~200 numbers each instead of ~1040 (that prologue cost the Adam launch 5-10 us)
        __shared__ double se[256];
        double acc = 0.0;
        for (int i = threadIdx.x; i < nextra; i += 256) acc += extra[i];
        se[threadIdx.x] = acc;
        __syncthreads();
        for (int d = 128; d > 0; d >>= 1) {
            if (threadIdx.x < d) se[threadIdx.x] += se[threadIdx.x + d];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[nblocks] = se[0];
        return;
    }
    if ((int)blockIdx.x == fin_block) {
        // (same launch, independent work) the loss kernel left its per-sample terms behind (GnbvPpoLoss.defer_stats): the logged
        // scalars, the KL stop decision and -- in the SAME thread, after that decision -- the optimizer step counter
        __shared__ float fscratch[kLossThreads / 64 + 1];
        ppo_stats_finish(fin, fin.scratch, fscratch, step);
        if (threadIdx.x == 0 && hyper != nullptr) bias_corrections();  // (thread 0 counted the step: program order)
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && step != nullptr && fin_block < 0) {
        // kl_slot holds the SUM of the ranks' approx-KL (it rode in front of the gradient in the all-reduce); sets the sticky
        // stop flag before this step's update
        if (stop_flag != nullptr && kl_slot != nullptr && target_kl > 0.f && (*kl_slot) * grad_scale > 1.5f * target_kl) *stop_flag = 1;
        if (!(stop_flag != nullptr && *stop_flag != 0)) *step += 1;
        if (hyper != nullptr) bias_corrections();
    }
    double acc = 0.0;
    const bool vec = (((uintptr_t)g & 15) == 0) && (gap == 0 || ((lo & 3) == 0 && (gap & 3) == 0));  // (no gap: `lo` = n marks nothing and need not be aligned)
    const int64_t n4 = vec ? n_eff / 4 : 0, lo4 = lo / 4, gap4 = gap / 4;
    const int64_t stride = (int64_t)nblocks * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t j = min(i0 + u * stride, n4 - 1);  // (clamped duplicate: unconditional request, masked below)
            j += j >= lo4 ? gap4 : 0;
            v[u] = reinterpret_cast<const float4 *>(g)[j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * stride < n4)
                acc += (double)v[u].x * (double)v[u].x + (double)v[u].y * (double)v[u].y + (double)v[u].z * (double)v[u].z + (double)v[u].w * (double)v[u].w;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_eff; i += stride) {
        const double v = (double)g[i + (i >= lo ? gap : 0)];
        acc += v * v;
    }
    __shared__ double s[256];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

// clip coefficient = min(1, max_norm / (total_norm + 1e-6))  (torch.nn.utils.clip_grad_norm_), evaluated by EVERY workgroup of
// the update from the partial sums (same order of additions everywhere: the same bits) instead of by a launch of its own.
__global__ void k_adam_flat(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                            int64_t n, const double *__restrict__ partial, int nparts, const double *__restrict__ extra, int nextra,
                            float max_norm, float grad_scale, float *__restrict__ norm_out /*[2]: norm, applied factor*/,
                            const float *__restrict__ coef_in /*NULL, or a [2] an earlier launch of this step wrote: use its factor*/,
                            int64_t skip_lo, int64_t skip_hi /*elements [skip_lo, skip_hi) are left alone (another rank's shard)*/,
                            const int *__restrict__ stop_flag,
                            const int64_t *__restrict__ step /*already incremented*/, float lr, float beta1, float beta2, float eps,
                            const int64_t *__restrict__ rot_table = nullptr, int rot_rows = 0, int rot_len = 0, int64_t *__restrict__ rot_out = nullptr,
                            int *__restrict__ rot_counter = nullptr, const float *__restrict__ hyper = nullptr /*[2] from the norm launch, or NULL*/)
{
    // (replayed minibatch graphs: the LAST launch of a minibatch leaves the next minibatch's row numbers in the buffer every kernel
    // of the graph reads them from -- no copy node, no host work between two replays.  Also when the update itself is masked.)
    if (rot_table != nullptr && blockIdx.x == 0) {
        const int c = (*rot_counter + 1) % rot_rows;
        for (int i = threadIdx.x; i < rot_len; i += blockDim.x) rot_out[i] = rot_table[(int64_t)c * rot_len + i];
        __syncthreads();
        if (threadIdx.x == 0) *rot_counter = c;
    }
    const bool stopped = stop_flag != nullptr && *stop_flag != 0;
    if (stopped) return;
    __shared__ double sh[256];
    float coef;
    if (coef_in != nullptr) {
        coef = coef_in[1];
    } else {
        double acc = 0.0;
        for (int i = threadIdx.x; i < nparts; i += 256) acc += partial[i];
        for (int i = threadIdx.x; i < nextra; i += 256) acc += extra[i];
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int d = 128; d > 0; d >>= 1) {
            if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
            __syncthreads();
        }
        const float norm = (float)sqrt(sh[0]) * grad_scale;  // norm of the averaged gradient
        float cf = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.0f;
        cf = cf > 1.0f ? 1.0f : cf;
        coef = cf * grad_scale;  // factor applied to the raw (summed) gradient
        if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out != nullptr) {
            norm_out[0] = norm;
            norm_out[1] = coef;
        }
    }
    // bias corrections in double like torch's scalar path (1 - beta**step evaluated in Python floats)
    float step_size, bc2_sqrt;
    if (hyper != nullptr) {  // (k_grad_sqnorm evaluated them once for this step)
        step_size = hyper[0];
        bc2_sqrt = hyper[1];
    } else {
        const double t = (double)(*step);
        const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
        step_size = (float)((double)lr / bc1);
        bc2_sqrt = (float)sqrt(bc2);
    }
    const int64_t gap = skip_hi > skip_lo ? skip_hi - skip_lo : 0;  // (the grid covers the n - gap live elements only)
    const int64_t live = n - gap, stride = (int64_t)gridDim.x * blockDim.x;
    auto one = [&](float &pp, const float gg, float &mm, float &vv) {
        const float gi = gg * coef;
        const float mi = mm + (gi - mm) * (1.0f - beta1);  // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = vv * beta2 + (1.0f - beta2) * gi * gi;  // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
        mm = mi;
        vv = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pp = pp - step_size * (mi / denom);
    };
    // 16-byte requests (four elements per lane and trip, the same expressions per element: the same bits) where the four streams are
    // aligned and the skipped slice starts and ends on a multiple of four (by itself no faster than 4-byte requests: measured; what
    // pays is the cache policy below)
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && (gap == 0 || (((skip_lo | gap) & 3) == 0));
    const int64_t n4 = vec ? live / 4 : 0;
    for (int64_t j4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j4 < n4; j4 += stride) {
        const int64_t j = 4 * j4, i = j < skip_lo ? j : j + gap;
        // (m, v and g are streamed -- their next use is a whole minibatch away, behind ~2 GB of other traffic: non-temporal both ways;
        // the parameters are read again by the next forward: plain.  -16 us per minibatch, csrc/conv_split.h ld4_nt)
        typedef float f4v __attribute__((ext_vector_type(4)));
        auto ldnt = [](const float *q) { const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(q)); return make_float4(t[0], t[1], t[2], t[3]); };
        auto stnt = [](float *q, const float4 &t) { __builtin_nontemporal_store((f4v){t.x, t.y, t.z, t.w}, reinterpret_cast<f4v *>(q)); };
        float4 p4 = *reinterpret_cast<const float4 *>(p + i), m4 = ldnt(m + i), v4 = ldnt(v + i);
        const float4 g4 = ldnt(g + i);
        one(p4.x, g4.x, m4.x, v4.x);
        one(p4.y, g4.y, m4.y, v4.y);
        one(p4.z, g4.z, m4.z, v4.z);
        one(p4.w, g4.w, m4.w, v4.w);
        stnt(m + i, m4);
        stnt(v + i, v4);
        *reinterpret_cast<float4 *>(p + i) = p4;
    }
    for (int64_t j = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < live; j += stride) {
        const int64_t i = j < skip_lo ? j : j + gap;
        one(p[i], g[i], m[i], v[i]);
    }
}

// ---------------------------------------------------------------------------
// Rollout side of MultiCategoricalDistribution (stable_baselines3/common/distributions.py:299-352 as used
// by ActorCriticPolicy.forward, policies.py:1024-1030): sample() + log_prob() of the sampled action in
// ONE launch instead of ~25 element-wise / reduction kernels per head.  wave = row: log-sum-exp of each head's
// logits, then the inverse CDF of softmax at u[row][head] in [0, 1) (deterministic: the first arg-max,
// torch.argmax semantics); the row's log-prob is the sum over heads in head order.
// ---------------------------------------------------------------------------
struct SampleArgs {
    int batch, n_logits, n_heads, deterministic;
    int head_dims[kMaxHeads], head_off[kMaxHeads];
};

// one WORKGROUP per row, one WAVE per head (round 4: a wave used to walk the row's heads one after the other -- six dependent chains of
// loads, exp and shuffle scans, 13-16 us on the rollout step's critical path): lanes stride over the head's logits (wave-parallel max /
// sum-exp, inclusive scan for the CDF); the row's log-prob is summed by thread 0 in head order, as the sequential loop did.
__global__ __launch_bounds__(64 * kMaxHeads) void k_multicategorical_sample(SampleArgs a, const float *__restrict__ logits, const float *__restrict__ uniforms,
                                                                            int64_t *__restrict__ actions, float *__restrict__ log_prob)
{
    __shared__ float s_lp[kMaxHeads];
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6, b = blockIdx.x;
    {
        const float *row = logits + (size_t)b * a.n_logits + a.head_off[h];
        const int d = a.head_dims[h];
        // max and first arg-max (torch.argmax semantics: lowest index among equals)
        float mx = -INFINITY;
        int amax = 0x7fffffff;
        for (int j = lane; j < d; j += 64) {
            const float v = row[j];
            if (v > mx) { mx = v; amax = j; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(mx, o, 64);
            const int oa = __shfl_xor(amax, o, 64);
            if (om > mx || (om == mx && oa < amax)) { mx = om; amax = oa; }
        }
        float s = 0.0f;
        for (int j = lane; j < d; j += 64) s += expf(row[j] - mx);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        int act = amax;
        if (!a.deterministic) {
            // inverse CDF: first index whose inclusive prefix sum of exp(l - max) exceeds u * sum
            const float target = uniforms[(size_t)b * a.n_heads + h] * s;
            float carry = 0.0f;
            act = d - 1;
            for (int j0 = 0; j0 < d; j0 += 64) {
                const int j = j0 + lane;
                float c = j < d ? expf(row[j] - mx) : 0.0f;
                for (int o = 1; o < 64; o <<= 1) {
                    const float up = __shfl_up(c, o, 64);
                    if (lane >= o) c += up;
                }
                c += carry;
                const unsigned long long hit = __ballot(j < d && c > target);
                if (hit) { act = j0 + __ffsll((long long)hit) - 1; break; }
                carry = __shfl(c, 63, 64);
            }
        }
        if (lane == 0) {
            actions[(size_t)b * a.n_heads + h] = act;
            s_lp[h] = row[act] - (mx + logf(s));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lp = 0.0f;
        for (int k = 0; k < a.n_heads; ++k) lp += s_lp[k];
        log_prob[b] = lp;
    }
}

// ===========================================================================
// C-ABI
// ===========================================================================
GNBV_API int gnbv_gather_minibatch(const int64_t *rows, int batch, int act_dim, const float *actions, const float *values,
                                   const float *log_probs, const float *advantages, const float *returns, float *o_actions,
                                   float *o_values, float *o_log_probs, float *o_adv, float *o_ret, void *stream)
{
    GNBV_CHECK_ARG(rows && actions && values && log_probs && advantages && returns && o_actions && o_values && o_log_probs &&
                   o_adv && o_ret && batch > 0 && act_dim > 0);
    hipLaunchKernelGGL(k_gather_minibatch, dim3((batch + 255) / 256), dim3(256), 0, gnbv_stream(stream), rows, batch, act_dim,
                       actions, values, log_probs, advantages, returns, o_actions, o_values, o_log_probs, o_adv, o_ret);
    return gnbv_launch_status();
}

GNBV_API int gnbv_ppo_loss(const GnbvPpoLoss *a, void *stream)
{
    GNBV_CHECK_ARG(a && a->batch > 1 && a->n_heads > 0 && a->n_heads <= kMaxHeads && a->n_logits > 0);
    GNBV_CHECK_ARG(a->logits && a->values && a->actions && a->old_values && a->old_log_prob && a->advantages && a->returns);
    GNBV_CHECK_ARG(a->d_logits && a->d_values && a->stats && a->stats_row);
    int sum = 0;
    for (int h = 0; h < a->n_heads; ++h) sum += a->head_dims[h];
    GNBV_CHECK_ARG(sum == a->n_logits && a->scratch);
    hipStream_t st = gnbv_stream(stream);
    const int blocks = (a->batch + kLossThreads / 64 - 1) / (kLossThreads / 64);
    bool small_heads = true;
    for (int h = 0; h < a->n_heads; ++h) small_heads = small_heads && a->head_dims[h] >= 1 && a->head_dims[h] <= 128;
    const char *ge = getenv("GENNBV_PPO_GENERIC");  // (=1: the run-time form for every lattice -- A/B runs, the bit-identity test)
    const bool generic_only = ge && ge[0] == '1';
    if (a->n_heads == 6 && small_heads && !generic_only)  // (the reference's lattice has six heads; any other count takes the run-time form)
        hipLaunchKernelGGL((k_ppo_fused<6, true>), dim3(blocks), dim3(kLossThreads), 0, st, *a, a->scratch, (int *)(a->scratch + 8 * (size_t)a->batch));
    else
        hipLaunchKernelGGL((k_ppo_fused<0, false>), dim3(blocks), dim3(kLossThreads), 0, st, *a, a->scratch, (int *)(a->scratch + 8 * (size_t)a->batch));
    return gnbv_launch_status();
}

__global__ __launch_bounds__(kLossThreads) void k_ppo_stats_finish(GnbvPpoLoss a)
{
    __shared__ float fscratch[kLossThreads / 64 + 1];
    ppo_stats_finish(a, a.scratch, fscratch, nullptr);
}

GNBV_API int gnbv_ppo_loss_finish(const GnbvPpoLoss *a, void *stream)
{
    GNBV_CHECK_ARG(a && a->defer_stats && a->batch > 1 && a->scratch && a->stats && a->stats_row);
    hipLaunchKernelGGL(k_ppo_stats_finish, dim3(1), dim3(kLossThreads), 0, gnbv_stream(stream), *a);
    return gnbv_launch_status();
}

GNBV_API int gnbv_ppo_loss_rsl(int batch, const float *log_prob, const float *old_log_prob, const float *advantages, const float *values,
                               const float *target_values, const float *returns, float clip_param, float value_loss_coef,
                               float entropy_coef, int use_clipped_value_loss, float *d_log_prob, float *d_values, float *d_entropy,
                               float *sums, void *stream)
{
    GNBV_CHECK_ARG(batch > 0 && log_prob && old_log_prob && advantages && values && target_values && returns);
    GNBV_CHECK_ARG(d_log_prob && d_values && d_entropy && clip_param > 0.0f);
    hipLaunchKernelGGL(k_ppo_loss_rsl, dim3(1), dim3(kLossThreads), 0, gnbv_stream(stream), batch, log_prob, old_log_prob, advantages,
                       values, target_values, returns, clip_param, value_loss_coef, entropy_coef, use_clipped_value_loss, d_log_prob,
                       d_values, d_entropy, sums);
    return gnbv_launch_status();
}

GNBV_API size_t gnbv_adam_workspace_bytes(void) { return 1026 * sizeof(double) + 64; }

GNBV_API int gnbv_clip_adam_step_ex(const GnbvAdamStep *a, void *stream)
{
    GNBV_CHECK_ARG(a && a->params && a->grads && a->exp_avg && a->exp_avg_sq && a->step && a->norm_out && a->workspace && a->n > 0);
    GNBV_CHECK_ARG(a->workspace_bytes >= gnbv_adam_workspace_bytes());
    GNBV_CHECK_ARG(a->table == nullptr || (a->out && a->counter && a->table_rows > 0 && a->row_len > 0));
    const bool sliced = a->sq_partial != nullptr && a->sq_parts > 0 && a->sq_hi > a->sq_lo;
    GNBV_CHECK_ARG(!sliced || (a->sq_lo >= 0 && a->sq_hi <= a->n));
    GNBV_CHECK_ARG(a->upd_skip_hi <= a->upd_skip_lo || (a->upd_skip_lo >= 0 && a->upd_skip_hi <= a->n));
    hipStream_t st = gnbv_stream(stream);
    double *partial = (double *)a->workspace;
    float *hyper = (float *)(partial + 1026);  // (the 64 bytes behind the partial sums)
    const int64_t gap = sliced ? a->sq_hi - a->sq_lo : 0, n_eff = a->n - gap;
    int blocks = (int)((n_eff + 256 * 16 - 1) / (256 * 16));
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    GnbvPpoLoss fin = {};
    const bool finish = a->loss_finish != nullptr;
    if (finish) {
        fin = *a->loss_finish;
        GNBV_CHECK_ARG(fin.defer_stats && fin.scratch && fin.stats && fin.stats_row && fin.kl_out == nullptr && fin.stop_flag == a->stop_flag);
    }
    const int fold = sliced ? 1 : 0, fin_block = finish ? blocks + fold : -1;
    hipLaunchKernelGGL(k_grad_sqnorm, dim3(blocks + fold + (finish ? 1 : 0)), dim3(256), 0, st, a->grads, n_eff, sliced ? a->sq_lo : a->n, gap, partial,
                       a->grad_scale, a->step, a->stop_flag, a->kl_slot, a->target_kl, blocks, sliced ? a->sq_partial : (const double *)nullptr,
                       sliced ? a->sq_parts : 0, fin_block, fin, hyper, a->lr, a->beta1, a->beta2);
    const int64_t upd_gap = a->upd_skip_hi > a->upd_skip_lo ? a->upd_skip_hi - a->upd_skip_lo : 0;
    int ab = (int)((a->n - upd_gap + 255) / 256);
    // (a sharded step updates the 0.8 M parameters outside the slice here: one element per lane was 3 125 workgroups, ~10 us of DISPATCH
    // for 23 MB of traffic -- two 16-byte trips per lane instead: 17 -> 8 us in the one-rank data-parallel step)
    if (upd_gap > 0) ab = (int)((a->n - upd_gap + 256 * 8 - 1) / (256 * 8));
    // (small parameter sets -- the reference's 20^3 encoder: 0.53 M -- two 16-byte trips per lane: 2 070 workgroups that each repeat the
    // prologue for a quarter of a trip of work became 259; above 2 M parameters the cap below decides as before)
    else if (ab <= 8192) ab = (int)((a->n + 256 * 8 - 1) / (256 * 8));
    ab = ab < 1 ? 1 : ab;
#ifndef GNBV_ADAM_BLOCKS
#define GNBV_ADAM_BLOCKS 8192  // (same-call A/B of the whole bench: 2048 workgroups +5 us per minibatch, 4096 +1-2 us)
#endif
    ab = ab > GNBV_ADAM_BLOCKS ? GNBV_ADAM_BLOCKS : ab;
    hipLaunchKernelGGL(k_adam_flat, dim3(ab), dim3(256), 0, st, a->params, a->grads, a->exp_avg, a->exp_avg_sq, a->n, (const double *)partial, blocks + fold,
                       (const double *)nullptr, 0, a->max_grad_norm, a->grad_scale, a->norm_out,
                       (const float *)nullptr, a->upd_skip_lo, a->upd_skip_hi, (const int *)a->stop_flag, (const int64_t *)a->step, a->lr, a->beta1, a->beta2, a->eps, a->table, a->table_rows, a->row_len, a->out,
                       a->counter, (const float *)hyper);
    return gnbv_launch_status();
}

// The Adam update of a parameter SHARD (data-parallel replicas that own 1 / world of a large slice: gennbv_amd/parallel.py) with the
// clip factor the step's main launch left in norm_out[1]; same step counter / stop flag.
GNBV_API int gnbv_adam_shard_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, const float *norm_out,
                                  float lr, float beta1, float beta2, float eps, const int64_t *step, const int *stop_flag, void *stream)
{
    GNBV_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && norm_out && step && n > 0);
    int ab = (int)((n + 255) / 256);
    ab = ab > 2048 ? 2048 : ab;
    hipLaunchKernelGGL(k_adam_flat, dim3(ab), dim3(256), 0, gnbv_stream(stream), params, grads, exp_avg, exp_avg_sq, n, (const double *)nullptr, 0,
                       (const double *)nullptr, 0, 0.0f, 1.0f, (float *)nullptr, norm_out, (int64_t)0, (int64_t)0, stop_flag, step, lr, beta1, beta2, eps);
    return gnbv_launch_status();
}

// sum(g^2) of a gradient shard as kSqParts fp64 partial sums in one fixed order (the sharded data-parallel update: every rank squares the
// shard of the REDUCED gradient it received; the partial sums ride an all-reduce and enter the clip factor through GnbvAdamStep.sq_partial)
constexpr int kSqParts = 256;
constexpr int kSqThreads = 1024;  // (round 5: 256 lanes per partial sum kept 4 requests x 64 K lanes in flight: 3.1 TB/s over a 55 MB shard)
__global__ __launch_bounds__(kSqThreads) void k_sq_partials(const float *__restrict__ g, int64_t n, double *__restrict__ partial)
{
    __shared__ double sh[kSqThreads];
    double acc = 0.0;
    const bool vec = (((uintptr_t)g & 15) == 0);
    const int64_t n4 = vec ? n / 4 : 0, stride = (int64_t)gridDim.x * kSqThreads;
    for (int64_t i0 = (int64_t)blockIdx.x * kSqThreads + threadIdx.x; i0 < n4; i0 += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const float4 *>(g)[min(i0 + u * stride, n4 - 1)];  // (clamped: unconditional requests)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * stride < n4)
                acc += (double)v[u].x * (double)v[u].x + (double)v[u].y * (double)v[u].y + (double)v[u].z * (double)v[u].z + (double)v[u].w * (double)v[u].w;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kSqThreads + threadIdx.x; i < n; i += stride) acc += (double)g[i] * (double)g[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int d = kSqThreads / 2; d > 0; d >>= 1) {
        if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

GNBV_API int gnbv_sq_partials_count(void) { return kSqParts; }

GNBV_API int gnbv_sq_partials(const float *grads, int64_t n, double *partial, void *stream)
{
    GNBV_CHECK_ARG(grads && partial && n > 0);
    hipLaunchKernelGGL(k_sq_partials, dim3(kSqParts), dim3(kSqThreads), 0, gnbv_stream(stream), grads, n, partial);
    return gnbv_launch_status();
}

static GnbvAdamStep adam_args(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float max_grad_norm, float lr, float beta1,
                              float beta2, float eps, int64_t *step, int *stop_flag, float grad_scale, const float *kl_slot, float target_kl,
                              float *norm_out, void *workspace, size_t workspace_bytes)
{
    GnbvAdamStep a = {};
    a.params = params; a.grads = grads; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.n = n;
    a.max_grad_norm = max_grad_norm; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.step = step; a.stop_flag = stop_flag; a.grad_scale = grad_scale; a.kl_slot = kl_slot; a.target_kl = target_kl;
    a.norm_out = norm_out; a.workspace = workspace; a.workspace_bytes = workspace_bytes;
    return a;
}

GNBV_API int gnbv_clip_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float max_grad_norm,
                                 float lr, float beta1, float beta2, float eps, int64_t *step, int *stop_flag, float grad_scale,
                                 const float *kl_slot, float target_kl, float *norm_out, void *workspace, size_t workspace_bytes, void *stream)
{
    const GnbvAdamStep a = adam_args(params, grads, exp_avg, exp_avg_sq, n, max_grad_norm, lr, beta1, beta2, eps, step, stop_flag, grad_scale, kl_slot,
                                     target_kl, norm_out, workspace, workspace_bytes);
    return gnbv_clip_adam_step_ex(&a, stream);
}

GNBV_API int gnbv_clip_adam_step_rotate(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float max_grad_norm,
                                        float lr, float beta1, float beta2, float eps, int64_t *step, int *stop_flag, float grad_scale,
                                        const float *kl_slot, float target_kl, float *norm_out, void *workspace, size_t workspace_bytes,
                                        const int64_t *table, int table_rows, int row_len, int64_t *out, int *counter, void *stream)
{
    GNBV_CHECK_ARG(table && out && counter && table_rows > 0 && row_len > 0);
    GnbvAdamStep a = adam_args(params, grads, exp_avg, exp_avg_sq, n, max_grad_norm, lr, beta1, beta2, eps, step, stop_flag, grad_scale, kl_slot,
                               target_kl, norm_out, workspace, workspace_bytes);
    a.table = table; a.table_rows = table_rows; a.row_len = row_len; a.out = out; a.counter = counter;
    return gnbv_clip_adam_step_ex(&a, stream);
}

GNBV_API int gnbv_multicategorical_sample(const float *logits, int batch, int n_logits, int n_heads, const int *head_dims,
                                          const float *uniforms, int deterministic, int64_t *actions, float *log_prob, void *stream)
{
    GNBV_CHECK_ARG(logits && head_dims && actions && log_prob && batch > 0 && n_heads > 0 && n_heads <= kMaxHeads);
    GNBV_CHECK_ARG(deterministic || uniforms);
    SampleArgs a;
    a.batch = batch; a.n_logits = n_logits; a.n_heads = n_heads; a.deterministic = deterministic;
    int off = 0;
    for (int h = 0; h < kMaxHeads; ++h) {
        a.head_dims[h] = h < n_heads ? head_dims[h] : 0;
        a.head_off[h] = off;
        if (h < n_heads) {
            GNBV_CHECK_ARG(head_dims[h] > 0);
            off += head_dims[h];
        }
    }
    GNBV_CHECK_ARG(off == n_logits);
    hipLaunchKernelGGL(k_multicategorical_sample, dim3(batch), dim3(64 * a.n_heads), 0, gnbv_stream(stream), a, logits, uniforms, actions, log_prob);
    return gnbv_launch_status();
}
