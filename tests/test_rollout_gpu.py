"""GPU: the whole C5 path on the gfx950 kernels -- ReplayFeedEnv (HIP state encoding, in-place observation rows), fused
policy head, one-launch bootstrap + add (gnbv_rollout_add), HIP GAE -- against the reference's own collect_rollouts
driving the reference's own env (fixture F11, oracle/gen_golden_rollout.py).  The GPU sampler draws from a different
random stream than torch's CPU Categorical, so the reference's sampled actions are forced; everything downstream of
the actions must then reproduce: every observation row bit for bit, env rewards bit for bit, values / log-probs /
bootstrapped rewards / advantages / returns within fp32 round-off, episode_starts exactly."""
import numpy as np
import pytest
import torch

from tests import golden_util as gu
from tests import rollout_util as ru
from tests.test_envstep_gpu import make_env_from_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _force_actions(algo, fx):
    """policy.forward -> (the reference's action of this call, V, log pi(action)) from the HIP head's own logits."""
    pol = algo.policy
    t = int(fx["T"])
    seq = iter([fx[f"r{r}/actions_in"][s] for r in range(2) for s in range(t)])

    def forward(obs, deterministic=False):
        fused = pol._fused_head(obs)
        assert fused is not None, "the rollout must run on the fused gfx950 head"
        logits, values = fused
        a = torch.from_numpy(next(seq).astype(np.int64)).to(DEV)
        lp = pol.action_dist.proba_distribution(logits).log_prob(a.float())
        return a, values.unsqueeze(1), lp
    pol.forward = forward


@pytest.mark.parametrize("fused_add", ["1", "0"])
def test_hip_rollout_reproduces_reference_collect_rollouts(fused_add, monkeypatch):
    fx = gu.load("F11_rollout")
    env, cfg = make_env_from_fixture(fx)
    t, n = int(fx["T"]), int(fx["n"])
    algo = ru.make_algo(env, DEV, "hip", t)
    algo.fused_add = fused_add == "1"
    algo._setup_learn(total_timesteps=10 ** 9)  # env.reset() into buffer row 0 (frame 0), episode starts = 1
    reset_rows = ru.unpack_rows(fx["reset_state"], fx["reset_grid"], fx["reset_rgb"])
    assert np.array_equal(algo._last_obs.cpu().numpy(), reset_rows)
    env.episode_length_buf = torch.from_numpy(fx["init_episode_length"].astype(np.int64)).to(DEV)
    _force_actions(algo, fx)
    for r in range(2):
        assert algo.collect_rollouts(env, None, algo.rollout_buffer, n_rollout_steps=t)
        buf = algo.rollout_buffer
        rows = ru.unpack_rows(fx[f"r{r}/obs_state"], fx[f"r{r}/obs_grid"], fx[f"r{r}/obs_rgb"])
        assert np.array_equal(buf.observations[:t].cpu().numpy(), rows), "observation rows (HIP env, written in place)"
        last = ru.unpack_rows(fx[f"r{r}/last_state"], fx[f"r{r}/last_grid"], fx[f"r{r}/last_rgb"])
        assert np.array_equal(buf.observations[t].cpu().numpy(), last)
        ru.check_rollout(fx, r, buf, algo, value_tol=2e-5)
    assert algo.num_timesteps == int(fx["num_timesteps"])



@pytest.mark.parametrize("g,compact", [(64, True), (20, False)])
def test_prepared_rollout_forward_is_bit_identical_to_the_general_path(g, compact):
    """ops/rollout_plan.RolloutForward (collect_rollouts' policy evaluation with the step-invariant work hoisted out, the parameter-only
    launches run once per rollout: GnbvEncoderParams.eval_prepared) issues the same kernels with the same arguments as
    ActorCriticPolicy_Train_Eval.forward: two rollouts at G = 64 with compact rows (the bench's kernel set) and at G = 20 with fp32
    observation rows (round 6: the plan's "flat" form -- the reference's own workload, whose fc layer cannot fold BatchNorm-2), a train()
    in between so that the second rollout must prepare again from CHANGED parameters -- every buffer bit for bit against
    `rollout_plan = False`."""
    from gennbv_amd.env import synthetic as S
    from gennbv_amd.env.config import TaskConfig
    from gennbv_amd.env.replay_feed import ReplayFeed, ReplayFeedEnv
    from gennbv_amd.network.hybrid_encoder import Hybrid_Encoder
    from gennbv_amd.sb3.policies import ActorCriticPolicy_Train_Eval
    from gennbv_amd.sb3.ppo_grid_obs import PPO_Grid_Obs
    n, t = 16, 6
    cfg = TaskConfig(camera_width=80, camera_height=60, grid_size=g)
    kw = dict(net_arch=[], features_extractor_class=Hybrid_Encoder, features_extractor_kwargs=dict(
        encoder_param={"hidden_shapes": [256, 256], "visual_dim": 256},
        net_param={"transformer_params": [[1, 256], [1, 256]], "append_hidden_shapes": [256, 256]},
        state_input_shape=(cfg.state_dim,), visual_input_shape=(cfg.stack, cfg.camera_height, cfg.camera_width)))
    out = {}
    for use_plan in (False, True):
        torch.manual_seed(0)
        np.random.seed(0)
        scene = S.make_scenes(n, g, seed=5, device=DEV)
        env = ReplayFeedEnv(cfg, scene, ReplayFeed.synthetic(scene, cfg, 5, seed=5), DEV, max_episode_length=4)
        algo = PPO_Grid_Obs(ActorCriticPolicy_Train_Eval, env, learning_rate=1e-4, n_steps=t, batch_size=32, n_epochs=1, gamma=0.99, gae_lambda=0.95,
                            clip_range=0.2, clip_range_vf=0.2, ent_coef=0.01, vf_coef=0.8, max_grad_norm=1.0, target_kl=None, seed=1, device=DEV,
                            compact_obs=compact, policy_kwargs=kw)
        algo.rollout_plan = use_plan
        algo._setup_learn(total_timesteps=10 ** 9)
        snaps = []
        for r in range(2):
            assert algo.collect_rollouts(env, None, algo.rollout_buffer, n_rollout_steps=t)
            buf = algo.rollout_buffer
            snaps.append({k: getattr(buf, k).detach().clone() for k in ("observations", "grid_i8", "autocorr", "actions", "values", "log_probs", "rewards",
                                                                         "advantages", "returns", "episode_starts") if getattr(buf, k, None) is not None})
            assert ("grid_i8" in snaps[-1]) == compact
            if r == 0:
                algo.train()
        plan = getattr(algo, "_rollout_plan_obj", None)
        assert (plan is not None and plan.prepared) == use_plan, "the prepared forward must have run (and only when asked for)"
        out[use_plan] = snaps
    for r in range(2):
        for k, a in out[False][r].items():
            assert torch.equal(a, out[True][r][k]), (r, k)
    assert float(out[True][1]["values"].abs().max()) > 0 and not torch.equal(out[True][0]["values"], out[True][1]["values"])


def test_flag_views_are_scoped_to_collect_rollouts():
    """`env.flag_views` (dones / time_outs as views of the kernels' bytes) is on only inside collect_rollouts -- also when a
    callback ends the rollout early or raises --, and `_last_episode_starts` is a copy: env steps between two rollouts (an evaluation on
    the same env) change neither it nor, through it, the next rollout's episode_starts row 0."""
    from gennbv_amd.env import synthetic as S
    from gennbv_amd.env.config import TaskConfig
    from gennbv_amd.env.replay_feed import ReplayFeed, ReplayFeedEnv
    from gennbv_amd.network.hybrid_encoder import Hybrid_Encoder
    from gennbv_amd.sb3.policies import ActorCriticPolicy_Train_Eval
    from gennbv_amd.sb3.ppo_grid_obs import PPO_Grid_Obs
    g, n, t = 16, 8, 5
    cfg = TaskConfig(camera_width=80, camera_height=60, grid_size=g)
    kw = dict(net_arch=[], features_extractor_class=Hybrid_Encoder, features_extractor_kwargs=dict(
        encoder_param={"hidden_shapes": [256, 256], "visual_dim": 256},
        net_param={"transformer_params": [[1, 256], [1, 256]], "append_hidden_shapes": [256, 256]},
        state_input_shape=(cfg.state_dim,), visual_input_shape=(cfg.stack, cfg.camera_height, cfg.camera_width)))
    torch.manual_seed(0)
    np.random.seed(0)
    scene = S.make_scenes(n, g, seed=3, device=DEV)
    env = ReplayFeedEnv(cfg, scene, ReplayFeed.synthetic(scene, cfg, 5, seed=3), DEV, max_episode_length=3)
    algo = PPO_Grid_Obs(ActorCriticPolicy_Train_Eval, env, learning_rate=1e-4, n_steps=t, batch_size=8, n_epochs=1, gamma=0.99, gae_lambda=0.95,
                        target_kl=None, seed=1, device=DEV, compact_obs=True, policy_kwargs=kw)
    algo._setup_learn(total_timesteps=10 ** 9)
    assert env.flag_views is False
    assert algo.collect_rollouts(env, None, algo.rollout_buffer, n_rollout_steps=t)
    assert env.flag_views is False, "the switch is put back at the end of the rollout"
    starts = algo._last_episode_starts.clone()
    for _ in range(3):  # somebody else steps the env (three steps: both ping-pong buffers of `dones` are overwritten, an episode ends)
        _, _, d, info = env.step(torch.zeros(n, 6, device=DEV))
        assert d.dtype == torch.bool and info["time_outs"].dtype == torch.bool
    assert torch.equal(algo._last_episode_starts, starts), "_last_episode_starts must not alias the env's buffers"

    class Stop:
        def __init__(self, raise_): self.raise_, self.k = raise_, 0
        def on_rollout_start(self): pass
        def on_rollout_end(self): pass
        def update_locals(self, loc): pass
        def on_step(self):
            self.k += 1
            if self.k == 2 and self.raise_:
                raise RuntimeError("callback failed")
            return self.k < 2
    algo._pending = None
    algo._last_obs = env.reset(obs_out=algo.rollout_buffer.first_obs_row(), grid_i8_out=algo.rollout_buffer.grid_i8[0])
    algo.rollout_buffer.update_autocorr(0)
    assert algo.collect_rollouts(env, Stop(False), algo.rollout_buffer, n_rollout_steps=t) is False
    assert env.flag_views is False, "early return of a callback"
    algo._pending = None
    with pytest.raises(RuntimeError):
        algo.collect_rollouts(env, Stop(True), algo.rollout_buffer, n_rollout_steps=t)
    assert env.flag_views is False, "exception inside the rollout"
