"""GPU: PPO_Grid_Obs.train() on the fused gfx950 path (HIP encoder, fused loss kernel, flat Adam,
device-side early stop, hipGraph replay) against the reference's train() goldens (F9)."""
import numpy as np
import pytest
import torch

from tests import golden_util as gu
from tests.test_policy_ppo_cpu import _ppo_from_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _check(ppo, fx, loss_tol=1e-4):
    ppo.train()
    log = ppo.logger.name_to_value
    for k in ("train/entropy_loss", "train/policy_gradient_loss", "train/value_loss", "train/approx_kl",
              "train/clip_fraction", "train/loss", "train/explained_variance"):
        ref = float(fx["log/" + k])
        assert abs(float(log[k]) - ref) <= loss_tol * max(1.0, abs(ref)), (k, log[k], ref)
    if ppo._hip is not None:
        assert int(ppo._hip["opt"].step_count.item()) == int(fx["n_optimizer_steps"])  # early-stop position
    for name, p in ppo.policy.named_parameters():
        a = p.detach().cpu().numpy()
        mine = a if a.size <= 70000 else a.reshape(-1)[::97]
        np.testing.assert_allclose(mine, fx["final/" + name], rtol=2e-3, atol=3e-4, err_msg=name)
    for k, v in ppo.policy.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), fx["final_bn/" + k], rtol=1e-4, atol=1e-5)
        if "num_batches" in k:
            assert int(v) == int(fx["final_bn/" + k]), k


@pytest.mark.parametrize("name", ["F9_ppo_train", "F9_ppo_train_earlystop"])
@pytest.mark.parametrize("graph", [False, True, "best of 2"])
def test_fused_train_matches_reference(name, graph):
    """graph = "best of 2": the minibatch is captured twice and the faster capture kept (PPO_Grid_Obs._best_of_captures: the candidates
    are timed by MASKED replays) -- the reference's losses, early-stop position, parameters and BatchNorm statistics must come out all
    the same, i.e. the timing replays leave no trace."""
    fx = gu.load(name)
    ppo = _ppo_from_fixture(fx, device=DEV, backend="hip")
    ppo.use_graph = bool(graph)
    if graph == "best of 2":
        ppo.graph_candidates = 2
    _check(ppo, fx)
    if graph == "best of 2":
        assert len(ppo.graph_capture_ms) == 2 and min(ppo.graph_capture_ms) > 0


def test_recapture_reuses_the_pinned_pool_and_its_addresses():
    """Round 5: what made one capture of the minibatch slower than another is the placement of its private memory pool, so the pool of
    the capture that was kept is pinned (a torch.cuda.MemPool held in `_hip["graph_pool"]`) and a re-capture -- here forced by a new
    clip range, a kernel argument baked into the graph -- allocates the same intermediates in the same blocks: same pool object, no
    new device memory reserved, and the update still follows the reference (the second call trains on from the first call's result:
    only finiteness and the step count are checked for it)."""
    fx = gu.load("F9_ppo_train")
    ppo = _ppo_from_fixture(fx, device=DEV, backend="hip")
    ppo.use_graph = True
    _check(ppo, fx)
    st = ppo._hip
    pool, captures = st["graph_pool"], st["captures"]  # (no reference to the graph itself: its blocks must be free for the re-capture)
    assert pool is not None and st["graph"] is not None
    torch.cuda.synchronize()

    def pool_segments():  # (address, size) of the device segments that belong to the pinned pool
        return sorted((sg["address"], sg["total_size"]) for sg in torch.cuda.memory_snapshot() if tuple(sg.get("segment_pool_id", ())) == tuple(pool.id))
    seg0 = pool_segments()
    assert seg0, "the captured minibatch allocates its intermediates from the pinned pool"
    steps = int(st["opt"].step_count.item())
    ppo.clip_range = lambda _: 0.15  # -> st["hyper"] changes -> the graph is dropped and captured again
    ppo.train()
    torch.cuda.synchronize()
    assert ppo._hip is st and st["graph"] is not None and st["captures"] == captures + 1, "the minibatch must have been captured again"
    assert st["graph_pool"] is pool, "re-captures go into the pinned pool"
    assert pool_segments() == seg0, ("the re-capture must find its blocks in the pool's existing segments", seg0, pool_segments())
    assert int(st["opt"].step_count.item()) > steps
    for p_ in ppo.policy.parameters():
        assert bool(torch.isfinite(p_).all())


def test_torch_backend_on_gpu_matches_reference_losses():
    """The torch-module path on the GPU (library conv/BN) also reproduces the logged losses."""
    fx = gu.load("F9_ppo_train")
    ppo = _ppo_from_fixture(fx, device=DEV, backend="torch")
    _check(ppo, fx, loss_tol=2e-4)


def test_ppo_loss_kernel_vs_torch_autograd():
    """d(loss)/d(logits, values) and the six statistics of k_ppo_loss against torch autograd on
    the reference's loss expression (ppo_grid_obs.py:209-262)."""
    from gennbv_amd.ops.ppo_ops import PpoLossOp
    from gennbv_amd.sb3.distributions import MultiCategoricalDistribution
    torch.manual_seed(0)
    B, dims = 96, [81, 81, 51, 1, 13, 13]
    op = PpoLossOp(B, dims, DEV, 4, 0.2, 0.2, 0.01, 0.8, 10.0, 0.05)
    logits = torch.randn(B, sum(dims), device=DEV, requires_grad=True)
    values = torch.randn(B, device=DEV, requires_grad=True)
    op.actions.copy_(torch.stack([torch.randint(0, n, (B,)) for n in dims], -1).float())
    op.old_values.copy_(values.detach() + 0.3 * torch.randn(B, device=DEV))
    op.advantages.copy_(torch.randn(B, device=DEV) * 2 + 0.5)
    op.returns.copy_(torch.randn(B, device=DEV))
    dist = MultiCategoricalDistribution(dims).proba_distribution(logits)
    log_prob, entropy = dist.log_prob(op.actions), dist.entropy()
    op.old_log_prob.copy_(log_prob.detach() + 0.25 * torch.randn(B, device=DEV))
    adv = (op.advantages - op.advantages.mean()) / (op.advantages.std() + 1e-8)
    ratio = torch.exp(log_prob - op.old_log_prob)
    pg = -torch.min(adv * ratio, adv * torch.clamp(ratio, 0.8, 1.2)).mean()
    vp = op.old_values + torch.clamp(values - op.old_values, -0.2, 0.2)
    vl = torch.nn.functional.mse_loss(op.returns, vp)
    el = -entropy.mean()
    loss = 10 * pg + 0.01 * el + 0.8 * vl
    loss.backward()
    dl, dv = op(logits.detach().contiguous(), values.detach().contiguous())
    torch.testing.assert_close(dl, logits.grad, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(dv, values.grad, rtol=1e-4, atol=1e-7)
    st = op.stats[0].cpu()
    lr_ = (log_prob - op.old_log_prob).detach()
    ref = torch.tensor([pg.item(), vl.item(), el.item(), ((torch.exp(lr_) - 1) - lr_).mean().item(),
                        (torch.abs(ratio - 1) > 0.2).float().mean().item(), loss.item()])
    torch.testing.assert_close(st[:6], ref, rtol=1e-5, atol=1e-6)
    assert int(op.stats_row.item()) == 1 and int(op.stop_flag.item()) == int(ref[3] > 0.075)


@pytest.mark.parametrize("dims", [[81, 81, 51, 1, 13, 13], [7, 5, 128, 1, 64, 2]])
def test_ppo_loss_six_head_kernel_is_bit_identical_to_the_runtime_form(dims, monkeypatch):
    """Round 6: with six heads of <= 128 categories (the reference's lattice) the loss kernel is the instantiation whose per-head code is one
    straight-line block (k_ppo_fused<6, true>: the heads' reductions interleave); GENNBV_PPO_GENERIC=1 selects the run-time form every other
    lattice takes.  Same operations per head in the same order: gradients, per-sample terms and statistics must be the same bits."""
    from gennbv_amd.ops.ppo_ops import PpoLossOp
    torch.manual_seed(1)
    B = 128
    logits = torch.randn(B, sum(dims), device=DEV) * 2
    values = torch.randn(B, device=DEV)
    acts = torch.stack([torch.randint(0, n, (B,)) for n in dims], -1).float().to(DEV)
    rnd = [torch.randn(B, device=DEV) for _ in range(4)]
    out = []
    for generic in ("0", "1"):
        monkeypatch.setenv("GENNBV_PPO_GENERIC", generic)
        op = PpoLossOp(B, dims, DEV, 4, 0.2, 0.2, 0.01, 0.8, 10.0, 0.05)
        op.actions.copy_(acts)
        op.old_values.copy_(values + 0.3 * rnd[0])
        op.advantages.copy_(rnd[1] * 2 + 0.5)
        op.returns.copy_(rnd[2])
        op.old_log_prob.copy_(-8.0 + 0.25 * rnd[3])
        dl, dv = op(logits, values)
        torch.cuda.synchronize()
        out.append((dl.clone(), dv.clone(), op.scratch[:8 * B].clone(), op.stats[0].clone()))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert bool(torch.isfinite(out[0][0]).all()) and float(out[0][0].abs().max()) > 0


def test_flat_adam_matches_torch_adam_with_clipping():
    from gennbv_amd.ops.ppo_ops import FlatAdam
    torch.manual_seed(1)
    m1 = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5)).to(DEV)
    m2 = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5)).to(DEV)
    m2.load_state_dict(m1.state_dict())
    o1 = torch.optim.Adam(m1.parameters(), lr=3e-3, eps=1e-5)
    o2 = FlatAdam(m2, lr=3e-3, eps=1e-5)
    x = torch.randn(64, 37, device=DEV)
    for it in range(25):
        o1.zero_grad(); o2.zero_grad()
        (m1(x) ** 2).sum().backward(); (m2(x) ** 2).sum().backward()
        torch.nn.utils.clip_grad_norm_(m1.parameters(), 1.0)
        o1.step(); o2.step(1.0)
    for a, b in zip(m1.parameters(), m2.parameters()):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-6)
    flag = torch.ones(1, dtype=torch.int32, device=DEV)
    before = o2.params.clone()
    o2.step(1.0, flag)  # masked: nothing moves, step count frozen
    assert torch.equal(before, o2.params) and int(o2.step_count.item()) == 25
    # gnbv_clip_adam_step_rotate: the same update, and the launch leaves row (counter + 1) % rows of the table in `out`
    # (the replayed minibatch graph's row numbers) -- also when the update is masked
    table = torch.arange(3 * 7, dtype=torch.int64, device=DEV).view(3, 7) * 11
    out = torch.full((7,), -1, dtype=torch.int64, device=DEV)
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    for want in (1, 2, 0, 1):
        o1.zero_grad(); o2.zero_grad()
        (m1(x) ** 2).sum().backward(); (m2(x) ** 2).sum().backward()
        torch.nn.utils.clip_grad_norm_(m1.parameters(), 1.0)
        o1.step(); o2.step(1.0, rotate=(table, out, counter))
        assert int(counter.item()) == want and torch.equal(out, table[want])
    for a, b in zip(m1.parameters(), m2.parameters()):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-6)
    before = o2.params.clone()
    o2.step(1.0, flag, rotate=(table, out, counter))
    assert torch.equal(before, o2.params) and int(counter.item()) == 2 and torch.equal(out, table[2])
    # GnbvAdamStep.sq_*: the squared sum of one gradient slice comes from its producer (here: computed on the side, in two parts)
    # and the norm pass skips the slice -- same update
    lo, hi = o2.slice_of(m2[0].weight)
    assert (lo, hi) == (0, 37 * 19)
    for it in range(3):
        o1.zero_grad(); o2.zero_grad()
        (m1(x) ** 2).sum().backward(); (m2(x) ** 2).sum().backward()
        torch.nn.utils.clip_grad_norm_(m1.parameters(), 1.0)
        g = o2.grads[lo + 4:hi - 3].double() ** 2
        part = torch.stack((g[:100].sum(), g[100:].sum()))
        o1.step(); o2.step(1.0, sq_slice=(lo + 4, hi - 3, part))
    for a, b in zip(m1.parameters(), m2.parameters()):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("n,offset", [(1, 0), (1023, 0), (4099, 1), (1 << 20, 0), (1728000, 0), (13824000 // 8 + 2, 3)])
def test_sq_partials_fixed_order_fp64(n, offset):
    """gnbv_sq_partials (the sharded data-parallel update's sum of squares of the gradient shard a rank received): 256 fp64 partial sums whose
    total equals sum(g^2) in fp64, the same bits on every call; any length, any 4-byte alignment (a shard starts wherever lo + rank * sh falls)."""
    from gennbv_amd import _lib
    lib = _lib.load()
    parts = int(lib.gnbv_sq_partials_count())
    gen = torch.Generator().manual_seed(n)
    full = (torch.randn(n + offset, generator=gen) * 3e-2).to(DEV)
    g = full[offset:]
    out = [torch.full((parts,), -1.0, dtype=torch.float64, device=DEV) for _ in range(2)]
    for o in out:
        _lib.check(lib.gnbv_sq_partials(g.data_ptr(), n, o.data_ptr(), _lib.stream_ptr(torch.device(DEV))), "gnbv_sq_partials")
    torch.cuda.synchronize()
    assert torch.equal(out[0], out[1]) and bool((out[0] >= 0).all())
    want = float((g.double() ** 2).sum())
    assert abs(float(out[0].sum()) - want) <= 1e-11 * max(want, 1e-30)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _one_rank_rccl_worker(rank, name, shard, order, port, out):
    """Child process of test_data_parallel_code_path_on_one_gpu (the environment switches were set by the parent before the spawn)."""
    import os
    import torch.distributed as dist
    from gennbv_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    parallel.capture_safe_env()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        fx = gu.load(name)
        ppo = _ppo_from_fixture(fx, device=DEV, backend="hip")
        parallel.attach(ppo, 1, always_sync=True)
        _check(ppo, fx)
        assert (getattr(ppo._hip["opt"], "shard", None) is not None) == shard
        if shard:
            assert ppo.dp_graph_mode == "one hipGraph incl. RCCL collectives", ppo.dp_graph_mode
        assert (ppo._hip.get("rows_rot") is not None) == (ppo._hip.get("graph") is not None)
        torch.cuda.synchronize()
        out["ok"] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,shard,order", [("F9_ppo_train", False, "r5"), ("F9_ppo_train_earlystop", False, "r5"), ("F9_ppo_train", True, "r5"),
                                              ("F9_ppo_train_earlystop", True, "r5")])
def test_data_parallel_code_path_on_one_gpu(name, shard, order, monkeypatch):
    """The multi-GPU branch (graph body -> RCCL all-reduce of the flat gradient + KL slot -> clip/Adam
    tail with the global KL decision) with a one-rank NCCL communicator must reproduce the goldens.
    shard: the sharded update of fc_grid.weight (reduce-scatter -> owner's Adam -> all-gather, GENNBV_FORCE_SHARD=1 makes the one
    rank its only owner) -- the RCCL reduce_scatter / all_gather calls captured in the minibatch hipGraph.
    The order of the step (round 5; the round-4 order and its switches were retired in round 6): the exchange of the late gradients is issued
    behind the second stream (the conv backward does not wait for the pose branch / fc_grid dW), and the Adam launch deals out the next
    minibatch's rows and statistics (rotation table) inside the graph.
    In a CHILD process (round 5): a process group's watchdog thread lives for the rest of its process, and this stack's has aborted twice in a
    few hundred processes that mix eager and captured collectives (gennbv_amd/parallel.py capture_safe_env) -- the pytest process never
    creates a NCCL process group, so such an abort fails this one case instead of ending the run."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("GENNBV_FORCE_SHARD", "1" if shard else "0")
    out = mp.Manager().dict()
    mp.spawn(_one_rank_rccl_worker, args=(name, shard, order, _free_port(), out), nprocs=1, join=True)
    assert out.get("ok") is True


@pytest.mark.parametrize("dims", [(81, 81, 51, 1, 13, 13), (7, 5, 3, 1, 4, 2), (200, 1, 65)])  # the reference's action lattice first (81-way heads > one wave)
def test_multicategorical_sample_kernel_distribution_and_log_prob(dims):
    """gnbv_multicategorical_sample (rollout side of MultiCategoricalDistribution, sb3 distributions.py:299-352):
    log_prob equals the torch evaluation of the sampled action bit-for-bit up to fp32 round-off, the mode
    equals torch.argmax, and the empirical frequencies match softmax within sampling error."""
    from gennbv_amd.sb3.distributions import MultiCategoricalDistribution
    dims = list(dims)
    dist = MultiCategoricalDistribution(dims)
    gen = torch.Generator().manual_seed(3)
    logits = (torch.randn(4096, sum(dims), generator=gen) * 2).to("cuda:0")
    torch.manual_seed(11)
    actions, lp = dist.sample_and_log_prob(logits)
    assert actions.shape == (4096, len(dims)) and actions.dtype == torch.int64
    ref = dist.proba_distribution(logits).log_prob(actions.float())
    assert torch.allclose(lp, ref, rtol=1e-5, atol=1e-5)
    for h, d in enumerate(dims):
        assert int(actions[:, h].min()) >= 0 and int(actions[:, h].max()) < d
    mode, lpm = dist.sample_and_log_prob(logits, deterministic=True)
    assert torch.equal(mode, dist.proba_distribution(logits).mode())
    # frequencies: one row repeated -> empirical distribution of head 0 vs softmax
    row = logits[:1].repeat(20000, 1)
    a, _ = dist.sample_and_log_prob(row)
    p = torch.softmax(row[0, :dims[0]], 0)
    freq = torch.bincount(a[:, 0], minlength=dims[0]).float() / a.shape[0]
    assert float((freq - p).abs().max()) < 4 * float((p * (1 - p) / a.shape[0]).sqrt().max()) + 1e-3


def test_checkpoint_roundtrip_of_the_hip_train_state(tmp_path):
    """SB3-zip save/load on the gfx950 train path: policy weights and the flat HIP Adam's moments / step survive
    save(include_optimizer=True) -> load(), and training continues bit-identically from the checkpoint."""
    from gennbv_amd.sb3 import save_util
    fx = gu.load("F9_ppo_train")
    a = _ppo_from_fixture(fx, device=DEV, backend="hip")
    a.train()
    path = str(tmp_path / "ck")
    a.save(path, include_optimizer=True)
    _, params, _, _ = save_util.load_from_zip_file(path)
    assert set(params) == {"policy", "policy.optimizer"}
    b = _ppo_from_fixture(fx, device=DEV, backend="hip")
    b.set_parameters(path)
    for (k, x), (_, y) in zip(a.policy.state_dict().items(), b.policy.state_dict().items()):
        assert torch.equal(x, y), k
    a.train()
    b.train()  # builds its flat Adam from the loaded torch-format state
    oa, ob = a._hip["opt"], b._hip["opt"]
    assert int(oa.step_count) == int(ob.step_count) > 0
    assert torch.equal(oa.exp_avg, ob.exp_avg) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)
    assert torch.equal(oa.params, ob.params)


def test_learn_with_int8_grid_copy_matches_fp32_rows(monkeypatch):
    """End to end over the replay env: every row of the buffer's int8 grid copy equals the grid slice of the fp32
    observation row the env wrote (rows 0..T, two rollouts incl. the carry-over of the last row), and a learn() with the
    copy enabled ends with exactly the parameters of a learn() that reads the fp32 rows."""
    from gennbv_amd.env import synthetic as S
    from gennbv_amd.env.config import TaskConfig
    from gennbv_amd.env.replay_feed import ReplayFeed, ReplayFeedEnv
    from gennbv_amd.network.hybrid_encoder import Hybrid_Encoder
    from gennbv_amd.sb3.policies import ActorCriticPolicy_Train_Eval
    from gennbv_amd.sb3.ppo_grid_obs import PPO_Grid_Obs
    n, g, t = 8, 16, 4
    monkeypatch.setenv("GENNBV_FUSED_BWD", "0")  # bit equality: the separate backward kernels on both sides,
    monkeypatch.setenv("GENNBV_ANALYTIC_BN1", "0")  # BatchNorm-1 statistics from the activations on both sides

    def run(i8: bool):
        torch.manual_seed(0)
        np.random.seed(0)
        cfg = TaskConfig(camera_width=64, camera_height=48, grid_size=g)
        scene = S.make_scenes(n, g, seed=2, device=DEV)
        feed = ReplayFeed.synthetic(scene, cfg, 5, seed=2)
        env = ReplayFeedEnv(cfg, scene, feed, DEV, max_episode_length=6)
        algo = PPO_Grid_Obs(ActorCriticPolicy_Train_Eval, env, learning_rate=1e-4, n_steps=t, batch_size=8, n_epochs=2, ent_coef=0.01,
                            vf_coef=0.8, max_grad_norm=1.0, target_kl=None, seed=1, device=DEV,
                            policy_kwargs=dict(net_arch=[], features_extractor_class=Hybrid_Encoder, features_extractor_kwargs=dict(
                                encoder_param={"hidden_shapes": [256, 256], "visual_dim": 256},
                                net_param={"transformer_params": [[1, 256], [1, 256]], "append_hidden_shapes": [256, 256]},
                                state_input_shape=(cfg.state_dim,), visual_input_shape=(cfg.stack, 48, 64), grid_size=g)))
        algo.grid_i8_rows = i8
        algo.learn(total_timesteps=2 * n * t)
        buf = algo.rollout_buffer
        assert (buf.grid_i8 is not None) == i8
        if i8:
            s0 = cfg.state_dim
            assert torch.equal(buf.grid_i8.float(), buf.observations[:, :, s0:s0 + g ** 3])
        return [p.detach().clone() for p in algo.policy.parameters()]

    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("g", [16, 20])
def test_learn_with_compact_observations_matches_flat_rows(monkeypatch, g):
    """compact_obs=True (rows = [state | state_rgb] fp32, grid as int8 only): the stored rows equal the flat rows of a
    run without it.  G = 16: both runs use the LDS-staged conv1 kernels (int8 / fp32 slab, same arithmetic) and
    learn() ends with exactly the same parameters; G = 20: the compact run takes the direct int8 conv1 kernels (other
    summation order than the staged fp32 ones), so one iteration is compared with a round-off tolerance."""
    from gennbv_amd.env import synthetic as S
    from gennbv_amd.env.config import TaskConfig
    from gennbv_amd.env.replay_feed import ReplayFeed, ReplayFeedEnv
    from gennbv_amd.network.hybrid_encoder import Hybrid_Encoder
    from gennbv_amd.sb3.policies import ActorCriticPolicy_Train_Eval
    from gennbv_amd.sb3.ppo_grid_obs import PPO_Grid_Obs
    n, t = 8, 4
    monkeypatch.setenv("GENNBV_FUSED_BWD", "0")  # bit equality at G = 16: the separate backward kernels on both sides,
    monkeypatch.setenv("GENNBV_ANALYTIC_BN1", "0")  # BatchNorm-1 statistics from the activations on both sides

    def run(compact: bool):
        torch.manual_seed(0)
        np.random.seed(0)
        cfg = TaskConfig(camera_width=64, camera_height=48, grid_size=g)
        scene = S.make_scenes(n, g, seed=2, device=DEV)
        feed = ReplayFeed.synthetic(scene, cfg, 5, seed=2)
        env = ReplayFeedEnv(cfg, scene, feed, DEV, max_episode_length=6)
        algo = PPO_Grid_Obs(ActorCriticPolicy_Train_Eval, env, learning_rate=1e-4, n_steps=t, batch_size=8, n_epochs=2, ent_coef=0.01,
                            vf_coef=0.8, max_grad_norm=1.0, target_kl=None, seed=1, device=DEV, compact_obs=compact,
                            policy_kwargs=dict(net_arch=[], features_extractor_class=Hybrid_Encoder, features_extractor_kwargs=dict(
                                encoder_param={"hidden_shapes": [256, 256], "visual_dim": 256},
                                net_param={"transformer_params": [[1, 256], [1, 256]], "append_hidden_shapes": [256, 256]},
                                state_input_shape=(cfg.state_dim,), visual_input_shape=(cfg.stack, 48, 64), grid_size=g)))
        algo.grid_i8_rows = False  # (flat rows without the int8 side copy; compact rows always carry the int8 grid)
        algo.learn(total_timesteps=(2 if g == 16 else 1) * n * t)
        buf = algo.rollout_buffer
        s0 = cfg.state_dim
        if compact:
            assert buf.observations.shape[-1] == cfg.obs_dim - g ** 3 and buf.grid_i8 is not None
            rows = torch.cat((buf.observations[..., :s0], buf.grid_i8.float(), buf.observations[..., s0:]), dim=-1)
            mb = next(iter(buf.get(8))).observations  # minibatch view -> the reference's flat rows
            assert mb.materialize().shape == (8, cfg.obs_dim)
        else:
            assert buf.grid_i8 is None
            rows = buf.observations.clone()
        return [p.detach().clone() for p in algo.policy.parameters()], rows

    (pa, ra), (pb, rb) = run(True), run(False)
    assert torch.equal(ra, rb)
    for x, y in zip(pa, pb):
        if g == 16:
            assert torch.equal(x, y)
        else:
            assert torch.allclose(x, y, rtol=0, atol=2e-6), float((x - y).abs().max())


def test_compact_observations_need_an_int8_capable_env():
    from gennbv_amd.network.hybrid_encoder import Hybrid_Encoder
    from gennbv_amd.sb3.policies import ActorCriticPolicy_Train_Eval
    from gennbv_amd.sb3.ppo_grid_obs import PPO_Grid_Obs
    from gennbv_amd.spaces import Box, MultiDiscrete

    class Env:
        num_envs, device = 2, DEV
        observation_space = Box(-np.inf, np.inf, (600 + 8000 + 8192,))
        action_space = MultiDiscrete([81, 81, 51, 1, 13, 13])

    with pytest.raises(ValueError, match="compact_obs"):
        PPO_Grid_Obs(ActorCriticPolicy_Train_Eval, Env(), n_steps=4, batch_size=4, device=DEV, compact_obs=True,
                     policy_kwargs=dict(net_arch=[], features_extractor_class=Hybrid_Encoder, features_extractor_kwargs=dict(
                         encoder_param={}, net_param={"append_hidden_shapes": [256, 256]}, state_input_shape=(600,),
                         visual_input_shape=(2, 64, 64), grid_size=20)))


@pytest.mark.parametrize("first", [True, False])  # True: the reference's `predict_values(new_obs)[0]` (env 0's value for every env)
def test_fused_rollout_add_equals_bootstrap_plus_add(first):
    """gnbv_rollout_add: `rewards += gamma * squeeze(terminal_value * time_outs)` + the five copies of add() in one launch
    leave the buffer rows bit-identical to the reference sequence."""
    from gennbv_amd.sb3.buffers import TensorRolloutBuffer_Grid_Obs
    from gennbv_amd.spaces import Box, MultiDiscrete
    n, t, d = 37, 3, 50
    gen = torch.Generator().manual_seed(3)
    space, aspace = Box(-np.inf, np.inf, (d,)), MultiDiscrete([81, 81, 51, 1, 13, 13])
    bufs = [TensorRolloutBuffer_Grid_Obs(t, space, aspace, device=DEV, n_envs=n) for _ in range(2)]
    gamma = 0.99
    for step in range(t):
        actions = torch.stack([torch.randint(0, k, (n,), generator=gen) for k in (81, 81, 51, 1, 13, 13)], -1).to(DEV)
        rewards = torch.randn(n, generator=gen).to(DEV)
        time_outs = (torch.rand(n, generator=gen) < 0.3).to(DEV)
        tv = torch.randn(n, 1, generator=gen).to(DEV)
        starts = (torch.rand(n, generator=gen) < 0.2).to(DEV)
        values, lp = torch.randn(n, 1, generator=gen).to(DEV), torch.randn(n, generator=gen).to(DEV)
        for b in bufs:
            b.observations[step].normal_(generator=None)
        r_ref = rewards + gamma * torch.squeeze((tv[0] if first else tv) * time_outs.unsqueeze(1), 1)
        bufs[0].add(bufs[0].observations[step], actions, r_ref, starts, values, lp)
        bufs[1].add_bootstrapped(bufs[1].observations[step], actions, rewards, time_outs, tv, gamma, starts, values, lp, broadcast_first=first)
    for name in ("actions", "rewards", "episode_starts", "values", "log_probs"):
        assert torch.equal(getattr(bufs[0], name), getattr(bufs[1], name)), name
    assert bufs[1].step == t and bufs[1].full


@pytest.mark.parametrize("name", ["F9_ppo_train", "F9_ppo_train_earlystop"])
@pytest.mark.parametrize("shard", [False, True])
@pytest.mark.parametrize("mode", ["eager behind a spin kernel", "hipGraph"])
def test_data_parallel_step_replay_order_stress_without_a_process_group(name, shard, mode, monkeypatch):
    """Round 6.  The data-parallel step (`_dp_step_body`: phase A -> exchange issued from the second stream -> conv
    backward -> tail) with a NULL exchange (parallel.attach_null: one rank, no process group, every collective the identity), so it runs in
    the pytest process itself, and must reproduce the reference's F9 like the plain step.  "eager behind a spin kernel": the device is held
    back ~15 ms at the head of every step, so the host enqueues the whole step -- both streams, every allocation and free -- before the
    first kernel runs, which is the order a graph replay has: the use-after-free of round 5 (the pose branch's upstream gradient handed to
    the conv backward's scratch while the second stream still read it) was invisible to eager launches because host time hid it."""
    from gennbv_amd import parallel
    monkeypatch.setenv("GENNBV_FORCE_SHARD", "1" if shard else "0")
    fx = gu.load(name)
    ppo = _ppo_from_fixture(fx, device=DEV, backend="hip")
    parallel.attach_null(ppo)
    ppo.use_graph = mode == "hipGraph"
    ppo.dp_stress_spin_cycles = 0 if ppo.use_graph else 30_000_000
    _check(ppo, fx)
    assert (getattr(ppo._hip["opt"], "shard", None) is not None) == shard
    assert (ppo._hip.get("graph") is not None) == ppo.use_graph
