"""CPU, world_size 2 over gloo: the data-parallel PPO update (gennbv_amd/parallel.py).

Two ranks own two envs each of the F9 fixture's recorded rollout; with BatchNorm in eval mode
and advantage normalisation off (the two per-minibatch statistics that stay rank-local by
design) the averaged-gradient update must equal the single-process update on the concatenated
global minibatches, the ranks must end bit-identical, and the KL early stop must trigger at
the same minibatch on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import golden_util as gu
from tests.test_policy_ppo_cpu import _ppo_from_fixture


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _freeze_bn(ppo):
    orig = ppo.policy.set_training_mode

    def mode(flag):
        orig(flag)
        ppo.policy.features_extractor.naive_encoder_grid.eval()
    ppo.policy.set_training_mode = mode


def _local_ppo(fx, envs, batch, target_kl):
    """PPO over the columns `envs` of the fixture's [T, N] buffer."""
    t = int(fx["T"])
    sub = {k: fx[k] for k in fx.files}
    ppo = _ppo_from_fixture(fx)
    from gennbv_amd.sb3.buffers import TensorRolloutBuffer_Grid_Obs
    full = ppo.rollout_buffer
    buf = TensorRolloutBuffer_Grid_Obs(t, ppo.observation_space, ppo.action_space, device="cpu", gamma=0.99,
                                       gae_lambda=0.95, n_envs=len(envs))
    for name in ("observations", "actions", "values", "log_probs", "rewards", "advantages", "returns"):
        src = getattr(full, name)
        getattr(buf, name)[: src.shape[0]].copy_(src[:, envs])
    buf.step = t
    ppo.rollout_buffer, ppo.n_envs, ppo.batch_size = buf, len(envs), batch
    # advantage normalisation ON: its statistics are those of the global minibatch (GradSync.global_adv_norm); BatchNorm is
    # frozen here because the torch modules of this CPU path have no cross-rank BatchNorm -- the fused gfx950 path has, and
    # tests/test_parallel_gpu.py runs it with BatchNorm in train mode
    ppo.normalize_advantage, ppo.target_kl = True, target_kl
    _freeze_bn(ppo)
    return ppo


def _worker(rank, world, port, target_kl, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from gennbv_amd import parallel
    fx = gu.load("F9_ppo_train")
    t = int(fx["T"])
    envs = [2 * rank, 2 * rank + 1]
    ppo = _local_ppo(fx, envs, batch=4, target_kl=target_kl)
    np.random.seed(5)
    ppo.rollout_buffer.indices = np.random.permutation(t * 2)  # same permutation on every rank
    ppo.rollout_buffer._indices_dev = None
    parallel.attach(ppo, world)
    ppo.train()
    vec = torch.cat([p.detach().reshape(-1) for p in ppo.policy.parameters()])
    gathered = [torch.zeros_like(vec) for _ in range(world)]
    dist.all_gather(gathered, vec)
    if rank == 0:
        out["identical"] = bool(torch.equal(gathered[0], gathered[1]))
        out["params"] = vec.numpy()
        out["n_rows"] = len(ppo.last_train_stats)
    dist.destroy_process_group()


def _single(target_kl):
    fx = gu.load("F9_ppo_train")
    t = int(fx["T"])
    ppo = _local_ppo(fx, [0, 1, 2, 3], batch=8, target_kl=target_kl)
    np.random.seed(5)
    perm = np.random.permutation(t * 2)
    # local index i = j*T + step of rank r  ->  global index (2r + j)*T + step; global minibatch k =
    # concat over ranks of the ranks' minibatch k
    glob = []
    for k in range(len(perm) // 4):
        for r in range(2):
            loc = perm[4 * k: 4 * k + 4]
            glob.extend(((2 * r + loc // t) * t + loc % t).tolist())
    ppo.rollout_buffer.indices = np.array(glob)
    ppo.rollout_buffer._indices_dev = None
    ppo.train()
    return torch.cat([p.detach().reshape(-1) for p in ppo.policy.parameters()]).numpy(), len(ppo.last_train_stats)


@pytest.mark.parametrize("target_kl", [None, 0.05])
def test_two_rank_update_equals_global_batch_update(target_kl):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), target_kl, out), nprocs=2, join=True)
    assert out["identical"], "ranks diverged"
    ref, rows = _single(target_kl)
    assert out["n_rows"] == rows  # same number of executed minibatches (early-stop position)
    np.testing.assert_allclose(out["params"], ref, rtol=2e-4, atol=2e-6)


def test_shard_range():
    from gennbv_amd.parallel import shard_range
    assert [shard_range(2048, r, 8) for r in (0, 7)] == [(0, 256), (1792, 2048)]
    with pytest.raises(AssertionError):
        shard_range(10, 0, 4)
