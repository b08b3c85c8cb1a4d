"""GPU probe (round 5): the one-rank RCCL data-parallel step on fixture F9 under the switches of _dp_step_body
(use_graph x shard, PROBE_REPEAT times): max |logged scalar - fixture| per combination.  Round 6: with PROBE_REPEAT=50 and
GENNBV_DP_SETTLE=0 this is the "200 captures in one process without the settle time" run of VERDICT r5 item 5a
(every algorithm object = eager warm-up collectives + a capture of the RCCL step + its replays)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

from tests import golden_util as gu
from tests.test_policy_ppo_cpu import _ppo_from_fixture
from gennbv_amd import parallel

DEV = "cuda:0"
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
parallel.capture_safe_env()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
fx = gu.load("F9_ppo_train")
KEYS = ("train/entropy_loss", "train/policy_gradient_loss", "train/value_loss", "train/approx_kl", "train/loss")
GRAPHS = [bool(int(x)) for x in os.environ.get("PROBE_GRAPH", "1,0").split(",")]
REPEAT = int(os.environ.get("PROBE_REPEAT", "1"))
for shard in (0, 1) * REPEAT:
    os.environ["GENNBV_FORCE_SHARD"] = str(shard)
    for graph in GRAPHS:
        ppo = _ppo_from_fixture(fx, device=DEV, backend="hip")
        ppo.use_graph = graph
        parallel.attach(ppo, 1, always_sync=True)
        ppo.train()
        torch.cuda.synchronize()
        log = ppo.logger.name_to_value
        err = max(abs(float(log[k]) - float(fx["log/" + k])) / max(1.0, abs(float(fx["log/" + k]))) for k in KEYS)
        N_DONE = globals().get("N_DONE", 0) + 1
        print(f"#{N_DONE} shard={shard} graph={int(graph)}  max rel err {err:.2e}  {'ok' if err <= 1e-4 else 'WRONG'}"
              f"   graph object: {ppo._hip.get('graph') is not None}", flush=True)
        del ppo
print(f"done: {globals().get('N_DONE', 0)} algorithm objects in one process, settle {os.environ.get('GENNBV_DP_SETTLE', 'default')} s", flush=True)
dist.destroy_process_group()
