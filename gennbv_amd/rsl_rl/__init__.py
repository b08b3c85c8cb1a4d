"""rsl_rl flavour of the PPO path (SURVEY §8 row C-alt): the vendored rsl_rl's storage and algorithm
(rsl_rl/storage/rollout_storage.py, rsl_rl/algorithms/ppo.py) on the gfx950 kernels."""
from .storage import RolloutStorage  # noqa: F401
from .ppo import PPO  # noqa: F401
