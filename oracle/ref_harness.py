"""Import the *reference* (zjwzcx/GenNBV, read-only at /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY -- works only in the build container where
/root/reference exists.  Nothing on the product path, no `-m gpu` test, not
`smoke()` and not `bench.py` may import this module: the reference cannot
travel to the GPU box.  Its single consumer is `oracle/gen_golden.py`, which
runs the reference's own Python on seeded inputs and writes small golden
fixtures under `tests/golden/`.

The reference needs Isaac Gym, gym, pycuda, open3d, torchvision, pytorch3d,
tensorboard, wandb and cv2, none of which exist here.  We register permissive
stub modules for them so that the hot-path files import:

  gennbv/utils.py, gennbv/env/env_train_gennbv.py, gennbv/env/env_train_base.py,
  gennbv/network/hybrid_encoder.py, gennbv/wrapper/env_wrapper_gennbv_train.py,
  stable_baselines3/ppo/ppo_grid_obs.py, stable_baselines3/common/{buffers,policies,
  distributions,on_policy_algorithm_grid_obs}.py, rsl_rl/{storage,algorithms}.

No reference source is copied: the stubs only model the *third-party* APIs the
reference imports (gym.spaces etc.), written from their public behaviour.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("GENNBV_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "gennbv"))


# --------------------------------------------------------------------------- #
# permissive stub module
# --------------------------------------------------------------------------- #
class _Anything:
    """Callable, attribute-able placeholder for third-party symbols."""

    def __init__(self, name="stub"):
        self.__dict__["_name"] = name

    def __call__(self, *a, **k):
        return _Anything(self._name + "()")

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Anything(self._name + "." + item)

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)

    def __repr__(self):
        return f"<stub {self._name}>"


class _StubModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        val = _Anything(self.__name__ + "." + item)
        return val


def _stub(name: str, **attrs) -> types.ModuleType:
    mod = _StubModule(name)
    mod.__path__ = []  # behave like a package so sub-imports resolve
    mod.__all__ = []
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, mod)
    return mod


# --------------------------------------------------------------------------- #
# minimal `gym` (public gym 0.21-0.23 semantics the reference relies on)
# --------------------------------------------------------------------------- #
def _install_gym():
    import numpy as np

    class Space:
        def __init__(self, shape=None, dtype=None, seed=None):
            self._shape = None if shape is None else tuple(shape)
            self.dtype = None if dtype is None else np.dtype(dtype)
            self._np_random = np.random.RandomState(seed)

        @property
        def shape(self):
            return self._shape

        def seed(self, seed=None):
            self._np_random = np.random.RandomState(seed)
            return [seed]

        def contains(self, x):
            return True

        def __contains__(self, x):
            return self.contains(x)

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            if shape is None:
                shape = np.asarray(low).shape if np.ndim(low) else np.asarray(high).shape
            shape = tuple(int(s) for s in shape)
            try:
                low = float(low) if np.isscalar(low) or not np.ndim(low) else low
            except Exception:
                pass
            self.low = np.full(shape, low, dtype=np.float64) if np.isscalar(low) else np.broadcast_to(
                np.asarray(low, dtype=np.float64), shape).copy()
            self.high = np.full(shape, high, dtype=np.float64) if np.isscalar(high) else np.broadcast_to(
                np.asarray(high, dtype=np.float64), shape).copy()
            with np.errstate(all="ignore"):
                self.low = self.low.astype(dtype)
                self.high = self.high.astype(dtype)
            self.bounded_below = -np.inf < self.low
            self.bounded_above = np.inf > self.high
            super().__init__(shape, dtype, seed)

        def sample(self):
            return self._np_random.uniform(size=self.shape).astype(self.dtype)

        def __repr__(self):
            return f"Box({self.shape}, {self.dtype})"

    class Discrete(Space):
        def __init__(self, n, seed=None):
            self.n = int(n)
            super().__init__((), np.int64, seed)

        def sample(self):
            return self._np_random.randint(self.n)

    class MultiDiscrete(Space):
        def __init__(self, nvec, dtype=np.int64, seed=None):
            self.nvec = np.asarray(nvec, dtype=dtype)
            super().__init__(self.nvec.shape, dtype, seed)

        def sample(self):
            return (self._np_random.random_sample(self.nvec.shape) * self.nvec).astype(self.dtype)

    class MultiBinary(Space):
        def __init__(self, n, seed=None):
            self.n = n
            super().__init__((n,) if np.isscalar(n) else tuple(n), np.int8, seed)

    class Dict(Space):
        def __init__(self, spaces=None, seed=None, **kw):
            self.spaces = dict(spaces or {}, **kw)
            super().__init__(None, None, seed)

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

        def items(self):
            return self.spaces.items()

    class Tuple(Space):
        def __init__(self, spaces, seed=None):
            self.spaces = tuple(spaces)
            super().__init__(None, None, seed)

    class Env:
        metadata = {}
        reward_range = (-float("inf"), float("inf"))
        spec = None
        action_space = None
        observation_space = None

        @property
        def unwrapped(self):
            return self

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)
            return getattr(self.env, name)

    class ObservationWrapper(Wrapper):
        pass

    class RewardWrapper(Wrapper):
        pass

    class ActionWrapper(Wrapper):
        pass

    class GoalEnv(Env):
        pass

    spaces = _stub("gym.spaces") if "gym" in sys.modules else None
    gym = _stub("gym", Env=Env, Wrapper=Wrapper, ObservationWrapper=ObservationWrapper,
                RewardWrapper=RewardWrapper, ActionWrapper=ActionWrapper, GoalEnv=GoalEnv,
                Space=Space, __version__="0.21.0")
    spaces = _stub("gym.spaces", Space=Space, Box=Box, Discrete=Discrete, MultiDiscrete=MultiDiscrete,
                   MultiBinary=MultiBinary, Dict=Dict, Tuple=Tuple)
    gym.spaces = spaces
    _stub("gym.envs")
    _stub("gym.envs.registration", EnvSpec=type("EnvSpec", (), {}))
    _stub("gym.wrappers")
    _stub("gym.wrappers.monitoring")
    _stub("gym.wrappers.monitoring.video_recorder")
    _stub("gym.utils")
    _stub("gym.utils.seeding")
    return gym


_INSTALLED = False


def install_stubs() -> None:
    """Register every stub; idempotent.  torch must be imported *before*."""
    global _INSTALLED
    if _INSTALLED:
        return
    import numpy as np
    import torch
    import torch.fx  # noqa: F401  (a stub inserted earlier would break torch's own import)

    _install_gym()

    # isaacgym: `from isaacgym.torch_utils import *` must deliver np / torch
    iso = _stub("isaacgym")
    _stub("isaacgym.gymapi")
    _stub("isaacgym.gymtorch")
    _stub("isaacgym.gymutil")
    _stub("isaacgym.terrain_utils")
    tu = _stub("isaacgym.torch_utils", np=np, torch=torch)
    tu.__all__ = ["np", "torch"]
    iso.__all__ = []

    # tensorboard
    if "torch.utils.tensorboard" not in sys.modules:
        tb = _stub("torch.utils.tensorboard", SummaryWriter=_Anything("SummaryWriter"))
        torch.utils.tensorboard = tb
    _stub("tensorboard")
    # pycuda
    _stub("pycuda")
    _stub("pycuda.driver")
    _stub("pycuda.autoinit")
    _stub("pycuda.compiler", SourceModule=_Anything("SourceModule"))
    # open3d / PIL may be missing
    _stub("open3d")
    try:
        import PIL  # noqa: F401
        import PIL.Image  # noqa: F401
    except Exception:
        _stub("PIL")
        _stub("PIL.Image")
    # torchvision (rgb_to_grayscale is therefore UNPINNED)
    _stub("torchvision")
    _stub("torchvision.transforms", ToPILImage=_Anything("ToPILImage"))
    _stub("torchvision.transforms.functional")
    _stub("pytorch3d")
    _stub("pytorch3d.loss")
    _stub("wandb")
    _stub("cv2")
    _stub("matplotlib") if importlib.util.find_spec("matplotlib") is None else None
    if importlib.util.find_spec("matplotlib") is None:
        _stub("matplotlib.pyplot")
    _INSTALLED = True


def import_reference():
    """Returns a namespace with the reference modules the oracle pins against."""
    if not reference_available():
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    import torch  # noqa: F401
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # import order matters: gennbv first, exactly like train_gennbv.py:3
    import gennbv  # noqa: F401
    ns = types.SimpleNamespace()
    ns.utils = importlib.import_module("gennbv.utils")
    ns.env_train = importlib.import_module("gennbv.env.env_train_gennbv")
    ns.env_base = importlib.import_module("gennbv.env.env_train_base")
    ns.hybrid_encoder = importlib.import_module("gennbv.network.hybrid_encoder")
    ns.wrapper = importlib.import_module("gennbv.wrapper.env_wrapper_gennbv_train")
    ns.buffers = importlib.import_module("stable_baselines3.common.buffers")
    ns.policies = importlib.import_module("stable_baselines3.common.policies")
    ns.distributions = importlib.import_module("stable_baselines3.common.distributions")
    ns.ppo_grid_obs = importlib.import_module("stable_baselines3.ppo.ppo_grid_obs")
    ns.on_policy = importlib.import_module("stable_baselines3.common.on_policy_algorithm_grid_obs")
    ns.rsl_storage = importlib.import_module("rsl_rl.storage.rollout_storage")
    ns.rsl_ppo = importlib.import_module("rsl_rl.algorithms.ppo")
    ns.callback = importlib.import_module("gennbv.callback")
    ns.evaluation = importlib.import_module("stable_baselines3.common.evaluation")
    return ns


if __name__ == "__main__":
    ref = import_reference()
    print("reference imported:", [k for k in vars(ref)])
