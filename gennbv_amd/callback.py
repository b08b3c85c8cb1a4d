"""Callbacks of the training loop: the SB3 protocol `PPO_Grid_Obs.learn()` drives
(stable_baselines3/common/callbacks.py BaseCallback / CheckpointCallback) and GenNBV's BestCKPTCallback
(gennbv/callback.py:25-70): a checkpoint every `save_freq` rollouts plus a "best" checkpoint whenever the mean of
an episode-info key over `ep_info_buffer` reaches a new maximum."""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

import torch


class BaseCallback:
    """The hooks learn()/collect_rollouts() call, with SB3's bookkeeping (n_calls, num_timesteps, locals)."""

    def __init__(self, verbose: int = 0):
        self.model = None
        self.n_calls = 0
        self.num_timesteps = 0
        self.verbose = verbose
        self.locals: Dict[str, Any] = {}
        self.globals: Dict[str, Any] = {}

    def init_callback(self, model) -> None:
        self.model = model

    def on_training_start(self, locals_: Dict[str, Any], globals_: Dict[str, Any]) -> None:
        self.locals, self.globals = locals_, globals_
        if self.model is None:
            self.model = locals_.get("self")
        self._on_training_start()

    def on_rollout_start(self) -> None:
        self._on_rollout_start()

    def update_locals(self, locals_: Dict[str, Any]) -> None:
        self.locals.update(locals_)

    def on_step(self) -> bool:
        self.n_calls += 1
        self.num_timesteps = self.model.num_timesteps
        return self._on_step()

    def on_rollout_end(self) -> None:
        self._on_rollout_end()

    def on_training_end(self) -> None:
        self._on_training_end()

    def _on_training_start(self) -> None: ...
    def _on_rollout_start(self) -> None: ...
    def _on_step(self) -> bool: return True
    def _on_rollout_end(self) -> None: ...
    def _on_training_end(self) -> None: ...


class CheckpointCallback(BaseCallback):
    """Save the model every `save_freq` calls of on_step (stable_baselines3/common/callbacks.py CheckpointCallback)."""

    def __init__(self, save_freq: int, save_path: str, name_prefix: str = "rl_model", verbose: int = 0):
        super().__init__(verbose)
        self.save_freq, self.save_path, self.name_prefix = save_freq, save_path, name_prefix

    def _on_training_start(self) -> None:
        if self.save_path is not None:
            os.makedirs(self.save_path, exist_ok=True)

    def _on_step(self) -> bool:
        if self.n_calls % self.save_freq == 0:
            self.model.save(os.path.join(self.save_path, f"{self.name_prefix}_{self.num_timesteps}_steps"))
        return True


class BestCKPTCallback(CheckpointCallback):
    """gennbv/callback.py:25-70.  Per rollout end: the periodic checkpoint (n_calls % save_freq) and, for every key in
    `key_list`, a `<prefix>_best_<key>` checkpoint when mean(ep_info[key]) exceeds the best seen so far."""

    def __init__(self, save_freq: int, save_path: str, name_prefix: str = "rl_model", verbose: int = 0, key_list: Optional[list] = None,
                 best_save_path: Optional[str] = None):
        super().__init__(save_freq, save_path, name_prefix, verbose)
        self.rollout_count = 0
        self.key_highest_value = {k: 0.0 for k in (key_list or [])}
        self._best_save_path = best_save_path

    @property
    def best_save_path(self) -> str:
        # the reference saves into logger.dir; this build's logger has no directory unless one is configured
        return self._best_save_path or getattr(self.model.logger, "dir", None) or self.save_path

    def _on_step(self) -> bool:  # periodic saving happens at rollout end in the reference's subclass
        return True

    def _on_rollout_end(self) -> None:
        if self.n_calls % self.save_freq == 0:
            self.model.save(os.path.join(self.save_path, f"{self.name_prefix}_{self.num_timesteps}_steps"))
        for key in self.key_highest_value:
            value = self.calculate_value(key)
            if value > self.key_highest_value[key]:
                self.key_highest_value[key] = value
                path = os.path.join(self.best_save_path, f"{self.name_prefix}_best_{key}")
                self.model.save(path)
                print(f"Saving Best {key} checkpoint to {path}: {value}")

    def calculate_value(self, key: str) -> float:
        buf = [e for e in (self.model.ep_info_buffer or []) if e is not None]
        assert len(buf) > 0 and key in buf[0], f"no key named {key}, can not save checkpoint"
        vals = [torch.as_tensor(e[key], dtype=torch.float32).reshape(-1).cpu() for e in buf]
        return float(torch.cat(vals).mean())


class ReconstructionCallBack(BestCKPTCallback):
    pass
