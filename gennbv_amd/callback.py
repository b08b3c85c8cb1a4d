"""Callbacks of the training loop: the SB3 protocol `PPO_Grid_Obs.learn()` drives
(stable_baselines3/common/callbacks.py BaseCallback / CheckpointCallback) and GenNBV's BestCKPTCallback
(gennbv/callback.py:25-70): a checkpoint every `save_freq` rollouts plus a "best" checkpoint whenever the mean of
an episode-info key over `ep_info_buffer` reaches a new maximum; and the evaluation callback the training script
installs, EvalCallback_Grid_Obs (stable_baselines3/common/callbacks.py:473-708) on the EventCallback base (:119-158)."""
from __future__ import annotations

import os
import warnings
from typing import Any, Dict, Optional

import numpy as np
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
        self.logger = None
        self.training_env = None
        self.parent: Optional["BaseCallback"] = None  # set by an EventCallback on the callbacks it triggers

    def init_callback(self, model) -> None:
        self.model = model
        self.training_env = model.get_env() if hasattr(model, "get_env") else getattr(model, "env", None)
        self.logger = getattr(model, "logger", None)
        self._init_callback()

    def _init_callback(self) -> None: ...

    def on_training_start(self, locals_: Dict[str, Any], globals_: Dict[str, Any]) -> None:
        self.locals, self.globals = locals_, globals_
        if self.model is None:
            self.model = locals_.get("self")
        self._on_training_start()

    def on_rollout_start(self) -> None:
        self._on_rollout_start()

    def update_locals(self, locals_: Dict[str, Any]) -> None:
        self.locals.update(locals_)
        self.update_child_locals(locals_)

    def update_child_locals(self, locals_: Dict[str, Any]) -> None: ...

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


class CallbackList(BaseCallback):
    """stable_baselines3/common/callbacks.py:161-207: callbacks called one after the other; training stops if any says so."""

    def __init__(self, callbacks: list):
        super().__init__()
        assert isinstance(callbacks, list)
        self.callbacks = callbacks

    def _init_callback(self) -> None:
        for cb in self.callbacks:
            cb.init_callback(self.model)

    def _on_training_start(self) -> None:
        for cb in self.callbacks:
            cb.on_training_start(self.locals, self.globals)

    def _on_rollout_start(self) -> None:
        for cb in self.callbacks:
            cb.on_rollout_start()

    def _on_step(self) -> bool:
        go_on = True
        for cb in self.callbacks:
            go_on = cb.on_step() and go_on  # every callback runs, also after one asked to stop
        return go_on

    def _on_rollout_end(self) -> None:
        for cb in self.callbacks:
            cb.on_rollout_end()

    def _on_training_end(self) -> None:
        for cb in self.callbacks:
            cb.on_training_end()

    def update_child_locals(self, locals_: Dict[str, Any]) -> None:
        for cb in self.callbacks:
            cb.update_locals(locals_)


class ConvertCallback(BaseCallback):
    """A plain function `f(locals, globals) -> bool | None` as a callback (callbacks.py ConvertCallback)."""

    def __init__(self, fn, verbose: int = 0):
        super().__init__(verbose)
        self.fn = fn

    def _on_step(self) -> bool:
        return True if self.fn is None else self.fn(self.locals, self.globals) is not False


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


class EventCallback(BaseCallback):
    """stable_baselines3/common/callbacks.py:119-158: a callback that triggers a child callback on some event."""

    def __init__(self, callback: Optional[BaseCallback] = None, verbose: int = 0):
        super().__init__(verbose)
        self.callback = callback
        if callback is not None:
            callback.parent = self

    def init_callback(self, model) -> None:
        super().init_callback(model)
        if self.callback is not None:
            self.callback.init_callback(self.model)

    def _on_training_start(self) -> None:
        if self.callback is not None:
            self.callback.on_training_start(self.locals, self.globals)

    def _on_event(self) -> bool:
        return self.callback.on_step() if self.callback is not None else True

    def update_child_locals(self, locals_: Dict[str, Any]) -> None:
        if self.callback is not None:
            self.callback.update_locals(locals_)


class EvalCallback_Grid_Obs(EventCallback):  # noqa: N801 (the reference's name)
    """stable_baselines3/common/callbacks.py:473-708.  Every `eval_freq` calls of on_step: run
    `evaluate_policy_grid_obs(model, eval_env, n_eval_episodes)` with the AUC and the Chamfer accuracy (:620-631),
    append (num_timesteps, episode rewards, episode lengths[, successes]) to `<log_path>/evaluations.npz` (:633-651),
    record eval/mean_reward, eval/mean_AUC (mean over envs of the per-env mean AUC), eval/mean_accuracy,
    eval/mean_ep_length[, eval/success_rate] and time/total_timesteps, dump the logger at num_timesteps (:667-681), save
    `<best_model_save_path>/best_model` and trigger `callback_on_new_best` on a strictly better mean reward (:683-691),
    then trigger `callback_after_eval` (:694-695).  Returns False (stop training) when a triggered callback does.

    `eval_kwargs` go to evaluate_policy_grid_obs (max_length / accuracy_fn of a tensor env; the reference hard-codes 50
    envs x 30 steps, evaluation.py:201-202)."""

    def __init__(self, eval_env, callback_on_new_best: Optional[BaseCallback] = None, callback_after_eval: Optional[BaseCallback] = None,
                 n_eval_episodes: int = 5, eval_freq: int = 10000, log_path: Optional[str] = None,
                 best_model_save_path: Optional[str] = None, deterministic: bool = True, render: bool = False, verbose: int = 1,
                 warn: bool = True, eval_kwargs: Optional[dict] = None):
        super().__init__(callback_after_eval, verbose=verbose)
        self.callback_on_new_best = callback_on_new_best
        if callback_on_new_best is not None:
            callback_on_new_best.parent = self
        self.n_eval_episodes, self.eval_freq = n_eval_episodes, eval_freq
        self.best_mean_reward = self.last_mean_reward = -np.inf
        self.deterministic, self.render, self.warn = deterministic, render, warn
        self.eval_env = eval_env
        self.best_model_save_path = best_model_save_path
        self.log_path = os.path.join(log_path, "evaluations") if log_path is not None else None
        self.evaluations_results, self.evaluations_timesteps, self.evaluations_length = [], [], []
        self._is_success_buffer, self.evaluations_successes = [], []
        self.eval_kwargs = dict(eval_kwargs or {})

    def _init_callback(self) -> None:
        if self.warn and not isinstance(self.training_env, type(self.eval_env)):
            warnings.warn(f"Training and eval env are not of the same type{self.training_env} != {self.eval_env}")
        if self.best_model_save_path is not None:
            os.makedirs(self.best_model_save_path, exist_ok=True)
        if self.log_path is not None:
            os.makedirs(os.path.dirname(self.log_path), exist_ok=True)
        if self.callback_on_new_best is not None:
            self.callback_on_new_best.init_callback(self.model)

    def _log_success_callback(self, locals_: Dict[str, Any], globals_: Dict[str, Any]) -> None:
        if locals_["done"]:
            ok = locals_["info"].get("is_success")
            if ok is not None:
                self._is_success_buffer.append(ok)

    def _on_step(self) -> bool:
        if not (self.eval_freq > 0 and self.n_calls % self.eval_freq == 0):
            return True
        from .eval.evaluate import evaluate_policy_grid_obs
        self._is_success_buffer = []
        ep_r, ep_l, mean_auc, ep_acc = evaluate_policy_grid_obs(
            self.model, self.eval_env, n_eval_episodes=self.n_eval_episodes, deterministic=self.deterministic, return_AUC=True,
            callback=self._log_success_callback, **self.eval_kwargs)
        if self.log_path is not None:
            self.evaluations_timesteps.append(self.num_timesteps)
            self.evaluations_results.append(ep_r)
            self.evaluations_length.append(ep_l)
            extra = {}
            if len(self._is_success_buffer) > 0:
                self.evaluations_successes.append(self._is_success_buffer)
                extra = dict(successes=self.evaluations_successes)
            np.savez(self.log_path, timesteps=self.evaluations_timesteps, results=self.evaluations_results,
                     ep_lengths=self.evaluations_length, **extra)
        mean_reward, std_reward = np.mean(ep_r), np.std(ep_r)
        mean_len, std_len = np.mean(ep_l), np.std(ep_l)
        self.last_mean_reward = mean_reward
        if self.verbose > 0:
            print(f"Eval num_timesteps={self.num_timesteps}, episode_reward={mean_reward:.2f} +/- {std_reward:.2f}")
            print(f"Episode length: {mean_len:.2f} +/- {std_len:.2f}")
        self.logger.record("eval/mean_reward", float(mean_reward))
        self.logger.record("eval/mean_AUC", float(np.mean(mean_auc.numpy())))
        self.logger.record("eval/mean_accuracy", float(np.mean(ep_acc)))
        self.logger.record("eval/mean_ep_length", mean_len)
        if len(self._is_success_buffer) > 0:
            rate = np.mean(self._is_success_buffer)
            if self.verbose > 0:
                print(f"Success rate: {100 * rate:.2f}%")
            self.logger.record("eval/success_rate", rate)
        self.logger.record("time/total_timesteps", self.num_timesteps, exclude="tensorboard")
        self.logger.dump(self.num_timesteps)
        go_on = True
        if mean_reward > self.best_mean_reward:
            if self.verbose > 0:
                print("New best mean reward!")
            if self.best_model_save_path is not None:
                self.model.save(os.path.join(self.best_model_save_path, "best_model"))
            self.best_mean_reward = mean_reward
            if self.callback_on_new_best is not None:
                go_on = self.callback_on_new_best.on_step()
        if self.callback is not None:
            go_on = go_on and self._on_event()
        return go_on

    def update_child_locals(self, locals_: Dict[str, Any]) -> None:
        if self.callback:
            self.callback.update_locals(locals_)
