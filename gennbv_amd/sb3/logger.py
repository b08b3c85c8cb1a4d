"""Tiny key-value logger with the `record` / `dump` surface PPO_Grid_Obs.train()/learn() use
(the reference's stable_baselines3/common/logger.py is observability, out of scope)."""
from __future__ import annotations

from typing import Any, Dict


class Logger:
    def __init__(self, verbose: int = 0):
        self.verbose = verbose
        self.name_to_value: Dict[str, Any] = {}
        self.history = []

    def record(self, key: str, value: Any, exclude=None) -> None:
        self.name_to_value[key] = value

    def dump(self, step: int = 0) -> None:
        self.history.append((step, dict(self.name_to_value)))
        if self.verbose >= 1:
            print(f"[step {step}] " + " ".join(f"{k}={v:.5g}" if isinstance(v, float) else f"{k}={v}"
                                                for k, v in sorted(self.name_to_value.items())))
        self.name_to_value = {}
