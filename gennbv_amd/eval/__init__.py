"""Evaluation path: Chamfer-distance accuracy, AUC of the coverage curve and the episode loop
of stable_baselines3/common/evaluation.py:136-378 for tensor envs."""
from .metrics import chamfer_distance, unique_rounded_points, reconstruction_accuracy_cm, auc_update, mean_auc  # noqa: F401
from .evaluate import evaluate_policy_grid_obs  # noqa: F401
