"""Behaviour-cloning training path: batches, policy builders, the training step (reference:
src/models/*_bc_module.py + Lightning's loop)."""
from .build import build_act_policy, build_dp_policy, build_rlbench_act_policy
from .configs import (ACT_MODEL, ACT_OPTIM, DP_MODEL, DP_OPTIM, RLBENCH_ACT_MODEL, RLBENCH_ACT_OPTIM, RLBENCH_DP_MODEL, RLBENCH_DP_OPTIM,
                      WORKLOADS)
from .synthetic import clone_batch, make_act_batch, make_dp_batch
from .trainer import BCTrainer

__all__ = ["build_act_policy", "build_dp_policy", "make_dp_batch", "DP_MODEL", "DP_OPTIM", "ACT_MODEL", "ACT_OPTIM", "WORKLOADS", "clone_batch", "make_act_batch", "BCTrainer"]
