"""Behaviour-cloning training path: batches, policy builders, the training step (reference:
src/models/*_bc_module.py + Lightning's loop)."""
from .build import build_act_policy
from .configs import ACT_MODEL, ACT_OPTIM, WORKLOADS
from .synthetic import clone_batch, make_act_batch
from .trainer import BCTrainer

__all__ = ["build_act_policy", "ACT_MODEL", "ACT_OPTIM", "WORKLOADS", "clone_batch", "make_act_batch", "BCTrainer"]
