"""The behaviour-cloning training step, Lightning-free.

Reproduces what Lightning's Trainer does around
``ManiSkill2ACTBCModule.training_step`` (/root/reference/src/models/maniskill2_act_bc_module.py:64-86)
and ``configure_optimizers`` (:347-367) under ``configs/trainer/ddp.yaml``:

    loss = policy(batch)["loss"] / accumulate_grad_batches ; backward            (every micro-batch)
    every `accumulate` micro-batches: clip_grad_norm_(0.5) ; AdamW.step ; OneCycleLR.step ; zero_grad

Data parallel = one process per GPU, ``torch.distributed`` over RCCL (backend "nccl" on ROCm),
DistributedDataParallel with gradient buckets overlapped with backward, ``no_sync()`` on the
non-stepping micro-batches and optional SyncBatchNorm (``sync_batchnorm: true`` in ddp.yaml:9).
Metrics stay on the device and are only read every ``log_every_n_steps`` (ddp.yaml:15) -- the
per-step ``.item()`` would serialise host and GPU.
"""
import contextlib

import torch
import torch.distributed as dist
import torch.nn as nn

from .configs import ACT_OPTIM


def freeze_unused_parameters(policy):
    """``is_pad_head`` is evaluated (act.py:274) but never reaches the loss, so its
    parameters never receive a gradient -- AdamW skips them in the reference (grad is None) and plain
    DDP would raise on them.  Freezing is numerically identical and lets DDP use a static graph."""
    frozen = []
    head = getattr(policy, "is_pad_head", None)
    if head is not None:
        for p in head.parameters():
            p.requires_grad_(False)
            frozen.append(p)
    return frozen


class BCTrainer:
    def __init__(self, policy, total_steps, optim=None, precision="fp32", device=None, distributed=False,
                 sync_batchnorm=True, bucket_cap_mb=32, log_every_n_steps=50):
        o = dict(ACT_OPTIM)
        if optim:
            o.update(optim)
        self.cfg = o
        self.device = torch.device(device) if device is not None else next(policy.parameters()).device
        self.precision = precision
        self.accumulate = int(o["accumulate_grad_batches"])
        self.clip = o["gradient_clip_val"]
        self.log_every_n_steps = log_every_n_steps
        freeze_unused_parameters(policy)
        self.policy = policy
        self.module = policy
        self.distributed = distributed and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if self.distributed:
            if sync_batchnorm:
                policy = nn.SyncBatchNorm.convert_sync_batchnorm(policy)
                self.policy = policy
            ids = [self.device.index] if self.device.type == "cuda" else None
            self.module = nn.parallel.DistributedDataParallel(
                policy, device_ids=ids, gradient_as_bucket_view=True, static_graph=True, bucket_cap_mb=bucket_cap_mb,
            )
        params = [p for p in self.policy.parameters() if p.requires_grad]
        # build_optimizer(cfg, policy, None): one group, every parameter decayed (src/utils/optimizer.py:33-37)
        self.optimizer = torch.optim.AdamW(params, lr=o["lr"], weight_decay=o["weight_decay"],
                                           fused=self.device.type == "cuda")
        self.scheduler = torch.optim.lr_scheduler.OneCycleLR(
            self.optimizer, max_lr=o["lr"], total_steps=max(int(total_steps), int(2 / o["pct_start"]) + 1), pct_start=o["pct_start"],
            anneal_strategy="cos", div_factor=o["div_factor"], final_div_factor=o["final_div_factor"],
        )
        self.micro = 0
        self.optimizer_steps = 0
        self._sums = None
        self._count = 0

    # ------------------------------------------------------------------------------------------
    def _autocast(self):
        if self.precision == "bf16":
            return torch.autocast(device_type=self.device.type, dtype=torch.bfloat16)
        return contextlib.nullcontext()

    def training_step(self, batch):
        """One micro-batch: forward, loss, backward and -- on accumulation boundaries -- the
        optimizer step.  Returns the (detached, on-device) loss dict of this micro-batch."""
        self.module.train()
        stepping = (self.micro + 1) % self.accumulate == 0
        sync_ctx = contextlib.nullcontext()
        if self.distributed and not stepping:
            sync_ctx = self.module.no_sync()
        with sync_ctx:
            with self._autocast():
                out = self.module(batch)
            loss = out["loss"]
            (loss / self.accumulate).backward()
        if stepping:
            if self.clip is not None and self.clip > 0:
                torch.nn.utils.clip_grad_norm_([p for g in self.optimizer.param_groups for p in g["params"]], self.clip)
            self.optimizer.step()
            self.scheduler.step()
            self.optimizer.zero_grad(set_to_none=True)
            self.optimizer_steps += 1
        self.micro += 1
        stats = torch.stack([out["loss"].detach().float(), out["action_loss"].detach().float(),
                             torch.as_tensor(out["kl_loss"], device=loss.device).detach().float()])
        self._sums = stats if self._sums is None else self._sums + stats
        self._count += 1
        return {"loss": stats[0], "action_loss": stats[1], "kl_loss": stats[2]}

    def metrics(self, reset=True):
        """Mean of loss / action_loss / kl_loss since the last call (all-reduced across ranks, like
        ``log_dict(..., sync_dist=True)``).  This is the only host<->device synchronisation."""
        if self._sums is None:
            return {}
        mean = self._sums / max(self._count, 1)
        if self.distributed:
            dist.all_reduce(mean)
            mean = mean / dist.get_world_size()
        vals = mean.tolist()
        if reset:
            self._sums, self._count = None, 0
        return {"train/loss": vals[0], "train/action_loss": vals[1], "train/kl_loss": vals[2]}

    def state_dict(self):
        return {"policy": self.policy.state_dict(), "optimizer": self.optimizer.state_dict(),
                "scheduler": self.scheduler.state_dict(), "micro": self.micro, "optimizer_steps": self.optimizer_steps}

    def load_state_dict(self, sd):
        self.policy.load_state_dict(sd["policy"])
        self.optimizer.load_state_dict(sd["optimizer"])
        self.scheduler.load_state_dict(sd["scheduler"])
        self.micro = sd["micro"]
        self.optimizer_steps = sd["optimizer_steps"]
