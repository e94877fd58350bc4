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


class _ShadowParam(torch.autograd.Function):
    """Forward: hand out the bf16 mirror of an fp32 master weight (no cast kernel).  Backward: pass
    the bf16 gradient to the flat optimizer, which folds it into the fp32 gradient buffer."""

    @staticmethod
    def forward(ctx, master, shadow, opt, k):
        ctx.opt, ctx.k = opt, k
        return shadow.view_as(shadow)

    @staticmethod
    def backward(ctx, g):
        ctx.opt.stash_grad(ctx.k, g)
        return None, None, None, None


def bf16_consumed_parameters(policy, fused_ffn=False):
    """ids of the parameters that autocast would cast to bf16 on every use: weights / biases of Linear,
    attention projections and 1-D convolutions.  The SA layer's own `linear` is excluded: its xyz
    columns are consumed in fp32 by the fused kernel."""
    from .. import _lib
    from ..policy.transformer import TransformerDecoderLayer, TransformerEncoderLayer

    ids, skip = set(), set()
    if fused_ffn:  # the fused feed-forward kernel (csrc/ffn.hip) consumes the fp32 masters directly
        for mod in policy.modules():
            if isinstance(mod, (TransformerEncoderLayer, TransformerDecoderLayer)) and mod.activation is torch.nn.functional.relu \
                    and _lib.load().pcm_ffn_ln_supported(mod.linear1.in_features, mod.linear1.out_features) and not mod.normalize_before:
                skip.update(id(p) for p in list(mod.linear1.parameters()) + list(mod.linear2.parameters()))
    for name, mod in policy.named_modules():
        if isinstance(mod, (nn.Linear, nn.Conv1d, nn.ConvTranspose1d)):
            if any(id(p) in skip for p in mod.parameters(recurse=False)):
                continue
            if name.split(".")[-1] == "linear" and hasattr(policy.get_submodule(name.rsplit(".", 1)[0]) if "." in name else policy, "pcd_nsample"):
                continue
            ids.update(id(p) for p in mod.parameters(recurse=False))
        elif isinstance(mod, nn.MultiheadAttention):
            ids.update(id(p) for p in (mod.in_proj_weight, mod.in_proj_bias) if p is not None)
    return ids


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
    """mode:
      "eager"  torch.optim.AdamW + OneCycleLR, DistributedDataParallel (+ SyncBatchNorm) when
               distributed -- the literal Lightning recipe; also the CPU path.
      "flat"   FlatAdamW (csrc/optim.hip) + host OneCycle; data parallel = ONE all-reduce of the flat
               gradient buffer per optimizer step (no DDP wrapper, no bucket hooks).
      "graph"  "flat" with forward+backward of a micro-batch captured ONCE into a hipGraph and
               replayed (the step is launch-bound in eager mode: ~2400 launches); needs static
               shapes (equal-size clouds) and per-rank BatchNorm statistics.
    """

    def __init__(self, policy, total_steps, optim=None, precision="fp32", device=None, distributed=False,
                 sync_batchnorm=True, bucket_cap_mb=32, log_every_n_steps=50, mode="eager", flat_optimizer_cls=None):
        o = dict(ACT_OPTIM)
        if optim:
            o.update(optim)
        self.cfg = o
        self.device = torch.device(device) if device is not None else next(policy.parameters()).device
        self.precision = precision
        self.accumulate = int(o["accumulate_grad_batches"])
        self.clip = o["gradient_clip_val"]
        self.log_every_n_steps = log_every_n_steps
        if mode not in ("eager", "flat", "graph", "hybrid"):
            raise ValueError(mode)
        if self.device.type != "cuda" and mode != "eager" and (flat_optimizer_cls is None or mode == "graph"):
            raise ValueError("flat/graph modes run on the HIP device only")
        self.mode = mode
        freeze_unused_parameters(policy)
        self.policy = policy
        self.module = policy
        self.distributed = distributed and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.distributed else 1
        total_steps = max(int(total_steps), int(2 / o["pct_start"]) + 1)
        if self.distributed and sync_batchnorm and mode != "graph":
            policy = nn.SyncBatchNorm.convert_sync_batchnorm(policy)
            self.policy = self.module = policy
        self.sync_batchnorm = bool(self.distributed and sync_batchnorm and mode != "graph")
        # fused transformer tail ops (csrc/drln.hip, ffn.hip) need a device-resident dropout seed: flat / graph modes
        self._fused_ctx = None
        if mode != "eager" and self.device.type == "cuda":
            from ..policy.fused_ops import FusedContext

            self._fused_ctx = FusedContext(self.device)
        params = [p for p in self.policy.parameters() if p.requires_grad]
        betas = tuple(o.get("betas", (0.9, 0.999)))
        if o.get("filter_bias_and_bn", False) and o["weight_decay"]:
            # build_optimizer_v2 -> param_groups_weight_decay (src/utils/optimizer.py:152-170, 296-300)
            no_decay = [p for n, p in self.policy.named_parameters() if p.requires_grad and (p.ndim <= 1 or n.endswith(".bias"))]
            ids = {id(p) for p in no_decay}
            decay = [p for p in params if id(p) not in ids]
            params = [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": o["weight_decay"]}]
        if mode == "eager":
            if self.distributed:
                ids = [self.device.index] if self.device.type == "cuda" else None
                self.module = nn.parallel.DistributedDataParallel(
                    self.policy, device_ids=ids, gradient_as_bucket_view=True, bucket_cap_mb=bucket_cap_mb,
                )
            # build_optimizer(cfg, policy, None): one group, every parameter decayed (src/utils/optimizer.py:33-37)
            self.optimizer = torch.optim.AdamW(params, lr=o["lr"], weight_decay=o["weight_decay"], betas=betas,
                                               fused=self.device.type == "cuda")
            self.scheduler = torch.optim.lr_scheduler.OneCycleLR(
                self.optimizer, max_lr=o["lr"], total_steps=total_steps, pct_start=o["pct_start"],
                anneal_strategy="cos", div_factor=o["div_factor"], final_div_factor=o["final_div_factor"],
            )
        else:
            from .schedule import OneCycle

            if flat_optimizer_cls is not None:  # tests inject a host stand-in to exercise the DP logic on gloo
                FlatAdamW = flat_optimizer_cls
            else:
                from .flat_optim import FlatAdamW

            sched = OneCycle(o["lr"], total_steps, o["pct_start"], o["div_factor"], o["final_div_factor"])
            # grad_scale 1/world turns the all-reduce SUM into DDP's mean inside the Adam kernel
            self.optimizer = FlatAdamW(params, sched, betas=betas, weight_decay=o["weight_decay"], max_norm=self.clip or 0.0,
                                       grad_scale=1.0 / self.world)
            self.scheduler = sched
            self._shadow_names = None
            if precision == "bf16" and hasattr(self.optimizer, "enable_bf16_mirror"):
                self.optimizer.enable_bf16_mirror(bf16_consumed_parameters(self.policy, fused_ffn=self._fused_ctx is not None))
                index = {id(p): k for k, p in enumerate(self.optimizer.params)}
                self._shadow_names = [(n, p, index[id(p)]) for n, p in self.policy.named_parameters()
                                      if id(p) in index and self.optimizer.shadow[index[id(p)]] is not None]
        self.micro = 0
        self.optimizer_steps = 0
        self._sums = None
        self._count = 0
        self._graph = None
        self._graph_acc = None
        self._static_batch = None
        self._static_stats = None
        self._static_sig = None

    # ------------------------------------------------------------------------------------------
    def _autocast(self):
        if self.precision == "bf16":
            return torch.autocast(device_type=self.device.type, dtype=torch.bfloat16)
        return contextlib.nullcontext()

    # ---- hybrid mode: eager tokenizer (ragged point clouds) + ONE hipGraph for everything behind the token matrix ------
    def _shadow_repl(self, only=None):
        shadows = getattr(self, "_shadow_names", None)
        if not shadows:
            return None
        opt = self.optimizer
        return {n: _ShadowParam.apply(p, opt.shadow[k], opt, k) for n, p, k in shadows if only is None or k in only}

    def _call_policy(self, batch, only=None, **kwargs):
        repl = self._shadow_repl(only)
        if repl is not None:
            return torch.func.functional_call(self.policy, repl, (batch,), kwargs)
        return self.module(batch, **kwargs)

    def _hybrid_setup(self, batch):
        """Split the parameters by stage, build the static inputs of the captured half and capture it."""
        from ..policy import fused_ops
        from .synthetic import clone_batch

        opt = self.optimizer
        index = {id(p): k for k, p in enumerate(opt.params)}
        tok = sorted(index[id(p)] for p in self.policy.tokenizer_parameters() if id(p) in index)
        tok_set = set(tok)
        self._subset_a, self._subset_b = tok, [k for k in range(len(opt.params)) if k not in tok_set]
        self._subset_a_set = tok_set
        buffers = {n: b.clone() for n, b in self.policy.named_buffers()}  # nothing below may count as training
        with fused_ops.activate(self._fused_ctx), self._autocast(), torch.no_grad():
            ragged, rest = self.policy.hybrid_split(clone_batch(batch))
            boundary = tuple(self._call_policy(ragged, stage="tokenize"))  # shapes / dtypes of the boundary
        self._static_sig = self._signature(rest)
        self._static_batch = self._clone_static(rest)
        # boundary[0] carries the gradient back to the tokenizer; the others (position embedding) are inputs only
        self._static_tokens = boundary[0].detach().clone().requires_grad_(True)
        self._static_extra = tuple(t.detach().clone() for t in boundary[1:])
        self._static_dtokens = torch.zeros_like(self._static_tokens)

        def stage_b(first=True):
            with fused_ops.activate(self._fused_ctx), self._autocast():
                data = self.policy.hybrid_merge(clone_batch(self._static_batch), (self._static_tokens,) + self._static_extra)
                out = self._call_policy(data)
            loss = out["loss"]
            self._static_tokens.grad = None
            (loss / self.accumulate).backward()
            self._static_dtokens.copy_(self._static_tokens.grad)
            opt.collect(first=first, subset=self._subset_b)
            opt_stats = torch.stack([loss.detach().float(), out.get("action_loss", loss).detach().float(),
                                     torch.as_tensor(out.get("kl_loss", 0.0), device=loss.device).detach().float()])
            return opt_stats

        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(3):
                stage_b()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for n, b in self.policy.named_buffers():
                b.copy_(buffers[n])
        def reset():
            for k in range(len(opt.params)):
                opt._stash[k] = None
                opt.params[k].grad = None

        from .._graphs import captured

        self._graph, self._static_stats = captured(lambda: stage_b(first=True))
        reset()
        self._graph_acc = None
        if self.accumulate > 1:  # later micro-batches of an accumulation window ADD their gradients
            self._graph_acc, self._static_stats_acc = captured(lambda: stage_b(first=False))
            reset()

    def _hybrid_step(self, batch):
        from ..policy import fused_ops

        if self._graph is None:
            self._hybrid_setup(batch)
        ragged, rest = self.policy.hybrid_split(batch)
        if self._signature(rest) != self._static_sig:
            raise ValueError("hybrid mode needs a fixed batch size / action layout (only the point clouds may be ragged)")
        with fused_ops.activate(self._fused_ctx), self._autocast():
            boundary = tuple(self._call_policy(ragged, only=self._subset_a_set, stage="tokenize"))  # eager: shapes follow the clouds
        tokens = boundary[0]
        with torch.no_grad():
            self._static_tokens.copy_(tokens)
            for dst, src in zip(self._static_extra, boundary[1:]):
                dst.copy_(src)
            self._copy_into(self._static_batch, rest)
        first = self.micro % self.accumulate == 0
        (self._graph if first else self._graph_acc).replay()
        tokens.backward(self._static_dtokens)
        self.optimizer.collect(first=first, subset=self._subset_a)
        return (self._static_stats if first else self._static_stats_acc).clone()

    def _forward_backward(self, batch, first=None):
        """`first`: is this the first micro-batch of an accumulation window (gradients overwrite the flat buffer) or a
        later one (they add)?  Must be passed explicitly when capturing: the choice is baked into the hipGraph."""
        from ..policy import fused_ops

        if first is None:
            first = self.micro % self.accumulate == 0

        shadows = getattr(self, "_shadow_names", None)
        with fused_ops.activate(self._fused_ctx), self._autocast():
            if shadows:
                opt = self.optimizer
                repl = {n: _ShadowParam.apply(p, opt.shadow[k], opt, k) for n, p, k in shadows}
                out = torch.func.functional_call(self.policy, repl, (batch,))
            else:
                out = self.module(batch)
        loss = out["loss"]
        (loss / self.accumulate).backward()
        if getattr(self.optimizer, "collect_mode", False):
            self.optimizer.collect(first=first)
        aux1 = out.get("action_loss", loss)
        aux2 = out.get("kl_loss", 0.0)
        return torch.stack([loss.detach().float(), aux1.detach().float(),
                            torch.as_tensor(aux2, device=loss.device).detach().float()])

    # ---- hipGraph capture of one micro-batch (forward + backward) -------------------------------
    @staticmethod
    def _signature(batch, prefix=()):
        sig = []
        for k in sorted(batch):
            v = batch[k]
            if isinstance(v, dict):
                sig.extend(BCTrainer._signature(v, prefix + (k,)))
            elif torch.is_tensor(v):
                sig.append((prefix + (k,), tuple(v.shape), v.dtype, tuple(getattr(v, "_pcm_host", ()) or ())))
        return tuple(sig)

    @staticmethod
    def _clone_static(batch):
        out = {}
        for k, v in batch.items():
            if isinstance(v, dict):
                out[k] = BCTrainer._clone_static(v)
            elif torch.is_tensor(v):
                c = v.clone()
                if hasattr(v, "_pcm_host"):
                    c._pcm_host = list(v._pcm_host)
                out[k] = c
            else:
                out[k] = v
        return out

    @staticmethod
    def _copy_into(static, batch):
        for k, v in batch.items():
            if isinstance(v, dict):
                BCTrainer._copy_into(static[k], v)
            elif torch.is_tensor(v) and static[k] is not v:
                static[k].copy_(v, non_blocking=True)

    def _capture(self, batch):
        from .synthetic import clone_batch

        self._static_sig = self._signature(batch)
        self._static_batch = self._clone_static(batch)
        buffers = {n: b.clone() for n, b in self.policy.named_buffers()}
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # warm-up off the capture stream: lazy inits, GEMM heuristics
            for _ in range(3):
                self._forward_backward(clone_batch(self._static_batch))
        cur.wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():  # the warm-up must not count as training: restore BN statistics
            for n, b in self.policy.named_buffers():
                b.copy_(buffers[n])
        self.optimizer.zero_grad()
        from .._graphs import captured

        # (thread_local error mode: RCCL's watchdog / other host threads may touch the HIP API while we capture)
        self._graph, self._static_stats = captured(lambda: self._forward_backward(clone_batch(self._static_batch), first=True))
        self._graph_acc = None
        if self.accumulate > 1 and getattr(self.optimizer, "collect_mode", False):
            # bf16 hand-off: "overwrite" vs "add into" the flat gradient buffer is decided in Python, i.e. at capture
            # time -- the later micro-batches of an accumulation window need their own graph (fp32 mode accumulates
            # through the .grad views and zeroes the buffer outside the graph, one graph serves both)
            self._graph_acc, self._static_stats_acc = captured(
                lambda: self._forward_backward(clone_batch(self._static_batch), first=False))
        self.optimizer.zero_grad()

    def prefetch_sampling(self, next_batch):
        """Hand the NEXT micro-batch over early (what a data loader's prefetch does): its FPS + kNN indices -- functions of
        the input coordinates only -- are computed on the side stream while the current step runs.  Pays off in hybrid
        mode, where the sampling otherwise sits in front of the captured graph; a no-op in graph mode (the sampling is
        part of the captured graph there) and on the host."""
        if self.mode == "graph" or next_batch is None:
            return
        pcds = next_batch["pcds"] if "pcds" in next_batch else next_batch.get("obs", {}).get("pcds")
        if pcds is None or not pcds["coord"].is_cuda:
            return
        pol = self.policy
        target = pol if hasattr(pol, "prefetch_sampling") else getattr(pol, "obs_encoder", None)
        if target is not None and getattr(target, "overlap_sampling", False) and hasattr(target, "prefetch_sampling"):
            target.prefetch_sampling(pcds)

    def training_step(self, batch, prefetch=None):
        """One micro-batch: forward, loss, backward and -- on accumulation boundaries -- the
        optimizer step.  Returns the (detached, on-device) loss dict of this micro-batch.
        `prefetch`: the next micro-batch, if it is already on the device (see prefetch_sampling)."""
        self.module.train()
        if prefetch is not None:
            self.prefetch_sampling(prefetch)
        if self._fused_ctx is not None:
            self._fused_ctx.set_step(self.micro)
        first = self.micro % self.accumulate == 0
        stepping = (self.micro + 1) % self.accumulate == 0
        if self.mode == "eager":
            sync_ctx = contextlib.nullcontext()
            if self.distributed and not stepping:
                sync_ctx = self.module.no_sync()
            with sync_ctx:
                stats = self._forward_backward(batch)
            if stepping:
                if self.clip is not None and self.clip > 0:
                    torch.nn.utils.clip_grad_norm_([p for g in self.optimizer.param_groups for p in g["params"]], self.clip)
                self.optimizer.step()
                self.scheduler.step()
                self.optimizer.zero_grad(set_to_none=True)
                self.optimizer_steps += 1
        else:
            if self.mode == "graph" and self._graph is None:
                try:
                    self._capture(batch)
                except Exception as e:  # capture is an optimisation: fall back to the same maths without replay
                    import warnings

                    warnings.warn(f"hipGraph capture failed ({type(e).__name__}: {e}); continuing in mode='flat'")
                    torch.cuda.synchronize()
                    self.mode, self._graph = "flat", None
                    opt = self.optimizer
                    for k, p in enumerate(opt.params):
                        # fp32 (non-collect) mode: autograd must keep accumulating INTO the flat buffer's views
                        p.grad = None if getattr(opt, "collect_mode", False) else opt.g_views[k]
                        if hasattr(opt, "_stash"):
                            opt._stash[k] = None
                    opt.flat_g.zero_()
            if self.mode == "hybrid":
                stats = self._hybrid_step(batch)
            elif self.mode == "graph":
                if self._signature(batch) != self._static_sig:
                    raise ValueError("graph mode needs the captured batch layout (equal shapes and cloud offsets); "
                                     "use mode='flat' for ragged batches")
                if first:
                    self.optimizer.zero_grad()
                self._copy_into(self._static_batch, batch)
                if first or self._graph_acc is None:
                    self._graph.replay()
                    stats = self._static_stats.clone()
                else:
                    self._graph_acc.replay()
                    stats = self._static_stats_acc.clone()
            else:
                if first:
                    self.optimizer.zero_grad()
                stats = self._forward_backward(batch)
            if stepping:
                if self.distributed:
                    dist.all_reduce(self.optimizer.flat_g)  # SUM; the 1/world is applied in the Adam kernel
                self.optimizer.step()
                self.optimizer_steps += 1
        self.micro += 1
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
        sd = {"policy": self.policy.state_dict(), "optimizer": self.optimizer.state_dict(), "micro": self.micro,
              "optimizer_steps": self.optimizer_steps, "mode": self.mode}
        if self.mode == "eager":
            sd["scheduler"] = self.scheduler.state_dict()
        return sd

    def load_state_dict(self, sd):
        self.policy.load_state_dict(sd["policy"])
        self.optimizer.load_state_dict(sd["optimizer"])
        if self.mode == "eager" and "scheduler" in sd:
            self.scheduler.load_state_dict(sd["scheduler"])
        self.micro = sd["micro"]
        self.optimizer_steps = sd["optimizer_steps"]
        mirror = getattr(self.optimizer, "flat_p_bf16", None)
        if mirror is not None:  # the bf16 weight copies the forward reads are normally refreshed by the Adam kernel
            with torch.no_grad():
                mirror.copy_(self.optimizer.flat_p)
