"""The behaviour-cloning training step, Lightning-free.

Reproduces what Lightning's Trainer does around
``ManiSkill2ACTBCModule.training_step`` (/root/reference/src/models/maniskill2_act_bc_module.py:64-86)
and ``configure_optimizers`` (:347-367) under ``configs/trainer/ddp.yaml``:

    loss = policy(batch)["loss"] / accumulate_grad_batches ; backward            (every micro-batch)
    every `accumulate` micro-batches: clip_grad_norm_(0.5) ; AdamW.step ; OneCycleLR.step ; zero_grad

Data parallel = one process per GPU, ``torch.distributed`` over RCCL (backend "nccl" on ROCm).  The reference gets
"gradient all-reduce overlapped with backward" from DistributedDataParallel's bucket hooks (mode="eager" keeps exactly
that).  The flat / graph / hybrid modes own their gradients in ONE flat buffer laid out in backward order, run backward
in STAGES (policy.backward_stages(), policy/staging.py) and start the all-reduce of a stage's contiguous gradient slab
on RCCL's stream while the next stage still computes -- with hipGraph replay each stage is its own graph, so the
collectives are launched between replays and never captured.
Metrics stay on the device and are only read every ``log_every_n_steps`` (ddp.yaml:15) -- the
per-step ``.item()`` would serialise host and GPU.
"""
import contextlib
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from .. import _lib
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

    from ..policy.precision import fp32_tokenizer_parameter_ids

    ids, skip = set(), fp32_tokenizer_parameter_ids(policy)  # the tokenizer's weights are consumed in fp32 (policy/precision.py)
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


class _Stage:
    """One backward stage: `lower` = name of the staging.cut that bounds it from below (None: runs to the inputs),
    `indices` = its parameters in the flat optimizer, `slab` = [start, end) floats of the flat gradient exchanged when
    the stage is done."""

    def __init__(self, lower, indices, slab):
        self.lower, self.indices, self.slab = lower, indices, slab
        self.index_set = set(indices)


def plan_slabs(stage_params, lead_end, numel):
    """Gradient slabs [start, end) of the flat buffer, one per backward stage, exchanged when that stage's backward is done.

    stage_params[si] = [(offset, numel), ...] of the stage's parameters.  The flat buffer holds the LEADING parameter group
    (the decayed weights, laid out in backward order) in [0, lead_end) and, with `filter_bias_and_bn`, a small undecayed
    group (biases, norm weights) of ALL stages behind it.  Slabs are cut from the leading group only: stage si ends where
    the first leading-group parameter of any later stage begins; the undecayed tail rides with the last slab.  A stage with
    no leading-group parameter (only biases / norm weights) gets an empty slab -- its gradients sit in the tail, which is
    exchanged after the last stage, i.e. never before they are complete.  Checked: slabs are monotone, disjoint, cover
    [0, numel), and every parameter of stage si lies in a slab with index >= si."""
    n_stage = len(stage_params)
    lead_starts = [min((o for o, _ in ps if o < lead_end), default=None) for ps in stage_params]
    slabs, cursor = [], 0
    for si in range(n_stage):
        if si == n_stage - 1:
            end = numel
        else:
            later = [s for s in lead_starts[si + 1:] if s is not None]
            end = min(later) if later else lead_end
            assert end >= cursor, "backward stages must be laid out in backward order inside the leading parameter group"
        slabs.append((cursor, end))
        cursor = end
    assert slabs[0][0] == 0 and slabs[-1][1] == numel and all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(slabs, slabs[1:]))
    for si, ps in enumerate(stage_params):
        for o, cnt in ps:
            owner = next(k for k, (a, b) in enumerate(slabs) if a <= o < b or (k == n_stage - 1 and o >= a))
            assert owner >= si and o + cnt <= slabs[owner][1], (
                "gradient of a stage-%d parameter at offset %d would be exchanged with stage %d, before it is complete" % (si, o, owner))
    return slabs


def undecayed_parameters(module):
    """The trainable parameters the reference's `build_optimizer_v2` leaves WITHOUT weight decay (`param_groups_weight_decay`,
    src/utils/optimizer.py:152-170, reached by the Diffusion-Policy modules, maniskill2_dp_bc_module.py:326-327): at most one dimension, or
    a name that ends in `.bias`.  Pinned by tests/golden/optim_ref.npz, which the reference's own function produced."""
    return [p for n, p in module.named_parameters() if p.requires_grad and (p.ndim <= 1 or n.endswith(".bias"))]


class BCTrainer:
    """mode:
      "eager"  torch.optim.AdamW + OneCycleLR, DistributedDataParallel (+ SyncBatchNorm) when
               distributed -- the literal Lightning recipe; also the CPU path.
      "flat"   FlatAdamW (csrc/optim.hip) + host OneCycle; data parallel = the flat gradient buffer all-reduced in
               backward-ordered slabs that overlap the rest of backward (no DDP wrapper, no bucket hooks).
      "graph"  "flat" with forward+backward of a micro-batch captured ONCE into hipGraphs (one per backward stage) and
               replayed (the step is launch-bound in eager mode: ~2400 launches); needs static
               shapes (equal-size clouds); BatchNorm statistics are per rank.
      "hybrid" eager tokenizer (ragged clouds, synchronised BatchNorm) + hipGraphs for everything behind the token matrix.
    `staged`: None = backward stages when data parallel (the exchange needs them); True forces them on one GPU (tests).
    """

    def __init__(self, policy, total_steps, optim=None, precision="fp32", device=None, distributed=False,
                 sync_batchnorm=True, bucket_cap_mb=32, log_every_n_steps=50, mode="eager", flat_optimizer_cls=None, staged=None,
                 side_weight_grads=False, external_sampling=True, defer_reductions=None, allow_eval_submodules=False):
        o = dict(ACT_OPTIM)
        self.side_weight_grads = bool(side_weight_grads)
        # a submodule the caller froze on purpose (policy.backbone.eval(): fixed BatchNorm statistics) stays in eval mode; see
        # training_step for what happens to one that is in eval mode by accident
        self.allow_eval_submodules = bool(allow_eval_submodules)
        # closing reductions batched per backward stage (policy/deferred.py); PCM_DEFER_REDUCTIONS=0 is the A/B switch
        self.defer_reductions = (os.environ.get("PCM_DEFER_REDUCTIONS", "1") != "0") if defer_reductions is None else bool(defer_reductions)
        self.external_sampling = bool(external_sampling)  # graph mode: FPS / kNN outside the captured graph (prefetchable)
        self._static_sampling = False
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
        if not _lib.on_hip(self.device) and mode != "eager" and (flat_optimizer_cls is None or mode == "graph"):
            raise ValueError("flat/graph modes run on the HIP device only")
        self.mode = mode
        freeze_unused_parameters(policy)
        self.policy = policy
        self.module = policy
        from ..policy.sync_bn import multi_rank

        self.distributed = bool(distributed and multi_rank())
        self.world = dist.get_world_size() if self.distributed else 1
        # total_steps is used as given (the reference hands Lightning's estimated_stepping_batches to torch's OneCycleLR,
        # src/utils/scheduler.py:101-143).  That scheduler divides by `pct_start * total_steps - 1`: the one value it cannot take is
        # zero (a negative first-phase end just means "no warm-up phase"); torch dies there with ZeroDivisionError, here with a message.
        total_steps = int(total_steps)
        if total_steps <= 0 or float(o["pct_start"] * total_steps) - 1 == 0:
            raise ValueError(f"OneCycleLR cannot form a cycle of total_steps={total_steps} with pct_start={o['pct_start']}")
        # ---- synchronised BatchNorm (configs/trainer/ddp.yaml:9).  eager: torch's SyncBatchNorm.  flat / hybrid: the
        # BatchNorm layers the fused kernels own exchange their statistics themselves (policy/sync_bn.py), every other
        # BatchNorm module becomes a torch SyncBatchNorm -- all of them run OUTSIDE the captured graphs in hybrid mode.
        # graph mode (round 4): the step is captured as a CHAIN of graphs cut at every collective (_graphs.SegmentedCapture), so
        # the tokenizer's synchronised statistics and the gradient slabs stay plain eager RCCL calls between graph replays.
        # That needs every BatchNorm of the policy to be owned by a fused kernel (a torch SyncBatchNorm module would issue its
        # collectives inside the capture): true for the ACT policies with the fused SA layer; otherwise graph mode keeps
        # per-rank statistics, as before, and bench.py picks hybrid.
        self.sync_batchnorm = bool(self.distributed and sync_batchnorm)
        if self.sync_batchnorm and mode == "graph" and not self.all_batchnorms_fused(policy):
            import warnings

            warnings.warn("BCTrainer(mode='graph', distributed=True, sync_batchnorm=True): this policy has BatchNorm modules no fused "
                          "kernel owns, so a captured step cannot exchange their statistics -- training continues with PER-RANK BatchNorm "
                          "statistics (the reference's configs/trainer/ddp.yaml:9 asks for synchronised ones).  Use mode='hybrid' to "
                          "keep them synchronised.", RuntimeWarning, stacklevel=2)
            self.sync_batchnorm = False
        self.segmented = bool(mode == "graph" and (self.distributed or os.environ.get("PCM_FORCE_SEGMENTS") == "1"))
        if self.sync_batchnorm:
            if mode == "eager":
                policy = nn.SyncBatchNorm.convert_sync_batchnorm(policy)
                self.policy = self.module = policy
            else:
                from ..policy.sync_bn import enable_sync_batchnorm

                enable_sync_batchnorm(policy)
        # fused transformer tail ops (csrc/drln.hip, ffn.hip) need a device-resident dropout seed: flat / graph modes
        self._fused_ctx = None
        if mode != "eager" and _lib.on_hip(self.device):
            from ..policy.fused_ops import FusedContext

            self._fused_ctx = FusedContext(self.device)
            self._fused_ctx.defer_pos_grads = True  # this loop flushes the sinks after every backward call (_segments_inner)
        trainable = [p for p in self.policy.parameters() if p.requires_grad]
        # ---- backward stages: parameters ordered by when their gradient is complete (latest-used first)
        want_stages = (self.distributed and mode != "eager") if staged is None else bool(staged)
        stage_defs = [(None, trainable)]
        if want_stages and mode != "eager" and hasattr(self.policy, "backward_stages"):
            seen, stage_defs = set(), []
            for lower, ps in self.policy.backward_stages():
                ps = [p for p in ps if p.requires_grad and id(p) not in seen]
                seen.update(id(p) for p in ps)
                stage_defs.append((lower, ps))
            assert all(id(p) in seen for p in trainable), "backward_stages() must cover every trainable parameter"
            if mode == "hybrid":
                tok = {id(p) for p in self.policy.tokenizer_parameters() if p.requires_grad}
                assert {id(p) for p in stage_defs[-1][1]} == tok, "hybrid mode: the last backward stage must be exactly the tokenizer"
        ordered = [p for _, ps in stage_defs for p in ps]
        betas = tuple(o.get("betas", (0.9, 0.999)))
        params = ordered
        if o.get("filter_bias_and_bn", False) and o["weight_decay"]:
            # build_optimizer_v2 -> param_groups_weight_decay (src/utils/optimizer.py:152-170, 296-300)
            nd_ids = {id(p) for p in undecayed_parameters(self.policy)}
            decay = [p for p in ordered if id(p) not in nd_ids]
            no_decay = [p for p in ordered if id(p) in nd_ids]
            # decayed group first and in backward order: its stage slabs are what the exchange overlaps; the small
            # undecayed group (biases, norm weights) rides with the last slab
            params = [{"params": decay, "weight_decay": o["weight_decay"]}, {"params": no_decay, "weight_decay": 0.0}]
        if mode == "eager":
            if self.distributed:
                ids = [self.device.index] if _lib.on_hip(self.device) else None
                self.module = nn.parallel.DistributedDataParallel(
                    self.policy, device_ids=ids, gradient_as_bucket_view=True, bucket_cap_mb=bucket_cap_mb,
                )
            # build_optimizer(cfg, policy, None): one group, every parameter decayed (src/utils/optimizer.py:33-37)
            self.optimizer = torch.optim.AdamW(params, lr=o["lr"], weight_decay=o["weight_decay"], betas=betas,
                                               fused=_lib.on_hip(self.device))
            self.scheduler = torch.optim.lr_scheduler.OneCycleLR(
                self.optimizer, max_lr=o["lr"], total_steps=total_steps, pct_start=o["pct_start"],
                anneal_strategy="cos", div_factor=o["div_factor"], final_div_factor=o["final_div_factor"],
            )
            self._stages = []
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
                # the mirror set depends on `tokenizer_fp32` (policy/precision.py) AS IT IS NOW: the flag must be set before the trainer
                # is built; training_step checks that it has not been flipped since (round-5 ADVICE)
                from ..policy.precision import tokenizer_owners

                self._tokenizer_flags = [(m, bool(getattr(m, "tokenizer_fp32", True))) for m in tokenizer_owners(self.policy)]
                index = {id(p): k for k, p in enumerate(self.optimizer.params)}
                self._shadow_names = [(n, p, index[id(p)]) for n, p in self.policy.named_parameters()
                                      if id(p) in index and self.optimizer.shadow[index[id(p)]] is not None]
            self._stages = self._plan_stages(stage_defs)
            if self.distributed:  # DDP would broadcast rank 0's weights and buffers at construction: same here
                with torch.no_grad():
                    flat_p = getattr(self.optimizer, "flat_p", None)
                    if flat_p is not None:
                        dist.broadcast(flat_p, src=0)
                        if getattr(self.optimizer, "flat_p_bf16", None) is not None:
                            self.optimizer.flat_p_bf16.copy_(flat_p)
                    else:
                        for p in self.optimizer.params:
                            dist.broadcast(p.data, src=0)
                    for b in self.policy.buffers():
                        dist.broadcast(b, src=0)
        self.exchange_description = (
            "flat gradient all-reduced over RCCL in %d backward-ordered slab(s) (%s MB), each launched when its stage's backward is "
            "done and overlapping the following stages" % (len(self._stages), " + ".join("%.0f" % ((s.slab[1] - s.slab[0]) * 4 / 1e6)
                                                                                      for s in self._stages))
            if mode != "eager" else "DistributedDataParallel bucket hooks (%d MB buckets)" % bucket_cap_mb)
        self.micro = 0
        self.optimizer_steps = 0
        self._sums = None
        self._count = 0
        self._graph = None       # graph mode: one graph per backward stage; hybrid mode: graphs of the captured stages
        self._graph_acc = None
        self._static_batch = None
        self._static_stats = None
        self._static_sig = None
        self._works = []
        self._stepping = True

    @staticmethod
    def all_batchnorms_fused(policy):
        """Every BatchNorm module of `policy` is in some owner's `fused_batchnorms()` (its statistics exchange is ours to place)."""
        fused = set()
        for owner in policy.modules():
            get = getattr(owner, "fused_batchnorms", None)
            if get is not None:
                fused.update(id(m) for m in get())
        return all(id(m) in fused for m in policy.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm))

    # ------------------------------------------------------------------------------------------ stages / exchange
    def _plan_stages(self, stage_defs):
        opt = self.optimizer
        index = {id(p): k for k, p in enumerate(opt.params)}
        offsets = getattr(opt, "offsets", None)
        if offsets is None:  # host stand-in: parameters packed back to back
            offsets, o = [], 0
            for p in opt.params:
                offsets.append(o)
                o += p.numel()
        numel = getattr(opt, "numel", None) or int(opt.flat_g.numel())
        idxs = [sorted(index[id(p)] for p in ps) for _, ps in stage_defs]
        segments = getattr(opt, "segments", None)
        lead_end = segments[0][0] + segments[0][1] if segments else numel  # end of the leading (decayed) parameter group
        slabs = plan_slabs([[(offsets[k], opt.params[k].numel()) for k in ix] for ix in idxs], lead_end, numel)
        return [_Stage(lower, idxs[si], slabs[si]) for si, (lower, _) in enumerate(stage_defs)]

    def _exchange(self, si):
        """Stage `si` of the stepping micro-batch is done: start the all-reduce of its gradient slab (asynchronous: it runs
        on the process group's own stream behind everything enqueued so far and beside whatever is enqueued next)."""
        if not (self.distributed and self._stepping):
            return
        a, b = self._stages[si].slab
        if b > a:
            self._works.append(dist.all_reduce(self.optimizer.flat_g[a:b], async_op=True))  # SUM; 1/world is applied by Adam

    def _finish_exchange(self):
        """Join the slab all-reduces before the optimizer reads the gradients.  Two events bracket the join on the compute
        stream: their distance is the part of the gradient exchange that backward did NOT hide (exchange_stats())."""
        timed = bool(self._works) and _lib.on_hip(self.device)
        if timed:
            ring = self.__dict__.setdefault("_exchange_events", [])
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        for w in self._works:
            w.wait()
        if timed:
            b.record()
            ring.append((a, b))
            del ring[:-32]
        self._works = []

    def exchange_stats(self):
        """Data-parallel runs: exposed gradient-exchange time per optimizer step (mean / max over the last <= 32 steps, ms) and
        the slab sizes.  Synchronises with the device: call it outside timed regions."""
        ring = self.__dict__.get("_exchange_events") or []
        if not ring:
            return None
        torch.cuda.synchronize(self.device)
        ms = [a.elapsed_time(b) for a, b in ring]
        return {"exposed_ms_mean": round(sum(ms) / len(ms), 4), "exposed_ms_max": round(max(ms), 4), "steps": len(ms),
                "slabs_mb": [round((s.slab[1] - s.slab[0]) * 4 / 1e6, 2) for s in self._stages]}

    # ------------------------------------------------------------------------------------------
    def _autocast(self):
        if self.precision == "bf16":
            return torch.autocast(device_type=self.device.type, dtype=torch.bfloat16)
        return contextlib.nullcontext()

    def _shadow_items(self, only=None):
        """(module, attribute name, master parameter, flat index) of every bf16-mirrored parameter (restricted to the flat
        indices in `only`), resolved once: the per-step substitution is then two dict operations per parameter instead
        of torch.func.functional_call's walk over the module tree (1 ms per call for this policy)."""
        shadows = getattr(self, "_shadow_names", None)
        if not shadows:
            return ()
        cache = self.__dict__.setdefault("_shadow_item_cache", {})
        key = None if only is None else id(only)
        if key not in cache:
            items = []
            for n, p, k in shadows:
                if only is not None and k not in only:
                    continue
                path, _, attr = n.rpartition(".")
                items.append((self.policy.get_submodule(path) if path else self.policy, attr, p, k))
            cache[key] = (tuple(items), only)  # `only` kept alive: its id is the cache key
        return cache[key][0]

    def _call_policy(self, batch, only=None, **kwargs):
        items = self._shadow_items(only)
        if not items:
            return self.module(batch, **kwargs)
        opt = self.optimizer
        try:
            # an instance attribute shadows the registered nn.Parameter for every `module.weight` read in forward();
            # named_parameters() / state_dict() keep seeing the fp32 masters
            for mod, attr, p, k in items:
                mod.__dict__[attr] = _ShadowParam.apply(p, opt.shadow[k], opt, k)
            return self.policy(batch, **kwargs)
        finally:
            for mod, attr, _, _ in items:
                mod.__dict__.pop(attr, None)

    def _loss_seed(self, loss):
        """d(loss / accumulate) / d loss as a cached device scalar: seeding backward with it replaces the division kernel, the
        ones_like fill of .backward() and the division's backward multiply (three one-element launches per micro-batch)."""
        key = (loss.device, loss.dtype, self.accumulate)
        seeds = self.__dict__.setdefault("_loss_seeds", {})
        if key not in seeds:
            if loss.is_cuda and torch.cuda.is_current_stream_capturing():
                return torch.full_like(loss, 1.0 / self.accumulate)  # first use inside a capture: not cached (graph-pool memory)
            seeds[key] = torch.full((), 1.0 / self.accumulate, device=loss.device, dtype=loss.dtype)
        return seeds[key]

    @staticmethod
    def _stats_of(out):
        loss = out["loss"]
        return torch.stack([loss.detach().float(), out.get("action_loss", loss).detach().float(),
                            torch.as_tensor(out.get("kl_loss", 0.0), device=loss.device).detach().float()])

    def _segments(self, make_out, first, stages, leaf=None):
        """Generator over the backward stages of one micro-batch.  The first next() runs the forward pass (make_out) and
        stage 0's backward; every further next() runs one more stage.  Yields (stage index, stats).  `leaf`: a leaf tensor
        below the last stage whose gradient is wanted as well (hybrid: the static token matrix)."""
        from ..policy import staging

        opt = self.optimizer
        # hybrid mode always hands gradients over explicitly (its two halves write disjoint parts of the flat buffer at
        # different times); the other modes do so only for the bf16 mirror, fp32 accumulates through the .grad views
        collect = getattr(opt, "collect_mode", False) or self.mode == "hybrid"
        from ..policy import deferred, rows_linear

        # weight-gradient products on a second stream (rows_linear._SideQueue): only where gradients are handed over
        # explicitly (nothing reads a dW before `collect`) and the step is replayed as a hipGraph (parallel branches).
        # OFF by default: measured on MI355X / ROCm 7.2 at C2 the replayed step gets SLOWER with the extra branches
        # (7.55 -> 7.93 ms with every dW on the side stream, 8.1 ms with only the 8192-row ones): DESIGN.md section 9
        rows_linear.SIDE.active = bool(collect and self.side_weight_grads and self.mode in ("graph", "hybrid"))
        # closing reductions of the fused backward kernels batched per backward stage (policy/deferred.py): only where the
        # gradients are handed over explicitly, i.e. nothing reads one before `collect`
        window = bool(collect and self.defer_reductions and not rows_linear.SIDE.active) and deferred.begin()
        failed = True  # stays True when the body raises or the generator is closed before its last stage (GeneratorExit)
        try:
            yield from self._segments_inner(make_out, first, stages, leaf, collect)
            failed = False
        finally:
            rows_linear.join_side()
            rows_linear.SIDE.active = False
            if window:
                deferred.end(failed=failed)

    def _segments_inner(self, make_out, first, stages, leaf, collect):
        from ..policy import deferred, rows_linear, staging

        opt = self.optimizer
        if len(stages) == 1:
            out = make_out()
            with deferred.backward_phase():
                out["loss"].backward(self._loss_seed(out["loss"]))
            if self._fused_ctx is not None:
                self._fused_ctx.flush_sinks()
            deferred.flush()
            rows_linear.join_side()
            if collect:
                opt.collect(first=first, subset=None if len(stages[0].indices) == len(opt.params) else stages[0].indices)
            yield 0, self._stats_of(out)
            return
        with staging.record() as rec:
            out = make_out()
        stats = self._stats_of(out)
        roots, grads = [out["loss"]], [self._loss_seed(out["loss"])]
        for si, st in enumerate(stages):
            inputs = rec.requested(st.lower) + [opt.params[k] for k in st.indices]
            last = si == len(stages) - 1
            if last and st.lower is None and leaf is not None:
                inputs = inputs + [leaf]
            if roots:
                with deferred.backward_phase():
                    torch.autograd.backward(roots, grads, inputs=inputs)
            if self._fused_ctx is not None:
                self._fused_ctx.flush_sinks()  # deferred position-embedding gradients of this stage (fused_ops.GradSink)
            deferred.flush()  # this stage's closing reductions, one launch per 24
            rows_linear.join_side()
            if collect:
                opt.collect(first=first, subset=st.indices)
            if st.lower is not None:
                roots = rec.roots(st.lower)
                grads = [t.grad for t in roots]
                for t in roots:
                    t.grad = None
                if last and leaf is not None and roots:  # the short way from the last cut down to the leaf, same segment
                    with deferred.backward_phase():
                        torch.autograd.backward(roots, grads, inputs=[leaf])
            yield si, stats

    def _forward_backward(self, batch, first=None):
        """One micro-batch, eagerly: forward, staged backward, gradient hand-off; exchanges every finished stage."""
        from ..policy import fused_ops

        if first is None:
            first = self.micro % self.accumulate == 0

        def make_out():
            with fused_ops.activate(self._fused_ctx), self._autocast():
                return self._call_policy(batch)

        if self.mode == "eager":
            out = make_out()
            (out["loss"] / self.accumulate).backward()
            return self._stats_of(out)
        stats = None
        for si, stats in self._segments(make_out, first, self._stages):
            self._exchange(si)
        return stats

    # ---- hipGraph capture helpers ---------------------------------------------------------------
    def _capture_chain(self, make_gen, nseg):
        """Data-parallel graph mode: the `nseg` backward stages of a generator captured as ONE chain of graphs that is cut at every
        collective -- the gradient slab exchange behind each stage (here) and the synchronised BatchNorm statistics inside the
        tokenizer's forward / backward nodes (policy/sync_bn.py -> _graphs.between).  Returns ([chain], stats)."""
        from .._graphs import SegmentedCapture

        chain = SegmentedCapture()
        stats = None
        with chain:
            gen = make_gen()
            for si in range(nseg):
                _, stats = next(gen)
                chain.between(lambda si=si: self._exchange(si))
            for _ in gen:
                pass
        self._finish_exchange()  # the capture pass issued real (meaningless) slab all-reduces: join them before anything else
        return [chain], stats

    def _capture_segments(self, make_gen, nseg):
        """Capture the `nseg` segments of a generator into one hipGraph each (shared memory pool: the later graphs use
        what the earlier ones saved for backward).  Returns (graphs, stats tensor)."""
        from .._graphs import finalize, new_graph

        if getattr(self, "segmented", False) and self.mode == "graph":
            return self._capture_chain(make_gen, nseg)

        gen = make_gen()
        graphs, stats, pool = [], None, None
        for _ in range(nseg):
            g = new_graph()
            kw = {} if pool is None else {"pool": pool}
            with torch.cuda.graph(g, capture_error_mode="thread_local", **kw):
                _, stats = next(gen)
            finalize(g)  # memset nodes -> kernel nodes (see _graphs.py), then instantiate
            pool = g.pool()
            graphs.append(g)
        for _ in gen:  # nothing left to run: lets the generator finish
            pass
        return graphs, stats

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
    def _copy_into(static, batch, _pairs=None):
        """The batch into the captured step's input tensors: one table-driven copy launch instead of a copy per tensor."""
        pairs = [] if _pairs is None else _pairs
        for k, v in batch.items():
            if isinstance(v, dict):
                BCTrainer._copy_into(static[k], v, pairs)
            elif torch.is_tensor(v) and static[k] is not v:
                pairs.append((static[k], v))
        if _pairs is None and pairs:
            from .. import _lib

            _lib.copy_batch(pairs)  # one launch for the whole batch (15 tensors in an ACT batch)

    def _reset_grads(self):
        opt = self.optimizer
        for k in range(len(opt.params)):
            if hasattr(opt, "_stash"):
                opt._stash[k] = None
            if getattr(opt, "collect_mode", False) or self.mode == "hybrid":
                opt.params[k].grad = None

    def _warm_up(self, gen):
        """Three un-captured runs off the capture stream (lazy inits, GEMM heuristics); BatchNorm buffers are restored:
        the warm-up must not count as training."""
        buffers = {n: b.clone() for n, b in self.policy.named_buffers()}
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(3):
                for _ in gen(True):
                    pass
        cur.wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for n, b in self.policy.named_buffers():
                b.copy_(buffers[n])

    # ---- graph mode: the whole micro-batch, one hipGraph per backward stage -------------------------
    def _capture(self, batch):
        from ..policy import fused_ops
        from .synthetic import clone_batch

        self._static_sig = self._signature(batch)
        self._static_batch = self._clone_static(batch)
        stages = self._stages
        # sampling outside the graph: FPS / kNN / index statistics depend on the coordinates only, so they are computed
        # eagerly on the side stream -- one batch ahead when the caller hands the next batch over (training_step(...,
        # prefetch=)) -- and reach the replayed graphs through static index buffers
        self._static_sampling = False
        target, pcds = self._sampling_target(), self._pcds_of(self._static_batch)
        if self.external_sampling and target is not None and hasattr(target, "install_static_sampling") and pcds is not None \
                and getattr(target, "sa_impl", None) == "fused":
            target.install_static_sampling(pcds, target.sampling_for(pcds, overlap=False))
            self._static_sampling = True

        def gen(first):
            def make_out():
                with fused_ops.activate(self._fused_ctx), self._autocast():
                    return self._call_policy(clone_batch(self._static_batch))

            return self._segments(make_out, first, stages)

        self._warm_up(gen)
        self.optimizer.zero_grad()
        self._reset_grads()
        self._graph, self._static_stats = self._capture_segments(lambda: gen(True), len(stages))
        self._graph_acc = None
        if self.accumulate > 1 and getattr(self.optimizer, "collect_mode", False):
            # bf16 hand-off: "overwrite" vs "add into" the flat gradient buffer is decided in Python, i.e. at capture
            # time -- the later micro-batches of an accumulation window need their own graphs (fp32 mode accumulates
            # through the .grad views and zeroes the buffer outside the graph, one set serves both)
            self._reset_grads()
            self._graph_acc, self._static_stats_acc = self._capture_segments(lambda: gen(False), len(stages))
        self.optimizer.zero_grad()
        self._reset_grads()

    # ---- hybrid mode: eager tokenizer (ragged point clouds) + hipGraphs for everything behind the token matrix ------
    def _hybrid_setup(self, batch):
        """Split the parameters by stage, build the static inputs of the captured half and capture it."""
        from ..policy import fused_ops
        from .synthetic import clone_batch

        opt = self.optimizer
        index = {id(p): k for k, p in enumerate(opt.params)}
        tok = sorted(index[id(p)] for p in self.policy.tokenizer_parameters() if id(p) in index)
        tok_set = set(tok)
        self._subset_a, self._subset_a_set = tok, tok_set
        if len(self._stages) > 1:
            assert self._stages[-1].index_set == tok_set  # the last stage is the tokenizer (checked in __init__)
            self._stages_b = self._stages[:-1]
        else:
            self._stages_b = [_Stage(None, [k for k in range(len(opt.params)) if k not in tok_set], self._stages[0].slab)]
        with fused_ops.activate(self._fused_ctx), self._autocast(), torch.no_grad():
            ragged, rest = self.policy.hybrid_split(clone_batch(batch))
            buffers = {n: b.clone() for n, b in self.policy.named_buffers()}  # this probe must not count as training
            boundary = tuple(self._call_policy(ragged, stage="tokenize"))  # shapes / dtypes of the boundary
            for n, b in self.policy.named_buffers():
                b.copy_(buffers[n])
        self._static_sig = self._signature(rest)
        self._static_batch = self._clone_static(rest)
        # boundary[0] carries the gradient back to the tokenizer; the others (position embedding) are inputs only
        self._static_tokens = boundary[0].detach().clone().requires_grad_(True)
        self._static_extra = tuple(t.detach().clone() for t in boundary[1:])
        self._static_dtokens = torch.zeros_like(self._static_tokens)
        stages_b = self._stages_b

        def gen(first):
            def make_out():
                with fused_ops.activate(self._fused_ctx), self._autocast():
                    data = self.policy.hybrid_merge(clone_batch(self._static_batch), (self._static_tokens,) + self._static_extra)
                    return self._call_policy(data)

            self._static_tokens.grad = None
            for si, stats in self._segments(make_out, first, stages_b, leaf=self._static_tokens):
                if si == len(stages_b) - 1:  # the captured half ends at the static token matrix: hand its gradient over
                    self._static_dtokens.copy_(self._static_tokens.grad)
                yield si, stats

        self._warm_up(gen)
        self._reset_grads()
        self._graph, self._static_stats = self._capture_segments(lambda: gen(True), len(stages_b))
        self._reset_grads()
        self._graph_acc = None
        if self.accumulate > 1:  # later micro-batches of an accumulation window ADD their gradients
            self._graph_acc, self._static_stats_acc = self._capture_segments(lambda: gen(False), len(stages_b))
            self._reset_grads()

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
            # boundary tensors + the rest of the batch into the captured graphs' inputs: one table-driven copy launch
            pairs = [(self._static_tokens, tokens)] + list(zip(self._static_extra, boundary[1:]))
            self._copy_into(self._static_batch, rest, pairs)
            from .. import _lib

            _lib.copy_batch(pairs)
        first = self.micro % self.accumulate == 0
        for si, g in enumerate(self._graph if first else self._graph_acc):
            g.replay()
            if len(self._stages) > 1:
                self._exchange(si)
        tokens.backward(self._static_dtokens)
        self.optimizer.collect(first=first, subset=self._subset_a)
        self._exchange(len(self._stages) - 1)
        return (self._static_stats if first else self._static_stats_acc).clone()

    def prefetch_sampling(self, next_batch):
        """Hand the NEXT micro-batch over early (what a data loader's prefetch does): its FPS + kNN indices -- functions of
        the input coordinates only -- are computed on the side stream while the current step runs.  In hybrid mode the
        sampling otherwise sits in front of the captured graph; in graph mode it is kept OUT of the captured graph
        (static index buffers, `_static_sampling`) precisely so that it can run one batch ahead."""
        if next_batch is None or (self.mode == "graph" and not self._static_sampling):
            return
        pcds = self._pcds_of(next_batch)
        if pcds is None or not pcds["coord"].is_cuda:
            return
        target = self._sampling_target()
        if target is not None:
            target.prefetch_sampling(pcds)

    @staticmethod
    def _pcds_of(batch):
        return batch["pcds"] if "pcds" in batch else batch.get("obs", {}).get("pcds")

    def _sampling_target(self):
        pol = self.policy
        target = pol if hasattr(pol, "prefetch_sampling") else getattr(pol, "obs_encoder", None)
        if target is not None and getattr(target, "overlap_sampling", False) and hasattr(target, "prefetch_sampling"):
            return target
        return None

    def training_step(self, batch, prefetch=None):
        """One micro-batch: forward, loss, backward and -- on accumulation boundaries -- the
        optimizer step.  Returns the (detached, on-device) loss dict of this micro-batch.
        `prefetch`: the next micro-batch, if it is already on the device (see prefetch_sampling)."""
        # Mode contract.  .train() walks ~230 modules (1 ms of host time), so it runs only when the ROOT module is in eval
        # mode (what policy.eval() / a rollout helper leaves behind).  A submodule the caller put into eval() on its own
        # (a frozen backbone's BatchNorm) therefore STAYS in eval -- Lightning's loop would have forced it back to train
        # every step.  On the first and then every 64th micro-batch the tree is checked until something is found: submodules
        # in eval mode are reported with ONE warning (not an error: freezing is legitimate; BCTrainer(allow_eval_submodules=
        # True) skips the check).  Captured graphs bake the mode in at capture time either way.
        if not self.module.training:
            self.module.train()
        if self.micro % 64 == 0:
            for m, flag in self.__dict__.get("_tokenizer_flags", ()):
                if bool(getattr(m, "tokenizer_fp32", True)) != flag:
                    raise RuntimeError("tokenizer_fp32 of %s changed after the trainer was built: the optimizer's bf16 mirrors were derived "
                                       "from the old value -- set the flag (policy/precision.set_tokenizer_fp32) BEFORE constructing "
                                       "BCTrainer" % type(m).__name__)
        if self.micro % 64 == 0 and not self.allow_eval_submodules and not self.__dict__.get("_warned_eval"):
            stale = [n for n, m in self.module.named_modules() if not m.training]
            if stale:
                import warnings

                self._warned_eval = True
                warnings.warn("training_step: %d submodule(s) in eval mode under a training root (%s ...): their BatchNorm statistics / "
                              "dropout are frozen.  Call policy.train() if that is not intended; BCTrainer(allow_eval_submodules=True) "
                              "silences this." % (len(stale), ", ".join(stale[:4])))
        if prefetch is not None:
            self.prefetch_sampling(prefetch)
        if self._fused_ctx is not None:
            self._fused_ctx.set_step(self.micro)
        first = self.micro % self.accumulate == 0
        stepping = self._stepping = (self.micro + 1) % self.accumulate == 0
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
                if self._static_sampling:  # this batch's indices: prefetched during the previous step, or computed now
                    target = self._sampling_target()
                    target.load_static_sampling(target.sampling_for(self._pcds_of(batch), overlap=True))
                self._copy_into(self._static_batch, batch)
                use_acc = not (first or self._graph_acc is None)
                if self.segmented:  # one chain: graphs and the collectives between them, in capture order
                    (self._graph_acc if use_acc else self._graph)[0].replay()
                else:
                    for si, g in enumerate(self._graph_acc if use_acc else self._graph):
                        g.replay()
                        self._exchange(si)
                stats = (self._static_stats_acc if use_acc else self._static_stats).clone()
            else:
                if first:
                    self.optimizer.zero_grad()
                stats = self._forward_backward(batch)
            if stepping:
                self._finish_exchange()
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
