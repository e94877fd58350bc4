"""Policy-side rollout path: what runs between two simulator steps.

* `TemporalAgg` -- ACT's temporal ensembling of overlapping action chunks
  (/root/reference/src/utils/misc.py:88-141, used by maniskill2_act_bc_module.py:154-161).
* `GraphedPolicy` -- replays one policy call (`ACTPCD.forward` without actions, act.py:177-182, or
  `DiffusionUnetPcdPolicy.predict_action` with its 100-iteration DDPM loop) as ONE hipGraph: a rollout step is
  latency-bound (b = 1-2 clouds, thousands of small launches), so launch overhead is the thing to remove.
"""
import numpy as np
import torch


class TemporalAgg:
    """Exponentially weighted average of every chunk's prediction for the current time step.

    buffer[r] holds the chunk predicted r calls ago shifted so that column c is "c steps after that call"; the
    prediction for *now* made by the chunk stored in row r (oldest first) is buffer[r, rows-1-r].  As in the
    reference, a row counts as populated iff it holds any non-zero entry, weights are exp(-k*i) with i = 0 for the
    OLDEST chunk, and until `chunk_size` chunks have arrived only the populated rows take part.
    """

    def __init__(self, apply=False, action_dim=8, chunk_size=20, k=0.01):
        self.apply = apply
        if apply:
            self.action_dim, self.chunk_size, self.k = action_dim, chunk_size, k
            self.full_action = False
            self.reset()

    def reset(self):
        # NB: like the reference, reset() does not clear `full_action`.
        self.action_buffer = np.zeros((self.chunk_size, self.chunk_size, self.action_dim))

    def _populated(self):
        if self.full_action:
            return self.chunk_size
        return int(np.count_nonzero(np.abs(self.action_buffer).reshape(self.chunk_size, -1).sum(axis=1)))

    def add_action(self, action):
        if self.full_action:
            self.action_buffer[:-1] = self.action_buffer[1:].copy()
            self.action_buffer[-1] = action
            return
        row = self._populated()
        self.action_buffer[row] = action
        if row == self.chunk_size - 1:
            self.full_action = True

    def get_action(self):
        rows = self._populated()
        w = np.exp(-self.k * np.arange(rows))
        w = w / w.sum()
        # row r (r = 0 oldest) contributes its column rows-1-r
        now = self.action_buffer[np.arange(rows), rows - 1 - np.arange(rows)]
        return (now * w[:, None]).sum(0)

    def __call__(self, action):
        if not self.apply:
            return action[0]
        self.add_action(action)
        return self.get_action()


def _static_like(obj):
    if torch.is_tensor(obj):
        t = obj.clone()
        if hasattr(obj, "_pcm_host"):
            t._pcm_host = obj._pcm_host
        return t
    if isinstance(obj, dict):
        return {k: _static_like(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_static_like(v) for v in obj)
    return obj


def _copy_into(dst, src):
    if torch.is_tensor(dst):
        dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_into(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            _copy_into(d, s)


class Bf16Weights:
    """bf16 copies of the parameters that autocast would otherwise re-cast on EVERY use (autocast's weight cache is
    off under no_grad): Linear / Conv1d / attention-projection weights and biases.  Inside the context the modules see
    the copies; `refresh()` re-casts from the fp32 masters into the same storage (safe for a captured graph)."""

    def __init__(self, policy):
        from ..bc.trainer import bf16_consumed_parameters
        from . import fused_ops

        ids = bf16_consumed_parameters(policy, fused_ffn=fused_ops.current() is not None)
        self.items = []
        for mod in policy.modules():
            for pname, p in list(mod._parameters.items()):
                if p is not None and id(p) in ids and p.dtype == torch.float32 and p.is_cuda:
                    shadow = torch.nn.Parameter(p.detach().to(torch.bfloat16), requires_grad=False)
                    self.items.append((mod, pname, p, shadow))

    def refresh(self):
        with torch.no_grad():
            for _, _, p, shadow in self.items:
                shadow.copy_(p)

    def __enter__(self):
        for mod, pname, _, shadow in self.items:
            mod._parameters[pname] = shadow
        return self

    def __exit__(self, *exc):
        for mod, pname, p, _ in self.items:
            mod._parameters[pname] = p
        return False


class GraphedPolicy:
    """Capture `fn(static_inputs)` once, then `__call__(inputs)` = copy inputs into the static buffers + one
    hipGraphLaunch.  Shapes (and the clouds' offsets) must stay those of the example: in a rollout they do -- every
    observation is resampled to the same number of points by the data transform before it reaches the policy.

    fn must be free of host synchronisation (offsets carry their host copy, see pointops._common.host_offsets).
    """

    def __init__(self, fn, example_inputs, warmup=2, autocast_dtype=torch.bfloat16, policy=None):
        dev = next(t for t in _flatten(example_inputs) if torch.is_tensor(t)).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedPolicy needs a HIP device: there is no CPU path")
        self.static_in = _static_like(example_inputs)
        self._fn, self._dtype = fn, autocast_dtype
        # bf16 weight copies made once (call .weights.refresh() after the fp32 masters change)
        self.weights = Bf16Weights(policy) if (policy is not None and autocast_dtype == torch.bfloat16) else None
        self._stream = torch.cuda.Stream(device=dev)
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        import contextlib

        with (self.weights if self.weights is not None else contextlib.nullcontext()):
            with torch.cuda.stream(self._stream):
                for _ in range(warmup):
                    self._run()
            torch.cuda.current_stream(dev).wait_stream(self._stream)
            torch.cuda.synchronize(dev)
            from .._graphs import finalize, new_graph

            self.graph = new_graph()
            with torch.cuda.graph(self.graph, stream=self._stream, capture_error_mode="thread_local"):
                self.static_out = self._run()
            finalize(self.graph)  # memset nodes -> kernel nodes (see _graphs.py), then instantiate

    def _run(self):
        with torch.no_grad(), torch.autocast("cuda", dtype=self._dtype, enabled=self._dtype is not None):
            return self._fn(self.static_in)

    def __call__(self, inputs=None):
        if inputs is not None:
            _copy_into(self.static_in, inputs)
        self.graph.replay()
        return self.static_out


def _flatten(obj):
    if isinstance(obj, dict):
        for v in obj.values():
            yield from _flatten(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _flatten(v)
    else:
        yield obj


def graphed_act(policy, example):
    """ACT rollout call: example = {"qpos", "goal_cond"?, "pcds"} (no "actions") -> static a_hat (B, T, Da)."""
    policy.eval()

    def fn(d):
        return policy(dict(d, pcds=dict(d["pcds"])))["a_hat"]

    return GraphedPolicy(fn, example, policy=policy)


def graphed_dp(policy, example):
    """Diffusion-Policy rollout call: example = {"obs": {...}} -> static {"action", "action_pred"}; the whole
    observation encoder + 100-iteration sampler is one graph (the variance noise is drawn inside it by the
    graph-safe Philox generator, so every replay samples afresh)."""
    policy.eval()

    def fn(d):
        obs = dict(d["obs"])
        if "pcds" in obs:
            obs["pcds"] = dict(obs["pcds"])
        return policy.predict_action({"obs": obs})

    return GraphedPolicy(fn, example, policy=policy)
