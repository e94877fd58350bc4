"""CPU: one steady-state training step issues no operator that synchronises the host with the device.

bench.py's headline mode replays the WHOLE step as one hipGraph (equal-size clouds); the graph is captured from the same Python that the
flat mode runs.  An operator that reads a device value on the host -- `.item()`, `bool(tensor)`, `nonzero`, `unique`, `masked_select`,
`torch.equal` -- aborts a capture on hardware ("operation not permitted when stream is capturing") and stalls the un-captured modes.
Rounds 5 and 6 changed the step's Python without a device to capture on; this runs the step on the host wave64 model (tests/wavesim; tensors
claim to be device tensors, so the fused paths are taken) under a dispatch-level probe and requires ZERO such operators in the third step
of an ACT and of a Diffusion-Policy trainer (the first steps may plan / allocate).  What it cannot see: syncs hidden inside library calls
on the device (none are made from this path: every kernel goes through the C ABI with raw pointers)."""
import collections
import traceback

import pytest
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from tests.wavesim import build as _build

pytestmark = pytest.mark.skipif(not __import__("os").path.exists(_build.CLANG), reason="needs the ROCm clang++ as host compiler (tests/wavesim)")

SYNC = ("_local_scalar_dense", "nonzero", "masked_select", "_unique", "unique_dim", "unique_consecutive", "aten.equal", "is_nonzero")


class SyncProbe(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.hits = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        full = str(func)
        if any(s in full for s in SYNC):
            fr = [f for f in traceback.extract_stack() if "pointcloudmatters_amd" in f.filename]
            where = "%s:%d %s" % (fr[-1].filename.split("pointcloudmatters_amd/")[-1], fr[-1].lineno, fr[-1].name) if fr else "?"
            self.hits[(full, where)] += 1
        return func(*args, **(kwargs or {}))


def test_the_probe_sees_a_host_read():
    with SyncProbe() as p:
        t = torch.arange(4.0)
        float(t.sum())
        bool((t > 1).any())
        torch.nonzero(t)
    assert sum(p.hits.values()) == 3, p.hits


@pytest.mark.parametrize("which", ["act", "dp"])
def test_steady_state_step_has_no_host_synchronisation(which):
    from tests.wavesim.backend import simulated_device

    with simulated_device(claim_cuda=True) as dev:
        from pointcloudmatters_amd.bc import DP_OPTIM, BCTrainer, build_act_policy, build_dp_policy, clone_batch, make_act_batch, make_dp_batch

        torch.manual_seed(0)
        if which == "act":
            policy = build_act_policy(pcd_npoints=64, sa_impl="fused").to(dev)
            batches = [make_act_batch(2, 256, seed=1000 + i, device=dev) for i in range(3)]
            optim = dict(accumulate_grad_batches=1)
        else:
            from tests.golden.make_golden import DP_SMALL

            policy = build_dp_policy(pcd_npoints=32, sa_impl="fused", **DP_SMALL).to(dev)
            batches = [make_dp_batch(3, 200, seed=12 + i, device=dev) for i in range(3)]
            optim = dict(DP_OPTIM)
        tr = BCTrainer(policy, total_steps=100, precision="bf16", device=dev, mode="flat", optim=optim)
        for i in range(2):
            tr.training_step(clone_batch(batches[i]), prefetch=batches[i + 1])
        with SyncProbe() as probe:
            out = tr.training_step(clone_batch(batches[2]), prefetch=batches[0])
        assert torch.isfinite(out["loss"]).item()
    assert not probe.hits, "host-synchronising operators in a steady-state step: %s" % dict(probe.hits)
