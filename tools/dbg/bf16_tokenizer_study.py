"""Build container only (imports the reference from /root/reference through tests/golden/make_golden.py): where does bf16 autocast hurt?
Per-tensor gradient error (max |g16 - g32| / max |g32|) of the REFERENCE's ACTPCD at the shipped widths, and of the Diffusion Policy
(our classes, pinned to the reference's at 1e-4), under torch.autocast("cpu", bf16) with parts of the tokenizer kept in fp32:
  ''   everything under autocast        'B'  backbone (PointNet) in fp32       'S'  set-abstraction layer in fp32       'BS' both
  'E'  (Diffusion Policy) the whole observation encoder in fp32
Result: profiles/r05_bf16_tokenizer_study.log; consequence: pointcloudmatters_amd/policy/precision.py.
    python tools/dbg/bf16_tokenizer_study.py > profiles/r05_bf16_tokenizer_study.log"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

torch.set_num_threads(8)
from oracle import pointops_cpu
from pointcloudmatters_amd.bc import build_dp_policy, make_act_batch, make_dp_batch
from pointcloudmatters_amd.policy import PointNet
from tests.golden import make_golden as mg
from tests.util import seeded_fill

ref = mg.install_reference()
c = mg.WIDE


def off(f):
    def g(*a, **k):
        with torch.autocast("cpu", enabled=False):
            return f(*a, **k)
    return g


def act(B, mode, seed=300):
    backbone = PointNet(in_channels=6, num_classes=0)
    model = mg._ref_act(ref, c, 128, backbone)
    seeded_fill(model, mg.WIDE_SEED)
    model.train()
    if "B" in mode:
        backbone.forward = off(backbone.forward)
    if "S" in mode:
        model.pcd_sampling = off(model.pcd_sampling)
    eps = torch.randn(B, c["latent_dim"], generator=torch.Generator().manual_seed(25))
    batch = make_act_batch(B, 150, seed=seed, ragged=True, num_queries=c["num_queries"])
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=mode != "fp32"):
        mg._run_ref_act(ref, model, batch, eps)
    return {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}


def dp(B, mode, seed=500):
    pol = build_dp_policy(pcd_npoints=64, pointops=pointops_cpu, sa_impl="reference", **mg.WIDE_DP)
    seeded_fill(pol, mg.WIDE_SEED + 1)
    pol.train()
    pol.obs_encoder.tokenizer_fp32 = mode == "E"
    batch = make_dp_batch(B, 100, seed=seed, ragged=True)
    gen = torch.Generator().manual_seed(38)
    batch["noise"], batch["timesteps"] = torch.randn(B, 16, 7, generator=gen), torch.randint(0, 100, (B,), generator=gen)
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=mode != "fp32"):
        out = pol(batch)
    out["loss"].backward()
    return {n: p.grad.detach().float().clone() for n, p in pol.named_parameters() if p.grad is not None}


def report(tag, g32, g16):
    errs = {n: ((g16[n] - g32[n]).abs().max().item() / g32[n].abs().max().item()) for n in g32 if g32[n].abs().max().item() >= 1e-6}
    v = np.array(list(errs.values()))
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(f"{tag}: {len(v)} tensors, median {np.median(v):.4f}  p90 {np.quantile(v, .9):.4f}  worst {v.max():.4f}   "
          + ", ".join(f"{n} {e:.3f}" for n, e in worst), flush=True)


print("== ACT: the reference's ACTPCD (d = 512, 8 heads, ffn 32, 1 enc + 2 dec layers, 131-token sequences), CPU autocast bf16 vs fp32")
for B in (2, 8, 16):
    g32 = act(B, "fp32")
    for mode in ("", "B", "S", "BS"):
        report(f"B={B:2d} fp32 parts '{mode:2s}'", g32, act(B, mode))
print("== Diffusion Policy (PointNet head 96, SA 96, projector [96, 128, 128], U-Net 128 / 256), CPU autocast bf16 vs fp32")
for B in (2, 8):
    g32 = dp(B, "fp32")
    for mode in ("", "E"):
        report(f"B={B:2d} fp32 parts '{mode:2s}'", g32, dp(B, mode))
