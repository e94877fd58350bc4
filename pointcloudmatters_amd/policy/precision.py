"""Where bf16 autocast applies and where it does not.

`precision="bf16"` means: the dense GEMMs and the attention of the ACT transformer / the Diffusion-Policy U-Net run on the matrix
cores in bf16 (BASELINE.json north_star: "MFMA is reserved only for the ACT transformer and Diffusion-Policy U-Net dense GEMMs").
The TOKENIZER -- the PointNet per-point MLP, the set-abstraction layer and, for the Diffusion Policy, the projector behind it
(/root/reference/src/models/components/pcd_encoder/pointnet.py:16-85, act/act.py:384-465,
diffusion_policy/vision/pcd_obs_encoder.py:100-239) -- stays in fp32 by default.

Why (measured with the REFERENCE's own classes under torch.autocast on the host, profiles/r05_bf16_tokenizer_study.log; the numbers
are per-tensor gradient errors against the fp32 run, max |difference| / max |reference|):

                                      tokenizer under autocast          tokenizer in fp32
    ACT   (d = 512, 2 / 8 samples)    median 0.8 %, worst 25 %          median 0.6 %, worst 5 %
    DP    (2 / 8 samples)             median 7.6 %, worst 68 - 123 %    median 1.2 %, worst 3 %

Every layer of the tokenizer is a Linear followed by a training-mode BatchNorm: the BatchNorm divides by the per-channel standard
deviation of its input, while bf16 rounds that input relative to its MAGNITUDE (mean included), and the backward pass subtracts two
nearly equal sums.  Six such layers in sequence turn a 2^-9 rounding into gradients that are tens of per cent off -- in ANY bf16
evaluation, the framework's included -- and the U-Net then trains on a perturbed condition.  The tokenizer is 1 - 3 % of the step's
FLOPs and bound by gathers, not by the matrix cores, so bf16 buys next to nothing there.  A half-way recipe does not help: bf16
OPERANDS with fp32 accumulation and fp32 outputs in the tokenizer's products (what a bf16-in / fp32-out GEMM would give at bf16 speed)
leave the worst tensor at 21 - 28 % (same log, rows "bS" / "bs"): the chain amplifies a 2^-9 perturbation wherever it enters.

`tokenizer_fp32` is a class attribute of ACTPCD / PCDObsEncoder (True); set it to False on an instance (or through
`set_tokenizer_fp32`) for the previous behaviour, autocast everywhere.
"""
import contextlib

import torch


def tokenizer_autocast(owner, like):
    """Context for the tokenizer of `owner`: autocast switched OFF when the owner keeps its tokenizer in fp32 and an autocast
    region is active for `like`'s device; otherwise nothing."""
    kind = like.device.type
    if getattr(owner, "tokenizer_fp32", True) and torch.is_autocast_enabled(kind):
        return torch.autocast(kind, enabled=False)
    return contextlib.nullcontext()


def tokenizer_owners(policy):
    """The modules of `policy` that own a point-cloud tokenizer (they carry the `tokenizer_fp32` switch and `tokenizer_modules()`)."""
    return [m for m in policy.modules() if hasattr(m, "tokenizer_modules")]


def set_tokenizer_fp32(policy, flag):
    owners = tokenizer_owners(policy)
    for m in owners:
        m.tokenizer_fp32 = bool(flag)
    return len(owners)


def fp32_tokenizer_parameter_ids(policy):
    """ids of the parameters consumed in fp32 although the step runs under bf16 autocast: no bf16 mirror / shadow for them."""
    ids = set()
    for owner in tokenizer_owners(policy):
        if getattr(owner, "tokenizer_fp32", True):
            for mod in owner.tokenizer_modules():
                ids.update(id(p) for p in mod.parameters())
    return ids
