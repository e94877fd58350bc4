#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE'S OWN PYTHON.

Runs only in the build container (needs /root/reference, which does not exist on the GPU box);
the .npz files it writes are committed and are the only thing the tests read.

What is imported from the reference, unmodified, by file path:
  src/models/components/act/{act,transformer,utils}.py      ACTPCD, Transformer, reparametrize ...
  src/models/components/loss/misc.py                        KLDivergence
  libs/pointops/functions/grouping.py                       the pure-PyTorch grouping()
  src/utils/{sparse_tensor_utils,rotation_conversions}.py   offset2batch, collate helpers

What has to be substituted, and why:
  * `pointops` (FPS / kNN): the reference's native module is CUDA-only and cannot be built here;
    the oracle's API (oracle/pointops_cpu.py) stands in.  => index kernels are NOT pinned by these
    fixtures (they are pinned by oracle == py_twin and property tests); everything downstream is.
  * `pointops._C`: an empty stub so that functions/grouping.py imports; grouping() never calls it.
  * the PointNet backbone: the reference's is built from spconv (third-party CUDA library, absent);
    ACTPCD takes the backbone as a constructor argument, so our Linear/BN restatement is passed in.
  * `torchvision.transforms.ToTensor`, `src.utils` package __init__: import-time only, unused.
  * `reparametrize` is wrapped to use a recorded eps (the reference draws it from the global RNG).
Dropout is 0 so that no other randomness enters.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def install_reference():
    from oracle import pointops_cpu

    # --- substitutes ---------------------------------------------------------------------------
    sys.modules["pointops"] = pointops_cpu
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class ToTensor:  # base class of act.py:33 ToTensorIfNot, never instantiated on the pcd path
        pass

    tvt.ToTensor = ToTensor
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    # --- reference packages, by path, without running their __init__.py (lightning/hydra imports) --
    _pkg("src", f"{REF}/src")
    utils = _pkg("src.utils", f"{REF}/src/utils")
    stu = _load("src.utils.sparse_tensor_utils", f"{REF}/src/utils/sparse_tensor_utils.py")
    utils.offset2batch = stu.offset2batch
    _load("src.utils.rotation_conversions", f"{REF}/src/utils/rotation_conversions.py")
    _pkg("src.models", f"{REF}/src/models")
    _pkg("src.models.components", f"{REF}/src/models/components")
    _pkg("src.models.components.act", f"{REF}/src/models/components/act")
    _pkg("src.models.components.loss", f"{REF}/src/models/components/loss")
    ref = types.SimpleNamespace()
    ref.act_utils = _load("src.models.components.act.utils", f"{REF}/src/models/components/act/utils.py")
    ref.transformer = _load("src.models.components.act.transformer", f"{REF}/src/models/components/act/transformer.py")
    ref.act = _load("src.models.components.act.act", f"{REF}/src/models/components/act/act.py")
    ref.loss = _load("src.models.components.loss.misc", f"{REF}/src/models/components/loss/misc.py")
    ref.collate = stu
    # the reference's pure-PyTorch grouping(), with an inert pointops._C
    fake_c = types.ModuleType("pointops_ref._C")
    fake_c.grouping_backward_cuda = fake_c.grouping_forward_cuda = None
    _pkg("pointops_ref", f"{REF}/libs/pointops/functions")
    sys.modules["pointops._C"] = fake_c
    ref.grouping = _load("pointops_ref.grouping", f"{REF}/libs/pointops/functions/grouping.py")
    return ref


SMALL = dict(hidden_dim=48, nhead=4, dim_feedforward=32, num_encoder_layers=2, num_decoder_layers=3, dropout=0.0,
             latent_dim=8, num_queries=10, kl_weight=10.0, action_dim=7, qpos_dim=9, goal_cond_dim=3, pcd_nsample=16)


def build_ours(pcd_npoints, seed):
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc import build_act_policy

    torch.manual_seed(seed)
    return build_act_policy(pcd_npoints=pcd_npoints, pointops=pointops_cpu, sa_impl="reference", **SMALL)


def build_reference_actpcd(ref, ours, pcd_npoints):
    from pointcloudmatters_amd.policy import PointNet

    c = SMALL
    backbone = PointNet(in_channels=6, num_classes=0)
    transformer = ref.transformer.Transformer(
        d_model=c["hidden_dim"], dropout=c["dropout"], nhead=c["nhead"], dim_feedforward=c["dim_feedforward"],
        num_encoder_layers=c["num_encoder_layers"], num_decoder_layers=c["num_decoder_layers"], normalize_before=False,
        return_intermediate_dec=True)
    encoder = ref.transformer.TransformerEncoder(
        d_model=c["hidden_dim"], dropout=c["dropout"], nhead=c["nhead"], dim_feedforward=c["dim_feedforward"],
        num_layers=c["num_encoder_layers"], normalize_before=False, activation="relu")
    model = ref.act.ACTPCD(
        backbone=backbone, transformer=transformer, encoder=encoder, hidden_dim=c["hidden_dim"],
        num_queries=c["num_queries"], num_cameras=1, action_dim=c["action_dim"], qpos_dim=c["qpos_dim"], env_state_dim=0,
        latent_dim=c["latent_dim"], action_loss=torch.nn.MSELoss(reduction="none"), klloss=ref.loss.KLDivergence(),
        kl_weight=c["kl_weight"], goal_cond_dim=c["goal_cond_dim"], pcd_nsample=c["pcd_nsample"], pcd_npoints=pcd_npoints)
    missing, unexpected = model.load_state_dict(ours.state_dict(), strict=True), None
    return model


def golden_act(ref):
    from pointcloudmatters_amd.bc import make_act_batch

    pcd_npoints = 32
    ours = build_ours(pcd_npoints, seed=1234)
    model = build_reference_actpcd(ref, ours, pcd_npoints)
    model.train()  # BatchNorm uses batch statistics, exactly like training_step
    batch = make_act_batch(3, 180, seed=77, ragged=True, num_queries=SMALL["num_queries"])
    eps = torch.randn(3, SMALL["latent_dim"], generator=torch.Generator().manual_seed(5))
    orig = ref.act.reparametrize
    ref.act.reparametrize = lambda mu, logvar: mu + logvar.div(2).exp() * eps
    try:
        dd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in batch.items()}
        dd["pcds"]["offset"] = dd["pcds"]["offset"].clone()
        out = model(dd)
        out["loss"].backward()
    finally:
        ref.act.reparametrize = orig
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    keep = ["linear.weight", "bn.weight", "backbone.conv1.0.weight", "backbone.conv5.0.weight",
            "transformer.encoder.layers.0.self_attn.in_proj_weight", "transformer.decoder.layers.0.multihead_attn.out_proj.weight",
            "transformer.decoder.layers.2.linear1.weight", "encoder.layers.1.linear2.weight", "latent_proj.weight",
            "action_head.weight", "query_embed.weight", "additional_pos_embed.weight", "input_proj_robot_state.weight"]
    fx = {"eps": eps.numpy()}
    for k, v in batch.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                fx[f"in.pcds.{kk}"] = vv.numpy()
        else:
            fx[f"in.{k}"] = v.numpy()
    for k, v in ours.state_dict().items():
        fx[f"w.{k}"] = v.numpy()
    for k in ("a_hat", "is_pad_hat", "mu", "logvar", "loss", "action_loss", "kl_loss", "src", "pos"):
        fx[f"out.{k}"] = out[k].detach().numpy()
    for k in keep:
        fx[f"grad.{k}"] = grads[k].numpy()
    fx["meta.grad_none"] = np.array(sorted(n for n, p in model.named_parameters() if p.grad is None))
    fx["meta.bn_running_mean"] = model.bn.running_mean.numpy()
    fx["meta.bn_running_var"] = model.bn.running_var.numpy()
    np.savez_compressed(os.path.join(OUT, "act_pcd_small.npz"), **fx)
    print("act_pcd_small.npz: loss", float(out["loss"]), "action", float(out["action_loss"]), "kl", float(out["kl_loss"]))


def golden_grouping(ref):
    g = torch.Generator().manual_seed(3)
    n, m, k, c = 50, 12, 16, 5
    xyz = torch.randn(n, 3, generator=g)
    new_xyz = torch.randn(m, 3, generator=g)
    feat = torch.randn(n, c, generator=g, requires_grad=True)
    idx = torch.randint(0, n, (m, k), generator=g, dtype=torch.int32)
    idx[2, 9:] = -1  # placeholders as produced for clouds smaller than nsample
    idx[7, 1:] = -1
    fx = {"xyz": xyz.numpy(), "new_xyz": new_xyz.numpy(), "feat": feat.detach().numpy(), "idx": idx.numpy()}
    for with_xyz in (True, False):
        out = ref.grouping.grouping(idx, feat, xyz, new_xyz, with_xyz=with_xyz)
        gout = torch.randn(out.shape, generator=g)
        (grad,) = torch.autograd.grad(out, feat, gout)
        tag = "xyz" if with_xyz else "feat"
        fx[f"out.{tag}"] = out.detach().numpy()
        fx[f"gout.{tag}"] = gout.numpy()
        fx[f"grad_feat.{tag}"] = grad.numpy()
    np.savez_compressed(os.path.join(OUT, "grouping_ref.npz"), **fx)
    print("grouping_ref.npz ok")


def golden_misc(ref):
    """Small deterministic pieces: sinusoid table, KL, collate/offset helpers."""
    fx = {}
    fx["sinusoid_12_48"] = ref.act_utils.get_sinusoid_encoding_table(12, 48).numpy()
    g = torch.Generator().manual_seed(9)
    mu, logvar = torch.randn(4, 8, generator=g), torch.randn(4, 8, generator=g)
    fx["kl.mu"], fx["kl.logvar"] = mu.numpy(), logvar.numpy()
    fx["kl.out"] = ref.loss.KLDivergence()(mu, logvar).numpy()
    off = torch.tensor([5, 9, 9, 14])
    fx["o2b.offset"] = off.numpy()
    fx["o2b.batch"] = ref.collate.offset2batch(off).numpy()
    fx["b2o.offset"] = ref.collate.batch2offset(ref.collate.offset2batch(torch.tensor([5, 9, 14]))).numpy()
    np.savez_compressed(os.path.join(OUT, "misc_ref.npz"), **fx)
    print("misc_ref.npz ok")


if __name__ == "__main__":
    assert os.path.isdir(REF), "run this in the build container (needs /root/reference)"
    torch.set_num_threads(1)
    ref = install_reference()
    golden_act(ref)
    golden_grouping(ref)
    golden_misc(ref)
