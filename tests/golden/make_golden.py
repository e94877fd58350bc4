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
OUT = os.environ.get("PCM_GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))  # PCM_GOLDEN_OUT: regenerate elsewhere (tools/check_golden_regen.py)
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
    # ---- Diffusion-Policy pieces ----------------------------------------------------------------
    import logging

    class RankedLogger(logging.LoggerAdapter):  # src.utils.RankedLogger is only used for one info() line
        def __init__(self, name, rank_zero_only=True):
            super().__init__(logging.getLogger(name), {})

    utils.RankedLogger = RankedLogger
    _load("src.utils.pytorch_utils", f"{REF}/src/utils/pytorch_utils.py")
    dpu = _pkg("src.utils.diffusion_policy", f"{REF}/src/utils/diffusion_policy")
    mam = _load("src.utils.diffusion_policy.module_attr_mixin", f"{REF}/src/utils/diffusion_policy/module_attr_mixin.py")
    dpu.ModuleAttrMixin = mam.ModuleAttrMixin
    base = f"{REF}/src/models/components/diffusion_policy"
    _pkg("src.models.components.diffusion_policy", base)
    _pkg("src.models.components.diffusion_policy.diffusion", f"{base}/diffusion")
    _pkg("src.models.components.diffusion_policy.vision", f"{base}/vision")
    pfx = "src.models.components.diffusion_policy"
    _load(f"{pfx}.diffusion.conv1d_components", f"{base}/diffusion/conv1d_components.py")
    _load(f"{pfx}.diffusion.positional_embedding", f"{base}/diffusion/positional_embedding.py")
    ref.unet = _load(f"{pfx}.diffusion.conditional_unet1d", f"{base}/diffusion/conditional_unet1d.py")
    ref.maskgen = _load(f"{pfx}.diffusion.mask_generator", f"{base}/diffusion/mask_generator.py")
    ref.pcd_enc = _load(f"{pfx}.vision.pcd_obs_encoder", f"{base}/vision/pcd_obs_encoder.py")
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


RLB_SMALL = dict(SMALL, action_dim=11, qpos_dim=11, goal_cond_dim=16, rot_type="6d", collision=True, position_loss_weight=10.0)


def golden_rlbench(ref):
    """Reference ACTRLBenchPCD (act.py:707-825): training forward / backward with the weighted position loss and the sigmoid
    gripper / collision outputs, and the rollout branch (6-D rotation -> quaternion through rotation_conversions.py)."""
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc import build_rlbench_act_policy, make_act_batch
    from pointcloudmatters_amd.policy import PointNet

    c = RLB_SMALL
    pcd_npoints = 32
    torch.manual_seed(4321)
    kw = {k: v for k, v in c.items() if k not in ("rot_type", "collision", "position_loss_weight")}
    ours = build_rlbench_act_policy(pcd_npoints=pcd_npoints, pointops=pointops_cpu, sa_impl="reference", **kw)
    transformer = ref.transformer.Transformer(
        d_model=c["hidden_dim"], dropout=c["dropout"], nhead=c["nhead"], dim_feedforward=c["dim_feedforward"],
        num_encoder_layers=c["num_encoder_layers"], num_decoder_layers=c["num_decoder_layers"], normalize_before=False,
        return_intermediate_dec=True)
    encoder = ref.transformer.TransformerEncoder(
        d_model=c["hidden_dim"], dropout=c["dropout"], nhead=c["nhead"], dim_feedforward=c["dim_feedforward"],
        num_layers=c["num_encoder_layers"], normalize_before=False, activation="relu")
    model = ref.act.ACTRLBenchPCD(
        backbone=PointNet(in_channels=6, num_classes=0), transformer=transformer, encoder=encoder, hidden_dim=c["hidden_dim"],
        num_queries=c["num_queries"], num_cameras=1, action_dim=c["action_dim"], qpos_dim=c["qpos_dim"], env_state_dim=0,
        latent_dim=c["latent_dim"], action_loss=torch.nn.MSELoss(reduction="none"), klloss=ref.loss.KLDivergence(),
        kl_weight=c["kl_weight"], goal_cond_dim=c["goal_cond_dim"], pcd_nsample=c["pcd_nsample"], pcd_npoints=pcd_npoints,
        rot_type="6d", collision=True, position_loss_weight=c["position_loss_weight"])
    model.load_state_dict(ours.state_dict(), strict=True)
    model.train()
    batch = make_act_batch(3, 160, seed=88, ragged=True, num_queries=c["num_queries"], action_dim=c["action_dim"],
                           qpos_dim=c["qpos_dim"], goal_cond_dim=c["goal_cond_dim"])
    batch["actions"][..., -2:] = (batch["actions"][..., -2:] > 0).float()  # gripper / collision targets are 0 / 1
    eps = torch.randn(3, c["latent_dim"], generator=torch.Generator().manual_seed(6))
    orig = ref.act.reparametrize
    ref.act.reparametrize = lambda mu, logvar: mu + logvar.div(2).exp() * eps
    try:
        dd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in batch.items()}
        out = model(dd)
        out["loss"].backward()
    finally:
        ref.act.reparametrize = orig
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    fx = {"eps": eps.numpy()}
    for k, v in batch.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                fx[f"in.pcds.{kk}"] = vv.numpy()
        else:
            fx[f"in.{k}"] = v.numpy()
    for k, v in ours.state_dict().items():
        fx[f"w.{k}"] = v.numpy()
    for k in ("a_hat", "mu", "logvar", "loss", "action_loss", "kl_loss"):
        fx[f"out.{k}"] = out[k].detach().numpy()
    for k in ("action_head.weight", "action_head.bias", "linear.weight", "backbone.conv1.0.weight", "latent_proj.weight",
              "transformer.decoder.layers.0.multihead_attn.out_proj.weight", "proj_goal_cond_emb.weight"):
        fx[f"grad.{k}"] = grads[k].numpy()
    # rollout branch: no actions -> zero latent, quaternion output
    model.eval()
    ev = make_act_batch(3, 160, seed=88, ragged=True, num_queries=c["num_queries"], action_dim=c["action_dim"],
                        qpos_dim=c["qpos_dim"], goal_cond_dim=c["goal_cond_dim"])
    ev.pop("actions"), ev.pop("is_pad")
    with torch.no_grad():
        eo = model(ev)
    fx["eval.a_hat"] = eo["a_hat"].numpy()
    np.savez_compressed(os.path.join(OUT, "act_rlbench_small.npz"), **fx)
    print("act_rlbench_small.npz: loss", float(out["loss"]), "eval a_hat", tuple(eo["a_hat"].shape))


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
    # pcd_collate_fn on two ACT-style samples and two DP-style samples (sparse_tensor_utils.py:36-82)
    g2 = torch.Generator().manual_seed(21)

    def cloud(n):
        return {"coord": torch.randn(n, 3, generator=g2), "grid_coord": torch.randint(0, 50, (n, 3), generator=g2),
                "feat": torch.randn(n, 6, generator=g2), "offset": torch.tensor([n])}

    sizes = [5, 3, 4, 6]
    clouds = [cloud(n) for n in sizes]
    for i, c in enumerate(clouds):
        for k, v in c.items():
            fx[f"collate.in.{i}.{k}"] = v.numpy()
    import copy
    act_samples = [{"pcds": [copy.deepcopy(clouds[i])], "qpos": torch.full((9,), float(i))} for i in range(2)]
    out = ref.collate.pcd_collate_fn(act_samples)
    for k, v in out["pcds"].items():
        fx[f"collate.act.pcds.{k}"] = v.numpy()
    fx["collate.act.qpos"] = out["qpos"].numpy()
    dp_samples = [{"obs": {"pcds": [copy.deepcopy(clouds[2 * i]), copy.deepcopy(clouds[2 * i + 1])], "qpos": torch.full((2, 9), float(i))},
                   "action": torch.full((4, 7), float(i))} for i in range(2)]
    out = ref.collate.pcd_collate_fn(dp_samples)
    for k, v in out["obs"]["pcds"].items():
        fx[f"collate.dp.pcds.{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "misc_ref.npz"), **fx)
    print("misc_ref.npz ok")


DP_SMALL = dict(down_dims=(16, 32, 64), diffusion_step_embed_dim=16, pcd_num_classes=24, pcd_hidden_dim=24,
                projector_channels=(24, 40, 40), n_groups=8)


def golden_dp(ref):
    """Reference PCDObsEncoder + ConditionalUnet1D + LowdimMaskGenerator, composed as compute_loss
    does (diffusion_unet_image_policy.py:233-313; that file itself needs diffusers and is not importable)."""
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc import build_dp_policy, make_dp_batch
    from pointcloudmatters_amd.policy import PointNet

    pcd_npoints = 32
    torch.manual_seed(4321)
    ours = build_dp_policy(pcd_npoints=pcd_npoints, pointops=pointops_cpu, sa_impl="reference", **DP_SMALL)
    sd = ours.state_dict()
    shape_meta = {"obs": {"pcds": {"shape": [6], "type": "pcd"}, "qpos": {"shape": [9], "type": "low_dim"}},
                  "action": {"shape": [7]}}
    enc = ref.pcd_enc.PCDObsEncoder(shape_meta=shape_meta, pcd_model=PointNet(in_channels=6, num_classes=24),
                                    share_pcd_model=True, n_obs_step=2, pcd_nsample=16, pcd_npoints=pcd_npoints,
                                    pcd_hidden_dim=24, projector_layers=1, projector_channels=[24, 40, 40])
    enc.load_state_dict({k[len("obs_encoder."):]: v for k, v in sd.items() if k.startswith("obs_encoder.")}, strict=True)
    unet = ref.unet.ConditionalUnet1D(input_dim=7, local_cond_dim=None, global_cond_dim=(40 + 9) * 2,
                                      diffusion_step_embed_dim=16, down_dims=[16, 32, 64], kernel_size=5, n_groups=8,
                                      cond_predict_scale=True)
    unet.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}, strict=True)
    mg = ref.maskgen.LowdimMaskGenerator(action_dim=7, obs_dim=0, max_n_obs_steps=2, fix_obs_steps=True, action_visible=False)
    enc.train(), unet.train()
    batch = make_dp_batch(3, 150, seed=11, ragged=True)
    g = torch.Generator().manual_seed(8)
    noise = torch.randn(3, 16, 7, generator=g)
    timesteps = torch.tensor([3, 57, 99])
    # --- compute_loss, with the identity-range normaliser the builder installs (scale 1, offset 0)
    qpos, action = batch["obs"]["qpos"], batch["action"]
    this_nobs = {"qpos": qpos[:, :2].reshape(-1, 9), "pcds": {k: v.clone() for k, v in batch["obs"]["pcds"].items()}}
    feat = enc(this_nobs)
    global_cond = feat.reshape(3, -1)
    mask = mg((3, 16, 7))
    acp = ours.noise_scheduler.alphas_cumprod[timesteps]  # diffusers absent: our restated schedule (parity unpinned)
    noisy = acp.sqrt()[:, None, None] * action + (1 - acp).sqrt()[:, None, None] * noise
    noisy[mask] = action[mask]
    pred = unet(noisy, timesteps, local_cond=None, global_cond=global_cond)
    loss = torch.nn.functional.mse_loss(pred, noise, reduction="none") * (~mask).float()
    loss = loss.reshape(3, -1).mean(1).mean()
    loss.backward()
    fx = {"noise": noise.numpy(), "timesteps": timesteps.numpy(), "out.loss": loss.detach().numpy(),
          "out.pred": pred.detach().numpy(), "out.global_cond": global_cond.detach().numpy(), "out.mask": mask.numpy()}
    for k, v in batch["obs"]["pcds"].items():
        fx[f"in.pcds.{k}"] = v.numpy()
    fx["in.qpos"], fx["in.action"] = qpos.numpy(), action.numpy()
    for k, v in sd.items():
        fx[f"w.{k}"] = v.numpy()
    g_enc = dict(enc.named_parameters())
    g_unet = dict(unet.named_parameters())
    for k in ("linear.weight", "projector.0.weight", "projector.4.weight", "key_model_map.pcd.conv1.0.weight", "key_model_map.pcd.final.weight"):
        fx[f"grad.obs_encoder.{k}"] = g_enc[k].grad.numpy()
    for k in ("diffusion_step_encoder.1.weight", "down_modules.0.0.blocks.0.block.0.weight", "mid_modules.1.cond_encoder.1.weight",
              "up_modules.1.2.conv.weight", "final_conv.1.weight", "down_modules.1.2.conv.weight", "up_modules.0.0.residual_conv.weight"):
        fx[f"grad.model.{k}"] = g_unet[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, "dp_pcd_small.npz"), **fx)
    print("dp_pcd_small.npz: loss", float(loss))


def golden_dp_rlbench(ref):
    """The RLBench Diffusion-Policy composition: as golden_dp, with the 512-d language goal appended to the global condition
    (diffusion_unet_image_policy.py:58-62,262-266; configs/model/rlbench_diffusion_policy_model.yaml:26-28) and the 11-d
    action / proprioception of configs/data/rlbench_diffusion_policy_pcd_dataset.yaml:17."""
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc import build_dp_policy, make_dp_batch
    from pointcloudmatters_amd.policy import PointNet

    pcd_npoints, ad, gd = 32, 11, 512
    torch.manual_seed(987)
    ours = build_dp_policy(pcd_npoints=pcd_npoints, pointops=pointops_cpu, sa_impl="reference", action_dim=ad, qpos_dim=ad, goal_dim=gd,
                           **DP_SMALL)
    sd = ours.state_dict()
    shape_meta = {"obs": {"pcds": {"shape": [6], "type": "pcd"}, "qpos": {"shape": [ad], "type": "low_dim"}},
                  "action": {"shape": [ad]}, "goal": {"task_emb": {"shape": [gd]}}}
    enc = ref.pcd_enc.PCDObsEncoder(shape_meta=shape_meta, pcd_model=PointNet(in_channels=6, num_classes=24),
                                    share_pcd_model=True, n_obs_step=2, pcd_nsample=16, pcd_npoints=pcd_npoints,
                                    pcd_hidden_dim=24, projector_layers=1, projector_channels=[24, 40, 40])
    enc.load_state_dict({k[len("obs_encoder."):]: v for k, v in sd.items() if k.startswith("obs_encoder.")}, strict=True)
    unet = ref.unet.ConditionalUnet1D(input_dim=ad, local_cond_dim=None, global_cond_dim=(40 + ad) * 2 + gd,
                                      diffusion_step_embed_dim=16, down_dims=[16, 32, 64], kernel_size=5, n_groups=8,
                                      cond_predict_scale=True)
    unet.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}, strict=True)
    mg = ref.maskgen.LowdimMaskGenerator(action_dim=ad, obs_dim=0, max_n_obs_steps=2, fix_obs_steps=True, action_visible=False)
    enc.train(), unet.train()
    batch = make_dp_batch(3, 150, seed=13, ragged=True, action_dim=ad, qpos_dim=ad, goal_dim=gd)
    g = torch.Generator().manual_seed(18)
    noise = torch.randn(3, 16, ad, generator=g)
    timesteps = torch.tensor([0, 41, 99])
    qpos, action, task_emb = batch["obs"]["qpos"], batch["action"], batch["goal"]["task_emb"]
    this_nobs = {"qpos": qpos[:, :2].reshape(-1, ad), "pcds": {k: v.clone() for k, v in batch["obs"]["pcds"].items()}}
    global_cond = torch.cat([enc(this_nobs).reshape(3, -1), task_emb], dim=-1)  # :262-266
    mask = mg((3, 16, ad))
    acp = ours.noise_scheduler.alphas_cumprod[timesteps]  # diffusers absent: our restated schedule (parity unpinned)
    noisy = acp.sqrt()[:, None, None] * action + (1 - acp).sqrt()[:, None, None] * noise
    noisy[mask] = action[mask]
    pred = unet(noisy, timesteps, local_cond=None, global_cond=global_cond)
    loss = torch.nn.functional.mse_loss(pred, noise, reduction="none") * (~mask).float()
    loss = loss.reshape(3, -1).mean(1).mean()
    loss.backward()
    fx = {"noise": noise.numpy(), "timesteps": timesteps.numpy(), "out.loss": loss.detach().numpy(), "out.pred": pred.detach().numpy(),
          "out.global_cond": global_cond.detach().numpy(), "in.qpos": qpos.numpy(), "in.action": action.numpy(),
          "in.task_emb": task_emb.numpy()}
    for k, v in batch["obs"]["pcds"].items():
        fx[f"in.pcds.{k}"] = v.numpy()
    for k, v in sd.items():
        fx[f"w.{k}"] = v.numpy()
    g_enc, g_unet = dict(enc.named_parameters()), dict(unet.named_parameters())
    for k in ("linear.weight", "projector.4.weight", "key_model_map.pcd.conv5.0.weight"):
        fx[f"grad.obs_encoder.{k}"] = g_enc[k].grad.numpy()
    for k in ("down_modules.0.0.cond_encoder.1.weight", "mid_modules.0.cond_encoder.1.weight", "final_conv.1.weight",
              "up_modules.0.0.blocks.0.block.0.weight"):
        fx[f"grad.model.{k}"] = g_unet[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, "dp_rlbench_small.npz"), **fx)
    print("dp_rlbench_small.npz: loss", float(loss), "global_cond", tuple(global_cond.shape))



def fg_mask(coord, offset, seed, frac=0.6):
    """A seeded per-point foreground mask (~`frac` of every cloud), like the datasets' `pcd["mask"]`
    (maniskill2_single_task_pcd_act.py:229, rlbench_single_task_act.py:307)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(coord.shape[0], generator=g) < frac


def golden_mask(ref):
    """The `use_mask` / `bg_ratio` branch of pcd_sampling (act.py:394-442, pcd_obs_encoder.py:131-180): reference ACTPCD and
    PCDObsEncoder with foreground/background split FPS.  Weights and inputs are those of act_pcd_small.npz / dp_pcd_small.npz
    (same seeds); only the mask, the sampled indices and the outputs are stored."""
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc import build_dp_policy, make_act_batch, make_dp_batch
    from pointcloudmatters_amd.policy import PointNet

    fx = {}
    pcd_npoints = 32
    ours = build_ours(pcd_npoints, seed=1234)
    old = np.load(os.path.join(OUT, "act_pcd_small.npz"))
    assert all(np.array_equal(old[f"w.{k}"], v.numpy()) for k, v in ours.state_dict().items())
    batch = make_act_batch(3, 180, seed=77, ragged=True, num_queries=SMALL["num_queries"])
    assert np.array_equal(old["in.pcds.coord"], batch["pcds"]["coord"].numpy())
    eps = torch.from_numpy(old["eps"])
    mask = fg_mask(batch["pcds"]["coord"], batch["pcds"]["offset"], seed=int(os.environ.get("PCM_MASK_SEED", 21)))
    fx["act.mask"] = mask.numpy()
    for tag, bg in (("bg25", 0.25), ("bg0", 0.0)):
        model = build_reference_actpcd(ref, ours, pcd_npoints)
        model.use_mask, model.bg_ratio = True, bg
        model.train()
        orig = ref.act.reparametrize
        ref.act.reparametrize = lambda mu, logvar: mu + logvar.div(2).exp() * eps
        try:
            dd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in batch.items()}
            dd["pcds"]["offset"] = dd["pcds"]["offset"].clone()
            dd["pcds"]["mask"] = mask
            with torch.no_grad():
                p, o = dd["pcds"]["coord"], dd["pcds"]["offset"]
                idx = model.pcd_sampling((p, torch.zeros(p.shape[0], model.backbone.num_channels), o), mask, return_index=True)[3]
            model.bn.running_mean.zero_(), model.bn.running_var.fill_(1.0), model.bn.num_batches_tracked.zero_()
            out = model(dd)
            out["loss"].backward()
        finally:
            ref.act.reparametrize = orig
        fx[f"act.{tag}.idx"] = idx.numpy()
        for k in ("a_hat", "loss", "src", "pos"):
            fx[f"act.{tag}.out.{k}"] = out[k].detach().numpy()
        grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
        for k in ("linear.weight", "backbone.conv1.0.weight", "transformer.encoder.layers.0.self_attn.in_proj_weight"):
            fx[f"act.{tag}.grad.{k}"] = grads[k].numpy()
        print(f"mask_ref.npz act {tag}: loss", float(out["loss"]), "idx[:6]", idx[:6].tolist())
    # ---- Diffusion-Policy observation encoder
    torch.manual_seed(4321)
    dp = build_dp_policy(pcd_npoints=pcd_npoints, pointops=pointops_cpu, sa_impl="reference", **DP_SMALL)
    sd = dp.state_dict()
    oldd = np.load(os.path.join(OUT, "dp_pcd_small.npz"))
    assert all(np.array_equal(oldd[f"w.{k}"], v.numpy()) for k, v in sd.items())
    shape_meta = {"obs": {"pcds": {"shape": [6], "type": "pcd"}, "qpos": {"shape": [9], "type": "low_dim"}},
                  "action": {"shape": [7]}}
    enc = ref.pcd_enc.PCDObsEncoder(shape_meta=shape_meta, pcd_model=PointNet(in_channels=6, num_classes=24),
                                    share_pcd_model=True, n_obs_step=2, pcd_nsample=16, pcd_npoints=pcd_npoints,
                                    pcd_hidden_dim=24, projector_layers=1, projector_channels=[24, 40, 40],
                                    use_mask=True, bg_ratio=0.25)
    enc.load_state_dict({k[len("obs_encoder."):]: v for k, v in sd.items() if k.startswith("obs_encoder.")}, strict=True)
    enc.train()
    dbatch = make_dp_batch(3, 150, seed=11, ragged=True)
    pc = {k: v.clone() for k, v in dbatch["obs"]["pcds"].items()}
    dmask = fg_mask(pc["coord"], pc["offset"], seed=22)
    pc["mask"] = dmask
    feat = enc({"qpos": dbatch["obs"]["qpos"][:, :2].reshape(-1, 9), "pcds": pc})
    (feat * torch.sin(torch.arange(feat.numel(), device=feat.device).float()).view_as(feat)).sum().backward()
    fx["dp.mask"] = dmask.numpy()
    fx["dp.bg25.feat"] = feat.detach().numpy()
    fx["dp.bg25.grad.linear.weight"] = enc.linear.weight.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "mask_ref.npz"), **fx)
    print("mask_ref.npz: dp feat", tuple(feat.shape))


def _ref_act(ref, c, pcd_npoints, backbone, cls=None, **extra):
    """The reference ACTPCD (or a subclass) for config dict `c` (keys of SMALL) around `backbone`."""
    transformer = ref.transformer.Transformer(
        d_model=c["hidden_dim"], dropout=c["dropout"], nhead=c["nhead"], dim_feedforward=c["dim_feedforward"],
        num_encoder_layers=c["num_encoder_layers"], num_decoder_layers=c["num_decoder_layers"], normalize_before=False,
        return_intermediate_dec=True)
    encoder = ref.transformer.TransformerEncoder(
        d_model=c["hidden_dim"], dropout=c["dropout"], nhead=c["nhead"], dim_feedforward=c["dim_feedforward"],
        num_layers=c["num_encoder_layers"], normalize_before=False, activation="relu")
    return (cls or ref.act.ACTPCD)(
        backbone=backbone, transformer=transformer, encoder=encoder, hidden_dim=c["hidden_dim"],
        num_queries=c["num_queries"], num_cameras=1, action_dim=c["action_dim"], qpos_dim=c["qpos_dim"], env_state_dim=0,
        latent_dim=c["latent_dim"], action_loss=torch.nn.MSELoss(reduction="none"), klloss=ref.loss.KLDivergence(),
        kl_weight=c["kl_weight"], goal_cond_dim=c["goal_cond_dim"], pcd_nsample=c["pcd_nsample"], pcd_npoints=pcd_npoints, **extra)


def _run_ref_act(ref, model, batch, eps):
    orig = ref.act.reparametrize
    ref.act.reparametrize = lambda mu, logvar: mu + logvar.div(2).exp() * eps
    try:
        dd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in batch.items()}
        dd["pcds"] = {k: v.clone() for k, v in dd["pcds"].items()}
        out = model(dd)
        out["loss"].backward()
    finally:
        ref.act.reparametrize = orig
    return out


def _sa_margin(layer, pcds, mask=None):
    """Smallest distance of the set-abstraction layer from a point where its GRADIENT jumps: |max_k BN(y)| (the ReLU of a
    group's maximum switching) and the gap between the two largest entries of a live group (the arg-max switching)."""
    from oracle import pointops_cpu

    saved = {k: v.clone() for k, v in layer.bn.state_dict().items()}
    with torch.no_grad():
        p, o = pcds["coord"], pcds["offset"]
        n_p, _, n_o, idx = layer.pcd_sampling((p, pcds["feat"], o), mask, return_index=True)[:4]
        g, _ = pointops_cpu.knn_query_and_group(pcds["feat"], p, offset=o, new_xyz=n_p, new_offset=n_o, nsample=layer.pcd_nsample, with_xyz=True)
        y = layer.linear(g)
        z = torch.nn.functional.batch_norm(y.reshape(-1, y.shape[-1]), None, None, layer.bn.weight, layer.bn.bias, True).view(y.shape)
        top = z.topk(2, dim=1).values
        live = top[:, 0] > 0
    layer.bn.load_state_dict(saved)  # pcd_sampling ran in training mode: the probe must not count as a step
    return float(min(top[:, 0].abs().min(), (top[:, 0] - top[:, 1])[live].min()))


def _kink_margin(run, modules, layer, pcds, mask=None):
    """Distance of a whole tokenizer from its gradient's jumps: `_sa_margin` of the set-abstraction layer and the smallest
    |pre-activation| in front of every ReLU of the point-cloud model (`modules` = its BatchNorm layers; `run()` = one
    forward pass).  One ReLU flipping among the ~80 000 of a 96-point fixture moves a weight gradient by ~1e-3 while every
    output stays within 1e-6 -- observed: the SAME CPU code on 1 vs 8 threads.  A fixture that close to a kink cannot be held
    to 1e-4 by ANY re-associated evaluation, so the batch seeds are chosen with all margins >= `3e-5` (rounding noise of
    these layers is ~5e-6); buffers are restored afterwards."""
    import copy

    mins, hooks = [], []
    owners = [m for m in modules]
    saved = [copy.deepcopy(m.state_dict()) for m in owners]
    for m in owners:
        hooks.append(m.register_forward_hook(lambda mod, i, o: mins.append(float(o.detach().abs().min()))))
    sa = _sa_margin(layer, pcds, mask)
    sa_saved = copy.deepcopy(layer.bn.state_dict())
    try:
        with torch.no_grad():
            run()
    finally:
        for h in hooks:
            h.remove()
        for m, sd in zip(owners, saved):
            m.load_state_dict(sd)
        layer.bn.load_state_dict(sa_saved)
    return min([sa] + mins)


PRESAMPLE_SEED = 20240


def golden_presample(ref):
    """`pre_sample=True` (act.py:366-376,509-530; pcd_obs_encoder.py:81-120,200-218): the set-abstraction layer on the RAW
    features in front of the backbone -- what configs/exp_maniskill2_{act,diffusion}_policy/maniskill2_model/
    scratch_pointnet_pcd_presample{,_wo_rgb,_wo_xyz}.yaml select.  Reference ACTPCD with feature widths 6 ([color, coord]) and 3
    (`_wo_rgb`: coord only), with and without `use_mask`; reference PCDObsEncoder (widths 6 and 3: `_wo_xyz`, colour only) +
    ConditionalUnet1D composed as compute_loss does.  Weights are NOT stored: both sides fill them with tests/util.seeded_fill
    (the stored checksum pins the fill); inputs, outputs and gradients are."""
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc import build_act_policy, build_dp_policy, make_act_batch, make_dp_batch
    from pointcloudmatters_amd.policy import PointNet
    from tests.util import seeded_fill

    fx = {}
    pcd_npoints = 32
    c = SMALL
    eps = torch.randn(3, c["latent_dim"], generator=torch.Generator().manual_seed(15))
    fx["act.eps"] = eps.numpy()
    for tag, cin, keys, use_mask in (("c6", 6, ("color", "coord"), False), ("c3", 3, ("coord",), False), ("c6mask", 6, ("color", "coord"), True)):
        backbone = PointNet(in_channels=cin, num_classes=c["hidden_dim"])
        extra = dict(pre_sample=True, in_channels=cin)
        if use_mask:
            extra.update(use_mask=True, bg_ratio=0.25)
        model = _ref_act(ref, c, pcd_npoints, backbone, **extra)
        assert tuple(model.linear.weight.shape) == (cin, 3 + cin) and model.bn.num_features == cin
        fx[f"act.{tag}.wsum"] = np.array(seeded_fill(model, PRESAMPLE_SEED))
        for seed in range(91 + cin, 200):  # first batch seed whose set-abstraction layer is away from its kinks
            batch = make_act_batch(3, 170, seed=seed, ragged=True, num_queries=c["num_queries"], feat_keys=keys)
            if use_mask:
                batch["pcds"]["mask"] = fg_mask(batch["pcds"]["coord"], batch["pcds"]["offset"], seed=33)
            def run(b=batch):
                dd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}
                model.forward_pcd_embed({k: v.clone() for k, v in dd["pcds"].items()})

            margin = _kink_margin(run, [blk[1] for blk in (backbone.conv1, backbone.conv2, backbone.conv3, backbone.conv4, backbone.conv5)],
                                  model, batch["pcds"], batch["pcds"].get("mask"))
            if margin >= 3e-5:
                break
        print(f"  act {tag}: batch seed {seed}, margin {margin:.2e}")
        # the product's class takes the same fill: same parameter names (strict load both ways)
        ours = build_act_policy(pcd_npoints=pcd_npoints, pointops=pointops_cpu, sa_impl="reference", pre_sample=True, in_channels=cin,
                                backbone_num_classes=c["hidden_dim"], **({"use_mask": True, "bg_ratio": 0.25} if use_mask else {}), **SMALL)
        ours.load_state_dict(model.state_dict(), strict=True)
        model.load_state_dict(ours.state_dict(), strict=True)
        model.train()
        out = _run_ref_act(ref, model, batch, eps)
        for k, v in batch.items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    fx[f"act.{tag}.in.pcds.{kk}"] = vv.numpy()
            else:
                fx[f"act.{tag}.in.{k}"] = v.numpy()
        for k in ("a_hat", "is_pad_hat", "mu", "logvar", "loss", "action_loss", "kl_loss", "src", "pos"):
            fx[f"act.{tag}.out.{k}"] = out[k].detach().numpy()
        grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
        keep = grads if tag == "c6" else ("linear.weight", "bn.weight", "bn.bias", "backbone.conv1.0.weight", "backbone.final.weight",
                                         "transformer.encoder.layers.0.self_attn.in_proj_weight", "action_head.weight")
        for k in keep:
            fx[f"act.{tag}.grad.{k}"] = grads[k].numpy()
        fx[f"act.{tag}.grad_none"] = np.array(sorted(n for n, p in model.named_parameters() if p.grad is None))
        fx[f"act.{tag}.bn_running_mean"], fx[f"act.{tag}.bn_running_var"] = model.bn.running_mean.numpy().copy(), model.bn.running_var.numpy().copy()
        with torch.no_grad():  # the sampled indices, as pcd_sampling returns them
            p, o = batch["pcds"]["coord"], batch["pcds"]["offset"]
            fx[f"act.{tag}.idx"] = model.pcd_sampling((p, batch["pcds"]["feat"], o), batch["pcds"].get("mask"), return_index=True)[3].numpy()
        print(f"presample_ref.npz act {tag}: loss", float(out["loss"]), "src", tuple(out["src"].shape))

    # ---- Diffusion Policy: encoder widths 6 / 3, point-cloud model width (20) != pcd_hidden_dim (24) so that the projector's
    # first convolution (pcd_obs_encoder.py:103-112) is pinned to `pcd_model.num_channels`
    for tag, cin, keys in (("c6", 6, ("color", "coord")), ("c3", 3, ("color",))):
        small = dict(DP_SMALL, pcd_num_classes=20)
        shape_meta = {"obs": {"pcds": {"shape": [cin], "type": "pcd"}, "qpos": {"shape": [9], "type": "low_dim"}}, "action": {"shape": [7]}}
        enc = ref.pcd_enc.PCDObsEncoder(shape_meta=shape_meta, pcd_model=PointNet(in_channels=cin, num_classes=20), share_pcd_model=True,
                                        n_obs_step=2, pcd_nsample=16, pcd_npoints=pcd_npoints, pcd_hidden_dim=24, projector_layers=1,
                                        projector_channels=[24, 40, 40], pre_sample=True, in_channel=cin)
        assert tuple(enc.linear.weight.shape) == (cin, 3 + cin) and enc.projector[0].in_channels == 20
        unet = ref.unet.ConditionalUnet1D(input_dim=7, local_cond_dim=None, global_cond_dim=(40 + 9) * 2, diffusion_step_embed_dim=16,
                                          down_dims=[16, 32, 64], kernel_size=5, n_groups=8, cond_predict_scale=True)
        ours = build_dp_policy(pcd_npoints=pcd_npoints, pointops=pointops_cpu, sa_impl="reference", pre_sample=True, in_channels=cin, **small)
        fx[f"dp.{tag}.wsum"] = np.array(seeded_fill(ours, PRESAMPLE_SEED + 1))
        sd = ours.state_dict()
        enc.load_state_dict({k[len("obs_encoder."):]: v for k, v in sd.items() if k.startswith("obs_encoder.")}, strict=True)
        unet.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}, strict=True)
        mg = ref.maskgen.LowdimMaskGenerator(action_dim=7, obs_dim=0, max_n_obs_steps=2, fix_obs_steps=True, action_visible=False)
        enc.train(), unet.train()
        for seed in range(17 + cin, 200):
            dbatch = make_dp_batch(3, 150, seed=seed, ragged=True, feat_keys=keys)
            pn = enc.key_model_map["pcd"]
            margin = _kink_margin(lambda b=dbatch: enc.encode_pcd(pn, {k: v.clone() for k, v in b["obs"]["pcds"].items()}),
                                  [blk[1] for blk in (pn.conv1, pn.conv2, pn.conv3, pn.conv4, pn.conv5)] + [enc.projector[1]],
                                  enc, dbatch["obs"]["pcds"])
            if margin >= 3e-5:
                break
        print(f"  dp {tag}: batch seed {seed}, margin {margin:.2e}")
        noise = torch.randn(3, 16, 7, generator=torch.Generator().manual_seed(28))
        timesteps = torch.tensor([7, 50, 93])
        qpos, action = dbatch["obs"]["qpos"], dbatch["action"]
        this_nobs = {"qpos": qpos[:, :2].reshape(-1, 9), "pcds": {k: v.clone() for k, v in dbatch["obs"]["pcds"].items()}}
        feat = enc(this_nobs)
        global_cond = feat.reshape(3, -1)
        mask = mg((3, 16, 7))
        acp = ours.noise_scheduler.alphas_cumprod[timesteps]  # diffusers absent: our restated schedule (parity unpinned)
        noisy = acp.sqrt()[:, None, None] * action + (1 - acp).sqrt()[:, None, None] * noise
        noisy[mask] = action[mask]
        pred = unet(noisy, timesteps, local_cond=None, global_cond=global_cond)
        loss = torch.nn.functional.mse_loss(pred, noise, reduction="none") * (~mask).float()
        loss = loss.reshape(3, -1).mean(1).mean()
        loss.backward()
        fx[f"dp.{tag}.noise"], fx[f"dp.{tag}.timesteps"] = noise.numpy(), timesteps.numpy()
        fx[f"dp.{tag}.out.loss"], fx[f"dp.{tag}.out.pred"] = loss.detach().numpy(), pred.detach().numpy()
        fx[f"dp.{tag}.out.global_cond"] = global_cond.detach().numpy()
        for k, v in dbatch["obs"]["pcds"].items():
            fx[f"dp.{tag}.in.pcds.{k}"] = v.numpy()
        fx[f"dp.{tag}.in.qpos"], fx[f"dp.{tag}.in.action"] = qpos.numpy(), action.numpy()
        for k, p in enc.named_parameters():
            if p.grad is not None:
                fx[f"dp.{tag}.grad.obs_encoder.{k}"] = p.grad.numpy()
        g_unet = dict(unet.named_parameters())
        for k in ("down_modules.0.0.cond_encoder.1.weight", "mid_modules.1.cond_encoder.1.weight", "final_conv.1.weight"):
            fx[f"dp.{tag}.grad.model.{k}"] = g_unet[k].grad.numpy()
        print(f"presample_ref.npz dp {tag}: loss", float(loss))
    np.savez_compressed(os.path.join(OUT, "presample_ref.npz"), **fx)


WIDE = dict(hidden_dim=512, nhead=8, dim_feedforward=32, num_encoder_layers=1, num_decoder_layers=2, dropout=0.0, latent_dim=32,
            num_queries=100, kl_weight=10.0, action_dim=7, qpos_dim=9, goal_cond_dim=3, pcd_nsample=16)
WIDE_SEED = 51200
WIDE_DP = dict(down_dims=(128, 256), diffusion_step_embed_dim=128, pcd_num_classes=96, pcd_hidden_dim=96, projector_channels=(96, 128, 128),
               n_groups=8)


def _store_grads(fx, prefix, named_grads):
    from tests.util import grad_digest

    for name, g in named_grads:
        for part, v in grad_digest(name, g.numpy()).items():
            fx[f"{prefix}{name}/{part}"] = v


def golden_wide(ref):
    """The SHIPPED WIDTHS (configs/model/maniskill2_act_pcd_model.yaml:49-68: d = 512, 8 heads, feed-forward 32, 100 queries,
    latent 32; Diffusion Policy encoder: PointNet head 96, SA 96, projector [96, 128, 128]) through the reference's own
    ACTPCD / Transformer / PCDObsEncoder / ConditionalUnet1D, so that the fused HIP kernels -- which engage only at these
    widths (E % 256 == 0, head_dim 64, feed-forward 512 / 32) -- are compared with the REFERENCE and not with a restatement.
    1 encoder + 2 decoder layers (the second decoder layer is a dead layer, act.py:270), 2 ragged clouds of ~150 points;
    variant "flash": 128 tokens per cloud -> 131-token sequences (csrc/attn_flash.hip takes query sets > 128), variant
    "small": 96 -> 99 tokens (csrc/attn_small.hip).  Weights: tests/util.seeded_fill on both sides (checksum stored); stored:
    inputs, outputs and a digest of EVERY gradient (tests/util.grad_digest).  Batch seeds are searched so that no ReLU
    pre-activation / arg-max gap of the whole model sits at the rounding level (see _kink_margin)."""
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc import build_act_policy, build_dp_policy, make_act_batch, make_dp_batch
    from pointcloudmatters_amd.policy import PointNet
    from tests.util import seeded_fill

    fx = {}
    c = WIDE
    eps = torch.randn(2, c["latent_dim"], generator=torch.Generator().manual_seed(25))
    fx["act.eps"] = eps.numpy()
    for tag, M in (("flash", 128), ("small", 96)):
        backbone = PointNet(in_channels=6, num_classes=0)
        model = _ref_act(ref, c, M, backbone)
        fx[f"act.{tag}.wsum"] = np.array(seeded_fill(model, WIDE_SEED))
        ours = build_act_policy(pcd_npoints=M, pointops=pointops_cpu, sa_impl="reference", **c)
        ours.load_state_dict(model.state_dict(), strict=True)  # same names, same shapes
        model.train()
        relus = [blk[1] for blk in (backbone.conv1, backbone.conv2, backbone.conv3, backbone.conv4, backbone.conv5)]
        relus += [m.linear1 for m in model.modules() if hasattr(m, "linear1") and hasattr(m, "linear2")]
        best = None
        for seed in range(300, 1300):
            batch = make_act_batch(2, 150, seed=seed, ragged=True, num_queries=c["num_queries"])

            def run(b=batch):
                orig = ref.act.reparametrize
                ref.act.reparametrize = lambda mu, logvar: mu + logvar.div(2).exp() * eps
                try:
                    dd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}
                    dd["pcds"] = {k: v.clone() for k, v in dd["pcds"].items()}
                    model(dd)
                finally:
                    ref.act.reparametrize = orig

            # the SA layer's input is the backbone's output here: compute its margin on those features
            with torch.no_grad():
                saved = [{k: v.clone() for k, v in m.state_dict().items()} for m in relus[:5]]
                feats = backbone({k: v.clone() for k, v in batch["pcds"].items()})
                for m, sd in zip(relus[:5], saved):
                    m.load_state_dict(sd)
            sa = _sa_margin(model, dict(batch["pcds"], feat=feats))
            if sa < 1e-5:
                continue
            margin = _kink_margin(run, relus, model, dict(batch["pcds"], feat=feats))
            if best is None or margin > best[0]:
                best = (margin, seed)
            if margin >= 4e-6:
                break
        margin, seed = best
        batch = make_act_batch(2, 150, seed=seed, ragged=True, num_queries=c["num_queries"])
        print(f"  act {tag}: batch seed {seed}, margin {margin:.2e}")
        assert margin >= 2e-6
        out = _run_ref_act(ref, model, batch, eps)
        for k, v in batch.items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    fx[f"act.{tag}.in.pcds.{kk}"] = vv.numpy()
            else:
                fx[f"act.{tag}.in.{k}"] = v.numpy()
        for k in ("a_hat", "is_pad_hat", "mu", "logvar", "loss", "action_loss", "kl_loss", "src", "pos"):
            fx[f"act.{tag}.out.{k}"] = out[k].detach().numpy()
        _store_grads(fx, f"act.{tag}.grad.", [(n, p.grad) for n, p in model.named_parameters() if p.grad is not None])
        fx[f"act.{tag}.grad_none"] = np.array(sorted(n for n, p in model.named_parameters() if p.grad is None))
        fx[f"act.{tag}.bn_running_mean"], fx[f"act.{tag}.bn_running_var"] = model.bn.running_mean.numpy().copy(), model.bn.running_var.numpy().copy()
        print(f"wide_ref.npz act {tag}: loss", float(out["loss"].detach()), "src", tuple(out["src"].shape))

    # ---- Diffusion Policy at the shipped encoder widths
    M = 64
    shape_meta = {"obs": {"pcds": {"shape": [6], "type": "pcd"}, "qpos": {"shape": [9], "type": "low_dim"}}, "action": {"shape": [7]}}
    enc = ref.pcd_enc.PCDObsEncoder(shape_meta=shape_meta, pcd_model=PointNet(in_channels=6, num_classes=96), share_pcd_model=True,
                                    n_obs_step=2, pcd_nsample=16, pcd_npoints=M, pcd_hidden_dim=96, projector_layers=1,
                                    projector_channels=[96, 128, 128])
    unet = ref.unet.ConditionalUnet1D(input_dim=7, local_cond_dim=None, global_cond_dim=(128 + 9) * 2, diffusion_step_embed_dim=128,
                                      down_dims=[128, 256], kernel_size=5, n_groups=8, cond_predict_scale=True)
    ours = build_dp_policy(pcd_npoints=M, pointops=pointops_cpu, sa_impl="reference", **WIDE_DP)
    fx["dp.wsum"] = np.array(seeded_fill(ours, WIDE_SEED + 1))
    sd = ours.state_dict()
    enc.load_state_dict({k[len("obs_encoder."):]: v for k, v in sd.items() if k.startswith("obs_encoder.")}, strict=True)
    unet.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}, strict=True)
    mg = ref.maskgen.LowdimMaskGenerator(action_dim=7, obs_dim=0, max_n_obs_steps=2, fix_obs_steps=True, action_visible=False)
    enc.train(), unet.train()
    pn = enc.key_model_map["pcd"]
    best = None
    for seed in range(500, 900):
        dbatch = make_dp_batch(2, 100, seed=seed, ragged=True)
        with torch.no_grad():
            saved = {k: v.clone() for k, v in pn.state_dict().items()}
            feats = pn({k: v.clone() for k, v in dbatch["obs"]["pcds"].items()})
            pn.load_state_dict(saved)
        sa = _sa_margin(enc, dict(dbatch["obs"]["pcds"], feat=feats))
        if sa < 1e-5:
            continue
        margin = _kink_margin(lambda b=dbatch: enc.encode_pcd(pn, {k: v.clone() for k, v in b["obs"]["pcds"].items()}),
                              [blk[1] for blk in (pn.conv1, pn.conv2, pn.conv3, pn.conv4, pn.conv5)] + [enc.projector[1]],
                              enc, dict(dbatch["obs"]["pcds"], feat=feats))
        if best is None or margin > best[0]:
            best = (margin, seed)
        if margin >= 4e-6:
            break
    margin, seed = best
    dbatch = make_dp_batch(2, 100, seed=seed, ragged=True)
    print(f"  dp: batch seed {seed}, margin {margin:.2e}")
    noise = torch.randn(2, 16, 7, generator=torch.Generator().manual_seed(38))
    timesteps = torch.tensor([12, 88])
    qpos, action = dbatch["obs"]["qpos"], dbatch["action"]
    this_nobs = {"qpos": qpos[:, :2].reshape(-1, 9), "pcds": {k: v.clone() for k, v in dbatch["obs"]["pcds"].items()}}
    global_cond = enc(this_nobs).reshape(2, -1)
    mask = mg((2, 16, 7))
    acp = ours.noise_scheduler.alphas_cumprod[timesteps]  # diffusers absent: our restated schedule (parity unpinned)
    noisy = acp.sqrt()[:, None, None] * action + (1 - acp).sqrt()[:, None, None] * noise
    noisy[mask] = action[mask]
    pred = unet(noisy, timesteps, local_cond=None, global_cond=global_cond)
    loss = torch.nn.functional.mse_loss(pred, noise, reduction="none") * (~mask).float()
    loss = loss.reshape(2, -1).mean(1).mean()
    loss.backward()
    fx["dp.noise"], fx["dp.timesteps"] = noise.numpy(), timesteps.numpy()
    fx["dp.out.loss"], fx["dp.out.pred"], fx["dp.out.global_cond"] = loss.detach().numpy(), pred.detach().numpy(), global_cond.detach().numpy()
    for k, v in dbatch["obs"]["pcds"].items():
        fx[f"dp.in.pcds.{k}"] = v.numpy()
    fx["dp.in.qpos"], fx["dp.in.action"] = qpos.numpy(), action.numpy()
    _store_grads(fx, "dp.grad.", [("obs_encoder." + n, p.grad) for n, p in enc.named_parameters() if p.grad is not None]
                 + [("model." + n, p.grad) for n, p in unet.named_parameters() if p.grad is not None])
    print("wide_ref.npz dp: loss", float(loss))
    np.savez_compressed(os.path.join(OUT, "wide_ref.npz"), **fx)


WIDE_BF16_B = 8  # samples of the bf16 fixture (the flip-prone layers see 8x more rows than in wide_ref.npz)


def _fp32_island(fn):
    """`fn` with autocast switched off inside: how the REFERENCE's tokenizer is kept in fp32 for the bf16 yardstick run (the
    product does the same through policy/precision.py)."""
    def inner(*a, **k):
        with torch.autocast("cpu", enabled=False):
            return fn(*a, **k)
    return inner


def _digest_errors(named_got, fx, prefix):
    from tests.util import digest_rel_error

    out = {}
    for name, g in named_got:
        ref = {k[len(prefix) + len(name) + 1:]: fx[k] for k in fx if k.startswith(prefix + name + "/")}
        out[name] = digest_rel_error(name, g.numpy(), ref)
    return out


def golden_wide_bf16(ref):
    """The fixture the TIMED configuration (bf16 autocast, fused kernels) is held to -- tests/test_bf16_fixture.py.

    wide_ref.npz pins the fused kernels in fp32 at 1e-4; its two-sample batches cannot pin bf16: one flipped ReLU gate of the CVAE
    encoder's CLS row owns 1/64 of a gradient matrix there (44 % measured on hardware, profiles/r04_wide_bf16_errors.log).  Here:
      * the same reference classes at the shipped widths (WIDE / WIDE_DP), WIDE_BF16_B samples;
      * stored: inputs, the fp32 reference outputs and a digest of EVERY fp32 gradient (the truth), and per tensor the error of
        the reference's OWN bf16 evaluation against that truth ("yard."): torch.autocast(bf16) around the reference model with its
        tokenizer (backbone + pcd_sampling / encode_pcd) kept in fp32 -- the precision recipe of policy/precision.py;
      * the batch seed is searched so that this yardstick run AND a second one on inputs jittered by 1e-3 (another rounding
        realisation) stay below 4 % on every tensor: no gate of the batch sits within a bf16 rounding of its kink, so a 10 % bound on
        the product's bf16 path is a statement about its arithmetic and not about a coin flip.
    Also stored: the yardstick with the tokenizer under autocast ("yard_all."), the evidence for that recipe."""
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc import build_dp_policy, make_act_batch, make_dp_batch
    from pointcloudmatters_amd.policy import PointNet
    from tests.util import seeded_fill

    fx, c, B, M = {}, WIDE, WIDE_BF16_B, 128
    eps = torch.randn(B, c["latent_dim"], generator=torch.Generator().manual_seed(26))
    fx["act.eps"] = eps.numpy()

    def act_model(fp32_tokenizer):
        backbone = PointNet(in_channels=6, num_classes=0)
        model = _ref_act(ref, c, M, backbone)
        wsum = seeded_fill(model, WIDE_SEED)
        model.train()
        if fp32_tokenizer:
            backbone.forward, model.pcd_sampling = _fp32_island(backbone.forward), _fp32_island(model.pcd_sampling)
        return model, wsum

    def act_grads(batch, bf16, fp32_tokenizer=True):
        model, wsum = act_model(fp32_tokenizer)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=bf16):
            out = _run_ref_act(ref, model, batch, eps)
        return model, wsum, out, [(n, p.grad.detach().float()) for n, p in model.named_parameters() if p.grad is not None]

    def jitter(batch, rel=1e-3):
        g = torch.Generator().manual_seed(5)
        out = {k: (dict(v) if isinstance(v, dict) else v) for k, v in batch.items()}
        for k in ("qpos", "actions", "goal_cond"):
            out[k] = batch[k] * (1 + rel * torch.randn(batch[k].shape, generator=g))
        return out

    best = None
    for seed in range(2300, 2340):
        batch = make_act_batch(B, 150, seed=seed, ragged=True, num_queries=c["num_queries"])
        _, _, _, g32 = act_grads(batch, False)
        tmp = {}
        _store_grads(tmp, "t.", g32)
        worst = 0.0
        for b2 in (batch, jitter(batch)):
            errs = _digest_errors(act_grads(b2, True)[3], tmp, "t.")
            worst = max(worst, max(e for e, scale in errs.values() if scale >= 1e-6))
        if best is None or worst < best[0]:
            best = (worst, seed)
        if worst < 0.04:
            break
    worst, seed = best
    print(f"  act: batch seed {seed}, worst bf16 error of the reference (fp32 tokenizer, two realisations) {worst:.3f}")
    assert worst < 0.06, worst
    batch = make_act_batch(B, 150, seed=seed, ragged=True, num_queries=c["num_queries"])
    model, wsum, out, g32 = act_grads(batch, False)
    fx["act.wsum"] = np.array(wsum)
    for k, v in batch.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                fx[f"act.in.pcds.{kk}"] = vv.numpy()
        else:
            fx[f"act.in.{k}"] = v.numpy()
    for k in ("a_hat", "mu", "logvar", "loss", "action_loss", "kl_loss", "src"):
        fx[f"act.out.{k}"] = out[k].detach().numpy()
    _store_grads(fx, "act.grad.", g32)
    for tag, fp32_tok in (("yard", True), ("yard_all", False)):
        _, _, out16, g16 = act_grads(batch, True, fp32_tok)
        for name, (e, scale) in _digest_errors(g16, fx, "act.grad.").items():
            fx[f"act.{tag}.{name}"] = np.array(e if scale >= 1e-6 else 0.0)
        fx[f"act.{tag}_out.loss"] = out16["loss"].detach().float().numpy()
    ya = max(float(fx[k]) for k in fx if k.startswith("act.yard."))
    yb = max(float(fx[k]) for k in fx if k.startswith("act.yard_all."))
    print(f"wide_bf16_ref.npz act: loss {float(out['loss'].detach()):.4f}; reference in bf16: worst tensor {ya:.3f} (fp32 tokenizer) / {yb:.3f} (all autocast)")

    # ---- Diffusion Policy (our classes are pinned to the reference's at 1e-4 by wide_ref.npz / dp_pcd_small.npz; the reference
    # encoder + U-Net are driven directly here, as in golden_wide)
    Md = 64
    shape_meta = {"obs": {"pcds": {"shape": [6], "type": "pcd"}, "qpos": {"shape": [9], "type": "low_dim"}}, "action": {"shape": [7]}}
    ours = build_dp_policy(pcd_npoints=Md, pointops=pointops_cpu, sa_impl="reference", **WIDE_DP)
    fx["dp.wsum"] = np.array(seeded_fill(ours, WIDE_SEED + 1))
    sd = ours.state_dict()
    mgen = ref.maskgen.LowdimMaskGenerator(action_dim=7, obs_dim=0, max_n_obs_steps=2, fix_obs_steps=True, action_visible=False)
    noise = torch.randn(B, 16, 7, generator=torch.Generator().manual_seed(39))
    timesteps = torch.randint(0, 100, (B,), generator=torch.Generator().manual_seed(40))

    def dp_grads(dbatch, bf16, fp32_tokenizer=True):
        enc = ref.pcd_enc.PCDObsEncoder(shape_meta=shape_meta, pcd_model=PointNet(in_channels=6, num_classes=96), share_pcd_model=True,
                                        n_obs_step=2, pcd_nsample=16, pcd_npoints=Md, pcd_hidden_dim=96, projector_layers=1,
                                        projector_channels=[96, 128, 128])
        unet = ref.unet.ConditionalUnet1D(input_dim=7, local_cond_dim=None, global_cond_dim=(128 + 9) * 2, diffusion_step_embed_dim=128,
                                          down_dims=[128, 256], kernel_size=5, n_groups=8, cond_predict_scale=True)
        enc.load_state_dict({k[len("obs_encoder."):]: v for k, v in sd.items() if k.startswith("obs_encoder.")}, strict=True)
        unet.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}, strict=True)
        enc.train(), unet.train()
        if fp32_tokenizer:
            enc.encode_pcd = _fp32_island(enc.encode_pcd)
        qpos, action = dbatch["obs"]["qpos"], dbatch["action"]
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=bf16):
            this_nobs = {"qpos": qpos[:, :2].reshape(-1, 9), "pcds": {k: v.clone() for k, v in dbatch["obs"]["pcds"].items()}}
            global_cond = enc(this_nobs).reshape(B, -1)
            mask = mgen((B, 16, 7))
            acp = ours.noise_scheduler.alphas_cumprod[timesteps]  # diffusers absent: our restated schedule (parity unpinned)
            noisy = acp.sqrt()[:, None, None] * action + (1 - acp).sqrt()[:, None, None] * noise
            noisy[mask] = action[mask]
            pred = unet(noisy, timesteps, local_cond=None, global_cond=global_cond)
            loss = torch.nn.functional.mse_loss(pred.float(), noise, reduction="none") * (~mask).float()
            loss = loss.reshape(B, -1).mean(1).mean()
        loss.backward()
        grads = [("obs_encoder." + n, p.grad.detach().float()) for n, p in enc.named_parameters() if p.grad is not None] \
            + [("model." + n, p.grad.detach().float()) for n, p in unet.named_parameters() if p.grad is not None]
        return loss.detach().float(), pred.detach().float(), grads

    def djitter(dbatch, rel=1e-3):
        g = torch.Generator().manual_seed(6)
        out = dict(dbatch, obs=dict(dbatch["obs"]))
        out["obs"]["qpos"] = dbatch["obs"]["qpos"] * (1 + rel * torch.randn(dbatch["obs"]["qpos"].shape, generator=g))
        out["action"] = dbatch["action"] * (1 + rel * torch.randn(dbatch["action"].shape, generator=g))
        return out

    best = None
    for seed in range(2500, 2530):
        dbatch = make_dp_batch(B, 100, seed=seed, ragged=True)
        tmp = {}
        _store_grads(tmp, "t.", dp_grads(dbatch, False)[2])
        worst = 0.0
        for b2 in (dbatch, djitter(dbatch)):
            errs = _digest_errors(dp_grads(b2, True)[2], tmp, "t.")
            worst = max(worst, max(e for e, scale in errs.values() if scale >= 1e-6))
        if best is None or worst < best[0]:
            best = (worst, seed)
        if worst < 0.04:
            break
    worst, seed = best
    print(f"  dp: batch seed {seed}, worst bf16 error of the reference (fp32 tokenizer, two realisations) {worst:.3f}")
    assert worst < 0.06, worst
    dbatch = make_dp_batch(B, 100, seed=seed, ragged=True)
    loss, pred, g32 = dp_grads(dbatch, False)
    fx["dp.noise"], fx["dp.timesteps"] = noise.numpy(), timesteps.numpy()
    fx["dp.out.loss"], fx["dp.out.pred"] = loss.numpy(), pred.numpy()
    for k, v in dbatch["obs"]["pcds"].items():
        fx[f"dp.in.pcds.{k}"] = v.numpy()
    fx["dp.in.qpos"], fx["dp.in.action"] = dbatch["obs"]["qpos"].numpy(), dbatch["action"].numpy()
    _store_grads(fx, "dp.grad.", g32)
    for tag, fp32_tok in (("yard", True), ("yard_all", False)):
        l16, _, g16 = dp_grads(dbatch, True, fp32_tok)
        for name, (e, scale) in _digest_errors(g16, fx, "dp.grad.").items():
            fx[f"dp.{tag}.{name}"] = np.array(e if scale >= 1e-6 else 0.0)
        fx[f"dp.{tag}_out.loss"] = l16.numpy()
    ya = max(float(fx[k]) for k in fx if k.startswith("dp.yard."))
    yb = max(float(fx[k]) for k in fx if k.startswith("dp.yard_all."))
    print(f"wide_bf16_ref.npz dp: loss {float(loss):.5f}; reference in bf16: worst tensor {ya:.3f} (fp32 tokenizer) / {yb:.3f} (all autocast)")
    np.savez_compressed(os.path.join(OUT, "wide_bf16_ref.npz"), **fx)


def golden_rollout(ref):
    """The policy side of a rollout step, from the reference's own Python:
      * TemporalAgg (src/utils/misc.py:88-141) fed a seeded sequence of action chunks;
      * ACTPCD in eval mode without "actions" (act.py:177-182: zero latent), weights = act_pcd_small.npz plus seeded
        BatchNorm running statistics;
      * the DDPM sampler: the reference ConditionalUnet1D + PCDObsEncoder in eval mode driven by the restated
        scheduler (oracle/ddpm_cpu.py -- diffusers is absent, so the scheduler arithmetic itself is parity-unpinned),
        composed as conditional_sample / predict_action do (diffusion_unet_image_policy.py:106-229).
    Inputs and weights are those of act_pcd_small.npz / dp_pcd_small.npz; this fixture adds only the new draws and
    the expected outputs."""
    from oracle import ddpm_cpu
    from pointcloudmatters_amd.policy import PointNet

    fx = {}
    # ---- TemporalAgg
    misc = _load("src.utils.misc", f"{REF}/src/utils/misc.py")
    rng = np.random.default_rng(17)
    chunks = rng.normal(size=(17, 6, 3))
    agg = misc.TemporalAgg(apply=True, action_dim=3, chunk_size=6, k=0.01)
    outs = [agg(c) for c in chunks[:14]]
    agg.reset()
    outs += [agg(c) for c in chunks[14:]]
    fx["tagg.chunks"], fx["tagg.out"], fx["tagg.reset_after"] = chunks, np.stack(outs), np.array(14)
    fx["tagg.noapply"] = misc.TemporalAgg(apply=False)(chunks[0])

    # ---- ACT, test-time branch
    act = np.load(os.path.join(OUT, "act_pcd_small.npz"))
    weights = {k[2:]: torch.from_numpy(act[k]) for k in act.files if k.startswith("w.")}
    g = torch.Generator().manual_seed(31)
    for k in list(weights):
        if k.endswith("running_mean"):
            weights[k] = 0.1 * torch.randn(weights[k].shape, generator=g)
            fx[f"act.buf.{k}"] = weights[k].numpy()
        elif k.endswith("running_var"):
            weights[k] = 0.5 + torch.rand(weights[k].shape, generator=g)
            fx[f"act.buf.{k}"] = weights[k].numpy()
    holder = types.SimpleNamespace(state_dict=lambda: weights)
    model = build_reference_actpcd(ref, holder, 32).eval()
    pcds = {k[len("in.pcds."):]: torch.from_numpy(act[k]) for k in act.files if k.startswith("in.pcds.")}
    with torch.no_grad():
        out = model({"qpos": torch.from_numpy(act["in.qpos"]), "goal_cond": torch.from_numpy(act["in.goal_cond"]), "pcds": pcds})
    assert out["mu"] is None and not out["is_training"]
    fx["act.a_hat"], fx["act.is_pad_hat"] = out["a_hat"].numpy(), out["is_pad_hat"].numpy()

    # ---- Diffusion Policy sampler
    dp = np.load(os.path.join(OUT, "dp_pcd_small.npz"))
    sd = {k[2:]: torch.from_numpy(dp[k]) for k in dp.files if k.startswith("w.")}
    for k in list(sd):
        if k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
            fx[f"dp.buf.{k}"] = sd[k].numpy()
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
            fx[f"dp.buf.{k}"] = sd[k].numpy()
    shape_meta = {"obs": {"pcds": {"shape": [6], "type": "pcd"}, "qpos": {"shape": [9], "type": "low_dim"}},
                  "action": {"shape": [7]}}
    enc = ref.pcd_enc.PCDObsEncoder(shape_meta=shape_meta, pcd_model=PointNet(in_channels=6, num_classes=24),
                                    share_pcd_model=True, n_obs_step=2, pcd_nsample=16, pcd_npoints=32,
                                    pcd_hidden_dim=24, projector_layers=1, projector_channels=[24, 40, 40])
    enc.load_state_dict({k[len("obs_encoder."):]: v for k, v in sd.items() if k.startswith("obs_encoder.")}, strict=True)
    unet = ref.unet.ConditionalUnet1D(input_dim=7, local_cond_dim=None, global_cond_dim=(40 + 9) * 2,
                                      diffusion_step_embed_dim=16, down_dims=[16, 32, 64], kernel_size=5, n_groups=8,
                                      cond_predict_scale=True)
    unet.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}, strict=True)
    enc.eval(), unet.eval()
    qpos = torch.from_numpy(dp["in.qpos"])
    dpc = {k[len("in.pcds."):]: torch.from_numpy(dp[k]) for k in dp.files if k.startswith("in.pcds.")}
    noises = torch.randn(101, 3, 16, 7, generator=g).numpy()
    with torch.no_grad():
        global_cond = enc({"qpos": qpos[:, :2].reshape(-1, 9), "pcds": dpc}).reshape(3, -1)

        def eps_model(x, t):
            return unet(torch.from_numpy(x), t, local_cond=None, global_cond=global_cond).numpy()

        traj = ddpm_cpu.sample(eps_model, (3, 16, 7), noises, num_train=100, num_inference=100, clip=1.0)
    fx["dp.noises"], fx["dp.global_cond"], fx["dp.action_pred"] = noises, global_cond.numpy(), traj
    fx["dp.action"] = traj[:, 1:1 + 8]  # start = To - 1, n_action_steps = 8 (identity normaliser)
    np.savez_compressed(os.path.join(OUT, "rollout_ref.npz"), **fx)
    print("rollout_ref.npz ok; |action_pred| max", float(np.abs(traj).max()))


def _install_optim_builders():
    """The reference's optimizer / scheduler builders, loaded by path.  omegaconf is absent: a dict with attribute access stands in for the
    DictConfig (import-time names + `to_container`).  Returns (Cfg class, optimizer module, scheduler module, restore function)."""
    import copy

    class Cfg(dict):  # DictConfig stand-in: attribute access over a dict (deepcopy-able, `in`, .get, .copy, .keys)
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

        def __deepcopy__(self, memo):
            return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})

    om = types.ModuleType("omegaconf")
    om.DictConfig = Cfg
    om.OmegaConf = types.SimpleNamespace(to_container=lambda cfg, resolve=False: dict(cfg))
    sys.modules["omegaconf"] = om
    _load("src.utils.misc", f"{REF}/src/utils/misc.py")
    _load("src.utils.registry", f"{REF}/src/utils/registry.py")
    sched = _load("src.utils.scheduler", f"{REF}/src/utils/scheduler.py")
    optim = _load("src.utils.optimizer", f"{REF}/src/utils/optimizer.py")
    # the reference's scheduler subclasses still pass `verbose=` (scheduler.py:118-139), which torch 2.10 removed from the base class: the
    # BASE initialiser is wrapped to drop it (torch side, like the torchvision stub); the reference subclass itself runs unmodified
    base_init = torch.optim.lr_scheduler.OneCycleLR.__init__

    def tolerant_init(self, *a, verbose=False, **kw):
        return base_init(self, *a, **kw)

    torch.optim.lr_scheduler.OneCycleLR.__init__ = tolerant_init

    def restore():
        torch.optim.lr_scheduler.OneCycleLR.__init__ = base_init

    return Cfg, optim, sched, restore


def golden_optim(ref):
    """configure_optimizers through the REFERENCE's own builders (maniskill2_act_bc_module.py:347-367 -> build_optimizer + build_scheduler;
    maniskill2_dp_bc_module.py:326-344 -> build_optimizer_v2 + build_scheduler), with the YAML values of
    configs/model/maniskill2_{act_pcd,diffusion_policy}_model.yaml:10-24.  omegaconf is absent: a dict with attribute access stands in for
    the DictConfig (import-time names + `to_container`); the registry, the scheduler subclass and the grouping functions are the
    reference's files, loaded by path.  Stored: the parameter NAMES of every group with its hyper-parameters, and the learning rate /
    beta1 in effect at every optimizer step of a 200-step cycle."""
    from tests.util import optimizer_zoo

    Cfg, optim, sched, restore = _install_optim_builders()

    fx, T = {}, 200
    for tag, builder, ocfg, scfg in (
            ("act", lambda c, m: optim.build_optimizer(c, m, None), Cfg(type="AdamW", lr=0.00005, weight_decay=0.05),
             Cfg(type="OneCycleLR", max_lr=0.00005, pct_start=0.1, anneal_strategy="cos", div_factor=100.0, final_div_factor=1000.0)),
            ("dp", lambda c, m: optim.build_optimizer_v2(c, m), Cfg(type="AdamW", betas=[0.9, 0.95], lr=0.0001, weight_decay=0.0001),
             Cfg(type="OneCycleLR", max_lr=0.0001, pct_start=0.15, anneal_strategy="cos", div_factor=100.0, final_div_factor=1000.0))):
        zoo = optimizer_zoo()
        names = {id(p): n for n, p in zoo.named_parameters()}
        opt = builder(ocfg, zoo)
        assert type(opt) is torch.optim.AdamW
        fx[f"{tag}.n_groups"] = np.array(len(opt.param_groups))
        for gi, g in enumerate(opt.param_groups):
            fx[f"{tag}.group{gi}.names"] = np.array([names[id(p)] for p in g["params"]])
            fx[f"{tag}.group{gi}.weight_decay"] = np.array(g["weight_decay"], np.float64)
            fx[f"{tag}.group{gi}.eps"] = np.array(g["eps"], np.float64)
            fx[f"{tag}.group{gi}.beta2"] = np.array(g["betas"][1], np.float64)
        scfg.total_steps = T  # the module sets trainer.estimated_stepping_batches here
        sch = optim.build_scheduler if hasattr(optim, "build_scheduler") else sched.build_scheduler
        sch = sch(scfg, optimizer=opt)
        assert isinstance(sch, torch.optim.lr_scheduler.OneCycleLR)
        lr, b1 = np.zeros((T, len(opt.param_groups))), np.zeros((T, len(opt.param_groups)))
        for k in range(T):
            for gi, g in enumerate(opt.param_groups):
                lr[k, gi], b1[k, gi] = g["lr"], g["betas"][0]
            opt.step()
            if k + 1 < T:
                sch.step()
        fx[f"{tag}.lr"], fx[f"{tag}.beta1"] = lr, b1
    restore()
    np.savez_compressed(os.path.join(OUT, "optim_ref.npz"), **fx)
    print("optim_ref.npz: act groups", int(fx["act.n_groups"]), "dp groups", int(fx["dp.n_groups"]), "lr[0], lr[peak], lr[-1] =",
          fx["act.lr"][0, 0], fx["act.lr"].max(), fx["act.lr"][-1, 0])


def golden_normalizer(ref):
    """LinearNormalizer (src/utils/diffusion_policy/normalizer.py:14-300) as the Diffusion-Policy datasets build it: per key through
    `get_range_normalizer_from_stat` (src/utils/normalize_utils.py:7-21; maniskill2_single_task_pcd_dp.py:96-111), and through `fit`.
    One action dimension is constant (the `ignore_dim` branch).  zarr is absent: an empty `zarr.Array` class satisfies the isinstance checks."""
    z = types.ModuleType("zarr")
    z.Array = type("Array", (), {})
    sys.modules["zarr"] = z
    dpu = sys.modules["src.utils.diffusion_policy"]
    mix = _load("src.utils.diffusion_policy.dict_of_tensor_mixin", f"{REF}/src/utils/diffusion_policy/dict_of_tensor_mixin.py")
    dpu.DictOfTensorMixin = mix.DictOfTensorMixin
    nz = _load("src.utils.diffusion_policy.normalizer", f"{REF}/src/utils/diffusion_policy/normalizer.py")
    dpu.LinearNormalizer, dpu.SingleFieldLinearNormalizer = nz.LinearNormalizer, nz.SingleFieldLinearNormalizer
    nu = _load("src.utils.normalize_utils", f"{REF}/src/utils/normalize_utils.py")

    rng = np.random.default_rng(21)
    action = (rng.normal(size=(64, 16, 7)) * np.array([0.05, 0.05, 0.05, 0.3, 0.3, 0.3, 1.0])).astype(np.float32)
    action[..., 6] = 1.0  # a gripper that never moves: range < range_eps
    qpos = (rng.normal(size=(64, 2, 9)) * 1.5 + 0.3).astype(np.float32)
    fx = {"action": action, "qpos": qpos}

    def stat(a):
        f = torch.from_numpy(a).reshape(-1, a.shape[-1])
        return {"min": f.min(0).values.numpy(), "max": f.max(0).values.numpy(), "mean": f.mean(0).numpy(), "std": f.std(0).numpy()}

    a = nz.LinearNormalizer()
    a["action"] = nu.get_range_normalizer_from_stat(stat(action))
    a["qpos"] = nu.get_range_normalizer_from_stat(stat(qpos))
    b = nz.LinearNormalizer()
    b.fit({"action": action, "qpos": qpos})
    for tag, n in (("stat", a), ("fit", b)):
        sd = n.state_dict()
        fx[f"{tag}.keys"] = np.array(sorted(sd))
        for k, v in sd.items():
            fx[f"{tag}.sd.{k}"] = v.detach().numpy()
        out = n.normalize({"action": torch.from_numpy(action[:5]), "qpos": torch.from_numpy(qpos[:5])})
        fx[f"{tag}.norm.action"], fx[f"{tag}.norm.qpos"] = out["action"].detach().numpy(), out["qpos"].detach().numpy()
        y = torch.from_numpy(rng.uniform(-1, 1, size=(4, 16, 7)).astype(np.float32)) if tag == "stat" else torch.from_numpy(fx["y"])
        fx["y"] = y.numpy()
        fx[f"{tag}.unnorm.action"] = n["action"].unnormalize(y).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "normalizer_ref.npz"), **fx)
    print("normalizer_ref.npz:", len(fx), "arrays; action scale", fx["stat.sd.params_dict.action.scale"])


def golden_wrappers(ref):
    """The reference's pointops PYTHON layer (libs/pointops/functions/*.py: autograd Functions, the interpolation weights, sqrt of
    dist2, the -1 masks of grouping, the *_and_group helpers), executed UNMODIFIED as the package `pointops`, over the C oracle's kernels:
    `pointops._C` is a 16-function stub that forwards the pybind argument lists (pointops_api.cpp:15-32) to oracle/pcm_oracle.c, and
    `torch.cuda.{Int,Float}Tensor` -- the wrappers' only CUDA dependence -- are pointed at the CPU tensor types while this runs.
    Pins the WRAPPER level of oracle/pointops_cpu.py (and through it of the product's pointops/*.py) to the reference's own code; the
    kernels under it stay pinned by restatement only."""
    from oracle import lib as olib

    L = olib.load()
    stub = types.ModuleType("pointops._C")

    def forward_to(cname):
        fn = getattr(L, cname)

        scalar = [t is not olib._F for t in olib._SIGS[cname]]  # int / float parameters (pointer parameters are void*)

        def call(*args):
            conv = []
            for a, is_scalar in zip(args, scalar):
                if torch.is_tensor(a) and is_scalar:
                    conv.append(a.item())  # pybind's int caster takes a 0-dim integer tensor (sampling.py:15-20 passes n_max as one)
                elif torch.is_tensor(a):
                    assert a.is_contiguous() and not a.is_cuda
                    conv.append(a.data_ptr())
                else:
                    conv.append(a)
            assert len(conv) == len(scalar), (cname, len(conv), len(scalar))
            rc = fn(*conv)
            if rc != 0:
                raise RuntimeError(f"{cname} returned {rc}")

        return call

    for name in olib._SIGS:
        if name.endswith("_cpu") and name not in ("pcm_opt_n_threads_cpu",):
            op = name[len("pcm_"):-len("_cpu")]
            setattr(stub, op + "_cuda", forward_to(name))
    saved = {k: sys.modules.get(k) for k in ("pointops", "pointops._C")}
    saved_types = (torch.cuda.IntTensor, torch.cuda.FloatTensor)
    torch.cuda.IntTensor, torch.cuda.FloatTensor = torch.IntTensor, torch.FloatTensor
    sys.modules["pointops._C"] = stub
    fdir = f"{REF}/libs/pointops/functions"
    spec = importlib.util.spec_from_file_location("pointops", f"{fdir}/__init__.py", submodule_search_locations=[fdir])
    po = importlib.util.module_from_spec(spec)
    sys.modules["pointops"] = po
    try:
        spec.loader.exec_module(po)
        sys.path.insert(0, ROOT)
        from tests.util import make_clouds, new_offsets

        fx = {}
        g = torch.Generator().manual_seed(17)
        xyz, off = make_clouds([120, 75, 200], seed=17)
        noff = new_offsets([40, 30, 64])
        n, c = xyz.shape[0], 5
        feat = torch.randn(n, c, generator=g)
        fx["xyz"], fx["offset"], fx["new_offset"], fx["feat"] = xyz.numpy(), off.numpy(), noff.numpy(), feat.numpy()
        sel = po.farthest_point_sampling(xyz, off, noff)
        fx["fps.idx"] = sel.numpy()
        q = xyz[sel.long()].contiguous()
        for tag, out in (("knn", po.knn_query(8, xyz, off, q, noff)), ("knn_self", po.knn_query(4, xyz, off)),
                         ("ball", po.ball_query(8, 0.15, 0.02, xyz, off, q, noff))):
            fx[f"{tag}.idx"], fx[f"{tag}.dist"] = out[0].numpy(), out[1].numpy()
        torch.manual_seed(5)
        ridx, rdist = po.random_ball_query(8, 0.15, 0.02, xyz, off, q, noff)
        torch.manual_seed(5)  # the order the wrapper drew (query.py:46-52), for callers that take it as an argument
        order, prev = [], 0
        for e in off.tolist():
            order.append(torch.randperm(e - prev, dtype=torch.int32) + prev)
            prev = e
        fx["rball.order"], fx["rball.idx"], fx["rball.dist"] = torch.cat(order).numpy(), ridx.numpy(), rdist.numpy()
        kidx = torch.from_numpy(fx["knn.idx"])

        def with_grads(tag, fn, inputs):
            leaves = [t.clone().requires_grad_(True) for t in inputs]
            out = fn(*leaves)
            w = torch.randn(out.shape, generator=g)
            grads = torch.autograd.grad((out * w).sum(), leaves, allow_unused=True)
            fx[f"{tag}.out"], fx[f"{tag}.w"] = out.detach().numpy(), w.numpy()
            for i, (t, gr) in enumerate(zip(inputs, grads)):
                fx[f"{tag}.in{i}"] = t.numpy()
                if gr is not None:  # (the relation step returns no gradient for its weight vector, attention.py:62)
                    fx[f"{tag}.grad{i}"] = gr.numpy()

        with_grads("grouping2", lambda f: po.grouping2(f, kidx), [feat])  # grouping.py:62: grouping2 = Grouping.apply (input, idx)
        with_grads("grouping_xyz", lambda f: po.grouping(kidx, f, xyz, q, with_xyz=True), [feat])
        with_grads("interp", lambda f: po.interpolation(q, xyz, f, noff, off, k=3), [feat[: q.shape[0]].contiguous()])
        with_grads("interp2", lambda f: po.interpolation2(q, xyz, f, noff, off, 3), [feat[: q.shape[0]].contiguous()])
        m = q.shape[0]
        sidx = po.knn_query(6, xyz, off)[0]
        with_grads("subtraction", lambda a, b: po.subtraction(a, b, sidx), [feat, torch.randn(n, c, generator=g)])
        with_grads("aggregation", lambda a, pz, wt: po.aggregation(a, pz, wt, sidx),
                   [torch.randn(n, 8, generator=g), torch.randn(n, 6, 8, generator=g), torch.randn(n, 6, 4, generator=g)])
        fx["sidx"] = sidx.numpy()
        it = torch.randint(0, n, (300,), generator=g).int()
        ir = torch.randint(0, n, (300,), generator=g).int()
        fx["attn.index_target"], fx["attn.index_refer"] = it.numpy(), ir.numpy()
        with_grads("attn_relation", lambda a, b, wt: po.attention_relation_step(a, b, wt, it, ir),
                   [torch.randn(n, 2, 4, generator=g), torch.randn(n, 2, 4, generator=g), torch.randn(4, generator=g)])
        with_grads("attn_fusion", lambda wt, v: po.attention_fusion_step(wt, v, it, ir),
                   [torch.randn(300, 2, generator=g), torch.randn(n, 2, 4, generator=g)])
        for tag, fn in (("knn_group", lambda f: po.knn_query_and_group(f, xyz, off, q, noff, nsample=8, with_xyz=True)),
                        ("ball_group", lambda f: po.ball_query_and_group(f, xyz, off, q, noff, max_radio=0.15, min_radio=0.02,
                                                                         nsample=8, with_xyz=True))):
            out = fn(feat)
            out = out[0] if isinstance(out, tuple) else out
            fx[f"{tag}.out"] = out.numpy()
        # query_and_group (utils.py:42-99): dilated neighbourhoods; the middle cloud (75 points) is smaller than 1 + 7 * 12 = 85 -> soft dilation
        for tag, dil in (("qg_d0", 0), ("qg_d1", 1), ("qg_soft", 11)):
            out, gidx = po.query_and_group(8, xyz, q, feat, None, off, noff, dilation=dil, with_feat=True, with_xyz=True)
            fx[f"{tag}.out"], fx[f"{tag}.idx"] = out.numpy(), gidx.numpy()
        np.savez_compressed(os.path.join(OUT, "wrappers_ref.npz"), **fx)
        print("wrappers_ref.npz:", len(fx), "arrays; m =", m, "; ball rows with -1:", int((fx["ball.idx"] < 0).any(1).sum()))
    finally:
        torch.cuda.IntTensor, torch.cuda.FloatTensor = saved_types
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def golden_trajectory(ref):
    """Six optimizer steps of the reference's training recipe, end to end: the reference `ACTPCD` (weights of act_pcd_small.npz), the
    optimizer and scheduler from the reference's `build_optimizer` / `build_scheduler` (maniskill2_act_bc_module.py:347-367), and per step what
    Lightning's automatic optimisation does with `gradient_clip_val: 0.5` (configs/trainer/default.yaml): zero_grad, forward, backward,
    clip_grad_norm_, optimizer.step, scheduler.step.  lr is raised to 1e-3 so that six steps move the weights visibly."""
    from pointcloudmatters_amd.bc import make_act_batch

    Cfg, optim, sched, restore = _install_optim_builders()
    pcd_npoints = 32
    ours = build_ours(pcd_npoints, seed=1234)
    model = build_reference_actpcd(ref, ours, pcd_npoints)
    model.train()
    opt = optim.build_optimizer(Cfg(type="AdamW", lr=0.001, weight_decay=0.05), model, None)
    sch = sched.build_scheduler(Cfg(type="OneCycleLR", max_lr=0.001, pct_start=0.1, anneal_strategy="cos", div_factor=100.0,
                                    final_div_factor=1000.0, total_steps=40), optimizer=opt)
    batches = [make_act_batch(3, 180, seed=77 + i, ragged=True, num_queries=SMALL["num_queries"]) for i in range(2)]
    eps = [torch.randn(3, SMALL["latent_dim"], generator=torch.Generator().manual_seed(5 + i)) for i in range(2)]
    fx = {"eps0": eps[0].numpy(), "eps1": eps[1].numpy()}
    for i, b in enumerate(batches):
        for k, v in b.items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    fx[f"in{i}.pcds.{kk}"] = vv.numpy()
            else:
                fx[f"in{i}.{k}"] = v.numpy()
    orig = ref.act.reparametrize
    losses, norms, lrs = [], [], []
    try:
        for step in range(6):
            e = eps[step % 2]
            ref.act.reparametrize = lambda mu, logvar, e=e: mu + logvar.div(2).exp() * e
            b = batches[step % 2]
            dd = {k: (dict(v) if isinstance(v, dict) else v.clone()) for k, v in b.items()}
            dd["pcds"] = {k: v.clone() for k, v in dd["pcds"].items()}
            opt.zero_grad()
            out = model(dd)
            out["loss"].backward()
            norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)))
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
            losses.append(float(out["loss"]))
    finally:
        ref.act.reparametrize = orig
        restore()
    fx["loss"], fx["grad_norm"], fx["lr"] = np.array(losses), np.array(norms), np.array(lrs)
    sd = model.state_dict()
    for k in ("linear.weight", "bn.weight", "bn.running_mean", "backbone.conv1.0.weight", "transformer.encoder.layers.0.linear1.weight",
              "transformer.decoder.layers.0.multihead_attn.out_proj.weight", "transformer.decoder.layers.2.linear1.weight",
              "action_head.weight", "action_head.bias", "query_embed.weight"):
        fx[f"final.{k}"] = sd[k].numpy()
    np.savez_compressed(os.path.join(OUT, "trajectory_ref.npz"), **fx)
    print("trajectory_ref.npz: losses", [round(x, 4) for x in losses], "grad norms", [round(x, 2) for x in norms])


def golden_dp_trajectory(ref):
    """Six optimizer steps of the Diffusion-Policy recipe: the reference `PCDObsEncoder` + `ConditionalUnet1D` + `LowdimMaskGenerator`
    composed as compute_loss does (as in golden_dp; weights of dp_pcd_small.npz), optimizer from the reference's `build_optimizer_v2` with the
    YAML's optimizer section INCLUDING its `betas: [0.9, 0.95]` (maniskill2_diffusion_policy_model.yaml:10-14) -- which the builder does not
    forward --, scheduler from `build_scheduler` (pct_start 0.15), Lightning's step order with the 0.5 clip.  lr raised to 1e-3."""
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc import build_dp_policy, make_dp_batch
    from pointcloudmatters_amd.policy import PointNet

    Cfg, optim, sched, restore = _install_optim_builders()
    pcd_npoints = 32
    torch.manual_seed(4321)
    ours = build_dp_policy(pcd_npoints=pcd_npoints, pointops=pointops_cpu, sa_impl="reference", **DP_SMALL)
    sd = ours.state_dict()
    shape_meta = {"obs": {"pcds": {"shape": [6], "type": "pcd"}, "qpos": {"shape": [9], "type": "low_dim"}}, "action": {"shape": [7]}}

    class RefPolicy(torch.nn.Module):  # the attribute names of DiffusionUnetImagePolicy (diffusion_unet_image_policy.py:84-86): same parameter names
        def __init__(self):
            super().__init__()
            self.obs_encoder = ref.pcd_enc.PCDObsEncoder(
                shape_meta=shape_meta, pcd_model=PointNet(in_channels=6, num_classes=24), share_pcd_model=True, n_obs_step=2,
                pcd_nsample=16, pcd_npoints=pcd_npoints, pcd_hidden_dim=24, projector_layers=1, projector_channels=[24, 40, 40])
            self.model = ref.unet.ConditionalUnet1D(input_dim=7, local_cond_dim=None, global_cond_dim=(40 + 9) * 2,
                                                    diffusion_step_embed_dim=16, down_dims=[16, 32, 64], kernel_size=5, n_groups=8,
                                                    cond_predict_scale=True)

    pol = RefPolicy()
    pol.obs_encoder.load_state_dict({k[len("obs_encoder."):]: v for k, v in sd.items() if k.startswith("obs_encoder.")}, strict=True)
    pol.model.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}, strict=True)
    pol.train()
    mg = ref.maskgen.LowdimMaskGenerator(action_dim=7, obs_dim=0, max_n_obs_steps=2, fix_obs_steps=True, action_visible=False)
    opt = optim.build_optimizer_v2(Cfg(type="AdamW", betas=[0.9, 0.95], lr=0.001, weight_decay=0.0001), pol)
    sch = sched.build_scheduler(Cfg(type="OneCycleLR", max_lr=0.001, pct_start=0.15, anneal_strategy="cos", div_factor=100.0,
                                    final_div_factor=1000.0, total_steps=40), optimizer=opt)
    batches = [make_dp_batch(3, 150, seed=11 + i, ragged=True) for i in range(2)]
    g = torch.Generator().manual_seed(8)
    noises = [torch.randn(3, 16, 7, generator=g) for _ in range(2)]
    tsteps = [torch.tensor([3, 57, 99]), torch.tensor([0, 20, 80])]
    fx = {}
    for i, b in enumerate(batches):
        for k, v in b["obs"]["pcds"].items():
            fx[f"in{i}.pcds.{k}"] = v.numpy()
        fx[f"in{i}.qpos"], fx[f"in{i}.action"] = b["obs"]["qpos"].numpy(), b["action"].numpy()
        fx[f"in{i}.noise"], fx[f"in{i}.timesteps"] = noises[i].numpy(), tsteps[i].numpy()
    losses, norms = [], []
    try:
        for step in range(6):
            b, noise, timesteps = batches[step % 2], noises[step % 2], tsteps[step % 2]
            qpos, action = b["obs"]["qpos"], b["action"]
            this_nobs = {"qpos": qpos[:, :2].reshape(-1, 9), "pcds": {k: v.clone() for k, v in b["obs"]["pcds"].items()}}
            opt.zero_grad()
            global_cond = pol.obs_encoder(this_nobs).reshape(3, -1)
            mask = mg((3, 16, 7))
            acp = ours.noise_scheduler.alphas_cumprod[timesteps]  # diffusers absent: our restated schedule (parity unpinned)
            noisy = acp.sqrt()[:, None, None] * action + (1 - acp).sqrt()[:, None, None] * noise
            noisy[mask] = action[mask]
            pred = pol.model(noisy, timesteps, local_cond=None, global_cond=global_cond)
            loss = (torch.nn.functional.mse_loss(pred, noise, reduction="none") * (~mask).float()).reshape(3, -1).mean(1).mean()
            loss.backward()
            norms.append(float(torch.nn.utils.clip_grad_norm_(pol.parameters(), 0.5)))
            opt.step()
            sch.step()
            losses.append(float(loss))
    finally:
        restore()
    fx["loss"], fx["grad_norm"] = np.array(losses), np.array(norms)
    fx["beta2"] = np.array([g_["betas"][1] for g_ in opt.param_groups])
    fsd = pol.state_dict()
    for k in ("obs_encoder.linear.weight", "obs_encoder.projector.0.weight", "obs_encoder.key_model_map.pcd.conv1.0.weight",
              "model.diffusion_step_encoder.1.weight", "model.diffusion_step_encoder.1.bias", "model.down_modules.0.0.blocks.0.block.0.weight",
              "model.mid_modules.1.cond_encoder.1.weight", "model.final_conv.1.weight", "model.final_conv.1.bias",
              "model.up_modules.0.0.blocks.0.block.1.weight"):
        fx[f"final.{k}"] = fsd[k].numpy()
    np.savez_compressed(os.path.join(OUT, "dp_trajectory_ref.npz"), **fx)
    print("dp_trajectory_ref.npz: losses", [round(x, 4) for x in losses], "grad norms", [round(x, 2) for x in norms], "beta2", fx["beta2"])


def golden_gridsample(ref):
    """GridSamplePCD (fnv, train, return_grid_coord) + NormalizeColorPCD from transformpcd.py, run as shipped on three
    seeded clouds (NumPy 2.2.6 here: coord / np.array(grid_size) promotes to float64)."""
    tp = _load("ref_transformpcd", f"{REF}/src/data/components/transformpcd.py")
    rng = np.random.default_rng(123)
    fx = {"grid_size": np.array(0.005)}
    gs = tp.GridSamplePCD(grid_size=0.005, hash_type="fnv", mode="train", keys=("coord", "color"), return_grid_coord=True)
    norm = tp.NormalizeColorPCD()
    for i, n in enumerate((3000, 1777, 4096)):
        coord = np.empty((n, 3), dtype=np.float32)
        coord[:, :2] = rng.uniform(-0.12, 0.12, (n, 2))
        coord[:, 2] = rng.uniform(0.005, 0.1, n)
        coord[n // 2:] = coord[: n - n // 2] + rng.uniform(0, 0.004, (n - n // 2, 3)).astype(np.float32)  # crowd the voxels
        color = rng.integers(0, 256, (n, 3)).astype(np.float32)
        scaled = coord / np.array(0.005)
        grid_all = np.floor(scaled).astype(int)
        grid_all -= grid_all.min(0)
        fx[f"{i}.coord"], fx[f"{i}.color"] = coord, color
        fx[f"{i}.grid_all"] = grid_all.astype(np.int64)
        fx[f"{i}.key_all"] = tp.GridSamplePCD.fnv_hash_vec(grid_all)
        np.random.seed(1000 + i)
        out = norm(gs({"coord": coord.copy(), "color": color.copy()}))
        fx[f"{i}.out.coord"], fx[f"{i}.out.color"] = out["coord"], out["color"]
        fx[f"{i}.out.grid_coord"] = out["grid_coord"].astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "gridsample_ref.npz"), **fx)
    print("gridsample_ref.npz ok:", [int(fx[f"{i}.out.coord"].shape[0]) for i in range(3)], "voxels")


if __name__ == "__main__":
    assert os.path.isdir(REF), "run this in the build container (needs /root/reference)"
    torch.set_num_threads(1)
    ref = install_reference()
    only = set(sys.argv[1:])  # e.g. `make_golden.py rollout` regenerates one fixture
    for name, fn in (("act", golden_act), ("grouping", golden_grouping), ("misc", golden_misc), ("dp", golden_dp),
                     ("rollout", golden_rollout), ("gridsample", golden_gridsample), ("rlbench", golden_rlbench),
                     ("dp_rlbench", golden_dp_rlbench), ("mask", golden_mask), ("presample", golden_presample), ("wide", golden_wide), ("wide_bf16", golden_wide_bf16),
                     ("optim", golden_optim), ("normalizer", golden_normalizer), ("wrappers", golden_wrappers), ("trajectory", golden_trajectory), ("dp_trajectory", golden_dp_trajectory)):
        if not only or name in only:
            fn(ref)
