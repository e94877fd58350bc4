"""Instantiate the policies from bc/configs.py (what hydra.utils.instantiate does for
configs/model/maniskill2_act_pcd_model.yaml:27-68 in the reference)."""
import torch.nn as nn

from ..policy import ACTPCD, KLDivergence, PointNet, Transformer, TransformerEncoder
from .configs import ACT_MODEL, DP_MODEL, RLBENCH_ACT_MODEL


def build_rlbench_act_policy(pcd_npoints, **kw):
    """ACTRLBenchPCD as configs/model/rlbench_act_pcd_model.yaml instantiates it."""
    return build_act_policy(pcd_npoints, _base=RLBENCH_ACT_MODEL, **kw)


def build_act_policy(pcd_npoints, pointops=None, sa_impl="reference", overlap_sampling=True, dead_decoder_layers="keep",
                     _base=None, backbone="pointnet", **overrides):
    """backbone: "pointnet" (the reference's per-point MLP) or "pointnext" (policy/pointnet2.PointNeXtBackbone: InvResMLP
    blocks in front of the same SA tokenizer -- no reference counterpart, BASELINE configs[3])."""
    c = dict(ACT_MODEL if _base is None else _base)
    c.update(overrides)
    if backbone == "pointnext":
        from ..policy.pointnet2 import PointNeXtBackbone

        backbone = PointNeXtBackbone(in_channels=c["in_channels"], width=c.get("pointnext_width", 64), blocks=c.get("pointnext_blocks", 2),
                                     nsample=c["pcd_nsample"], out_channels=512, pointops=pointops, sa_impl=sa_impl)
    else:
        backbone = PointNet(in_channels=c["in_channels"], num_classes=c.get("backbone_num_classes", 0))
    if c.get("pre_sample", False) and backbone.num_channels != c["hidden_dim"]:
        # act.py:509-530: with pre_sample the backbone's output IS the token matrix
        raise ValueError("pre_sample: the backbone's output width (%d) must equal hidden_dim (%d)" % (backbone.num_channels, c["hidden_dim"]))
    transformer = Transformer(
        d_model=c["hidden_dim"], dropout=c["dropout"], nhead=c["nhead"], dim_feedforward=c["dim_feedforward"],
        num_encoder_layers=c["num_encoder_layers"], num_decoder_layers=c["num_decoder_layers"],
        normalize_before=c["normalize_before"], return_intermediate_dec=c["return_intermediate_dec"],
    )
    encoder = TransformerEncoder(
        d_model=c["hidden_dim"], dropout=c["dropout"], nhead=c["nhead"], dim_feedforward=c["dim_feedforward"],
        num_layers=c["num_encoder_layers"], normalize_before=c["normalize_before"], activation="relu",
    )
    cls, extra = ACTPCD, {}
    if "rot_type" in c:
        from ..policy import ACTRLBenchPCD

        cls = ACTRLBenchPCD
        extra = dict(rot_type=c["rot_type"], collision=c["collision"], position_loss_weight=c["position_loss_weight"])
    return cls(
        backbone=backbone, transformer=transformer, encoder=encoder, hidden_dim=c["hidden_dim"],
        num_queries=c["num_queries"], num_cameras=1, action_dim=c["action_dim"], qpos_dim=c["qpos_dim"],
        env_state_dim=0, latent_dim=c["latent_dim"], action_loss=nn.MSELoss(reduction="none"),
        klloss=KLDivergence(), kl_weight=c["kl_weight"], goal_cond_dim=c["goal_cond_dim"],
        pcd_nsample=c["pcd_nsample"], pcd_npoints=pcd_npoints, pointops=pointops, sa_impl=sa_impl,
        overlap_sampling=overlap_sampling, dead_decoder_layers=dead_decoder_layers,
        use_mask=c.get("use_mask", False), bg_ratio=c.get("bg_ratio", 0.0), pre_sample=c.get("pre_sample", False), **extra,
    )


def build_dp_policy(pcd_npoints, pointops=None, sa_impl="reference", overlap_sampling=True, obs_encoder="pointnet_sa", **overrides):
    """configs/exp_maniskill2_diffusion_policy/maniskill2_model/scratch_pointnet_pcd.yaml, instantiated.
    obs_encoder: "pointnet_sa" (the reference's PCDObsEncoder) or "patchbert" (policy/pointnet2.PatchBertObsEncoder: patch
    tokens + transformer encoder -- no reference counterpart, BASELINE configs[4]; pcd_npoints = number of patches)."""
    from ..policy.diffusion import DDPMSchedule, DiffusionUnetPcdPolicy, PCDObsEncoder

    c = dict(DP_MODEL)
    c.update(overrides)
    shape_meta = {"obs": {"pcds": {"shape": [c["in_channels"]], "type": "pcd"},
                          "qpos": {"shape": [c["qpos_dim"]], "type": "low_dim"}},
                  "action": {"shape": [c["action_dim"]]}}
    if c.get("goal_dim"):  # language goal: rlbench_diffusion_policy_model.yaml:26-28
        shape_meta["goal"] = {"task_emb": {"shape": [int(c["goal_dim"])]}}
    if obs_encoder == "patchbert":
        from ..policy.pointnet2 import PatchBertObsEncoder

        enc = PatchBertObsEncoder(shape_meta, num_groups=pcd_npoints, group_size=c.get("patch_size", 32),
                                  hidden_dim=c.get("bert_dim", 384), depth=c.get("bert_depth", 4), nhead=c.get("bert_heads", 6),
                                  out_channels=c["projector_channels"][-1], n_obs_step=c["n_obs_steps"], pointops=pointops,
                                  sa_impl=sa_impl)
    else:
        pcd_model = PointNet(in_channels=c["in_channels"], num_classes=c["pcd_num_classes"])
        enc = PCDObsEncoder(shape_meta=shape_meta, pcd_model=pcd_model, share_pcd_model=True, n_obs_step=c["n_obs_steps"],
                            pcd_nsample=c["pcd_nsample"], pcd_npoints=pcd_npoints, pcd_hidden_dim=c["pcd_hidden_dim"],
                            projector_layers=c["projector_layers"], projector_channels=c["projector_channels"],
                            pointops=pointops, sa_impl=sa_impl, overlap_sampling=overlap_sampling,
                            use_mask=c.get("use_mask", False), bg_ratio=c.get("bg_ratio", 0.0),
                            pre_sample=c.get("pre_sample", False), in_channel=c["in_channels"])
    sched = DDPMSchedule(num_train_timesteps=c["num_train_timesteps"], beta_schedule="squaredcos_cap_v2",
                         prediction_type="epsilon")
    pol = DiffusionUnetPcdPolicy(shape_meta=shape_meta, noise_scheduler=sched, obs_encoder=enc, horizon=c["horizon"],
                                 n_action_steps=c["n_action_steps"], n_obs_steps=c["n_obs_steps"],
                                 diffusion_step_embed_dim=c["diffusion_step_embed_dim"], down_dims=c["down_dims"],
                                 kernel_size=c["kernel_size"], n_groups=c["n_groups"],
                                 cond_predict_scale=c["cond_predict_scale"])
    # identity-range normaliser by default (synthetic data is already O(1)); datasets call policy.normalizer.fit
    pol.normalizer.set_range("qpos", [-1.0] * c["qpos_dim"], [1.0] * c["qpos_dim"])
    pol.normalizer.set_range("action", [-1.0] * c["action_dim"], [1.0] * c["action_dim"])
    return pol
