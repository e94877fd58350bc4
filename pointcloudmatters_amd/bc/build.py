"""Instantiate the policies from bc/configs.py (what hydra.utils.instantiate does for
configs/model/maniskill2_act_pcd_model.yaml:27-68 in the reference)."""
import torch.nn as nn

from ..policy import ACTPCD, KLDivergence, PointNet, Transformer, TransformerEncoder
from .configs import ACT_MODEL


def build_act_policy(pcd_npoints, pointops=None, sa_impl="reference", overlap_sampling=True, **overrides):
    c = dict(ACT_MODEL)
    c.update(overrides)
    backbone = PointNet(in_channels=c["in_channels"], num_classes=0)
    transformer = Transformer(
        d_model=c["hidden_dim"], dropout=c["dropout"], nhead=c["nhead"], dim_feedforward=c["dim_feedforward"],
        num_encoder_layers=c["num_encoder_layers"], num_decoder_layers=c["num_decoder_layers"],
        normalize_before=c["normalize_before"], return_intermediate_dec=c["return_intermediate_dec"],
    )
    encoder = TransformerEncoder(
        d_model=c["hidden_dim"], dropout=c["dropout"], nhead=c["nhead"], dim_feedforward=c["dim_feedforward"],
        num_layers=c["num_encoder_layers"], normalize_before=c["normalize_before"], activation="relu",
    )
    return ACTPCD(
        backbone=backbone, transformer=transformer, encoder=encoder, hidden_dim=c["hidden_dim"],
        num_queries=c["num_queries"], num_cameras=1, action_dim=c["action_dim"], qpos_dim=c["qpos_dim"],
        env_state_dim=0, latent_dim=c["latent_dim"], action_loss=nn.MSELoss(reduction="none"),
        klloss=KLDivergence(), kl_weight=c["kl_weight"], goal_cond_dim=c["goal_cond_dim"],
        pcd_nsample=c["pcd_nsample"], pcd_npoints=pcd_npoints, pointops=pointops, sa_impl=sa_impl,
        overlap_sampling=overlap_sampling,
    )
