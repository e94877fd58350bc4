"""Frozen experiment shapes.  Hydra is absent on the GPU box, so the values the hot
path needs are restated here with the YAML they come from.

ACT hyper-parameters: /root/reference/configs/model/maniskill2_act_pcd_model.yaml:11-68,
exp_maniskill2_act_policy/maniskill2_model/scratch_pointnet_pcd.yaml:10-22,
exp_maniskill2_act_policy/maniskill2_pcd_task/PickCube-v0.yaml (action 7, qpos 9, goal 3),
configs/data/maniskill2_act_pcd_dataset.yaml:14 (chunk_size 100), configs/trainer/ddp.yaml:4-15.
Synthetic cloud sizes follow SURVEY.md section 8(d) / BASELINE.md section 3 (M = N/2).
"""

ACT_MODEL = dict(
    hidden_dim=512, nhead=8, dim_feedforward=32, num_encoder_layers=4, num_decoder_layers=7, dropout=0.1,
    normalize_before=False, return_intermediate_dec=True, latent_dim=32, kl_weight=10.0, num_queries=100,
    action_dim=7, qpos_dim=9, goal_cond_dim=3, in_channels=6, pcd_nsample=16,
)

ACT_OPTIM = dict(lr=5e-5, weight_decay=0.05, pct_start=0.1, div_factor=100.0, final_div_factor=1000.0,
                 gradient_clip_val=0.5, accumulate_grad_batches=2)

# RLBench ACT: /root/reference/configs/model/rlbench_act_pcd_model.yaml:4-65, configs/data/rlbench_act_pcd_dataset.yaml:13-19
# (chunk_size 100, action_dim = qpos_dim = 11: position 3 + 6-D rotation + gripper + collision; 512-d task embedding)
RLBENCH_ACT_MODEL = dict(ACT_MODEL, action_dim=11, qpos_dim=11, goal_cond_dim=512, kl_weight=10.0, rot_type="6d", collision=True,
                         position_loss_weight=10.0)
# exp_rlbench_act_policy/rlbench_model/scratch_pointnet_pcd.yaml:9-12: batch 8, accumulate_grad_batches 4 (tests/test_configs_vs_yaml.py)
RLBENCH_ACT_OPTIM = dict(ACT_OPTIM, lr=1e-4, pct_start=0.15, accumulate_grad_batches=4)

# Diffusion Policy: /root/reference/configs/model/maniskill2_diffusion_policy_model.yaml:10-60,
# exp_maniskill2_diffusion_policy/maniskill2_model/scratch_pointnet_pcd.yaml:10-35 (PointNet num_classes 96,
# SA hidden 96, projector [96,128,128] with 1 layer), data chunk_size 16, qpos 9-d.
DP_MODEL = dict(
    horizon=16, n_action_steps=8, n_obs_steps=2, num_train_timesteps=100, diffusion_step_embed_dim=128,
    down_dims=(512, 1024, 2048), kernel_size=5, n_groups=8, cond_predict_scale=True, action_dim=7, qpos_dim=9,
    in_channels=6, pcd_num_classes=96, pcd_hidden_dim=96, projector_layers=1, projector_channels=(96, 128, 128),
    pcd_nsample=16,
)

# betas: the YAML asks for [0.9, 0.95] (maniskill2_diffusion_policy_model.yaml:12), but `build_optimizer_v2` -- what the module's
# configure_optimizers calls (maniskill2_dp_bc_module.py:326-327) -- forwards only lr / foreach / type from the config
# (src/utils/optimizer.py:304-318): the optimizer the reference TRAINS with has torch's default betas (0.9, 0.999), beta1 then cycled by
# OneCycleLR.  Found by running the reference's own builders (tests/golden/optim_ref.npz: beta2 = 0.999 in both groups); the effective
# values are what the trainer uses, the YAML's are kept beside them (tests/test_configs_vs_yaml.py reads `yaml_betas`).
DP_OPTIM = dict(lr=1e-4, weight_decay=1e-4, betas=(0.9, 0.999), yaml_betas=(0.9, 0.95), pct_start=0.15, div_factor=100.0,
                final_div_factor=1000.0, gradient_clip_val=0.5, accumulate_grad_batches=1, filter_bias_and_bn=True)

# RLBench Diffusion Policy: /root/reference/configs/model/rlbench_diffusion_policy_model.yaml:3-28 (AdamW lr 1e-4, wd 0.05, default
# betas, build_optimizer_v2 -> no decay on biases / norm weights; goal = 512-d task embedding appended to the global
# condition, diffusion_unet_image_policy.py:58-62,262-266), configs/data/rlbench_diffusion_policy_pcd_dataset.yaml:12-18
# (chunk_size 16, action_dim 11 = position 3 + 6-D rotation + gripper + collision; qpos has the same width),
# exp_rlbench_diffusion_policy/rlbench_model/scratch_pointnet_pcd.yaml:9-13 (batch 16, accumulate_grad_batches 2).
RLBENCH_DP_MODEL = dict(DP_MODEL, action_dim=11, qpos_dim=11, goal_dim=512)
RLBENCH_DP_OPTIM = dict({k: v for k, v in DP_OPTIM.items() if k != "yaml_betas"}, weight_decay=0.05, betas=(0.9, 0.999), accumulate_grad_batches=2)

# name -> per-GPU batch, points per cloud, tokens per cloud (pcd_npoints), compute dtype
WORKLOADS = {
    # BASELINE.json configs[0]: CPU plumbing case
    "C1": dict(policy="act", batch=2, n_points=512, pcd_npoints=128, dtype="fp32", ragged=False),
    # configs[1]: the single-GPU case the headline metric is quoted on
    "C2": dict(policy="act", batch=8, n_points=1024, pcd_npoints=512, dtype="bf16", ragged=False),
    # configs[3]: per-GPU shape of the 8-GPU ACT run
    "C4": dict(policy="act", batch=8, n_points=2048, pcd_npoints=1024, dtype="bf16", ragged=False),
    # the shipped ACT config (scratch_pointnet_pcd.yaml:10, maniskill2_act_pcd_model.yaml:67-68)
    # configs[2]: StackCube, 1024-pt clouds, Diffusion Policy (B=64 samples x To=2 clouds)
    "C3": dict(policy="dp", batch=64, n_points=1024, pcd_npoints=512, dtype="bf16", ragged=False),
    # configs[4]: per-GPU shape of the RLBench 4096-pt Diffusion-Policy run
    "C5": dict(policy="dp", batch=16, n_points=4096, pcd_npoints=2048, dtype="bf16", ragged=False),
    "REF": dict(policy="act", batch=8, n_points=4096, pcd_npoints=2048, dtype="bf16", ragged=True),
    # configs[3] with the backbone it names: PointNeXt (InvResMLP blocks, policy/pointnet2.py) in front of the SA tokenizer
    "C4N": dict(policy="act", batch=8, n_points=2048, pcd_npoints=1024, dtype="bf16", ragged=False, backbone="pointnext"),
    # configs[4] with the encoder it names: PointBERT-style patch tokens + transformer encoder -> Diffusion Policy
    "C5B": dict(policy="dp", batch=16, n_points=4096, pcd_npoints=128, dtype="bf16", ragged=False, obs_encoder="patchbert"),
    # configs[4]-shaped ACT: RLBench multi-view fused cloud (ragged ~4096 points) -> 2048 tokens, ACTRLBenchPCD head
    "RLB": dict(policy="act_rlbench", batch=8, n_points=4096, pcd_npoints=2048, dtype="bf16", ragged=True),
    # configs[4] as the reference's RLBench Diffusion-Policy experiment really runs it: 512-d task embedding as goal, 11-d
    # action / proprioception, two micro-batches of 16 per optimizer step, ragged fused clouds (mode="hybrid")
    "RLBDP": dict(policy="dp_rlbench", batch=16, n_points=4096, pcd_npoints=2048, dtype="bf16", ragged=True),
    # the `scratch_pointnet_pcd_presample.yaml` experiments at the C2 / REF shapes: SA layer on the raw features, PointNet on the
    # sampled points (policy.pre_sample, act.py:509-530)
    "C2P": dict(policy="act", batch=8, n_points=1024, pcd_npoints=512, dtype="bf16", ragged=False, pre_sample=True),
    "REFP": dict(policy="act", batch=8, n_points=4096, pcd_npoints=2048, dtype="bf16", ragged=True, pre_sample=True),
    # C2 with ragged clouds: the headline shape as real data delivers it (mode="hybrid")
    "C2R": dict(policy="act", batch=8, n_points=1024, pcd_npoints=512, dtype="bf16", ragged=True),
    # C3 with ragged clouds (what GridSamplePCD really delivers): exercises mode="hybrid" for the Diffusion-Policy trainer
    "C3R": dict(policy="dp", batch=64, n_points=1024, pcd_npoints=512, dtype="bf16", ragged=True),
}
