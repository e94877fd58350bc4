"""bc/configs.py (frozen dicts: Hydra is absent on the GPU box) against the reference's own YAML files.

Runs only where /root/reference exists (the build container); skipped on the GPU box.  Every value the frozen dicts carry is
compared with the YAML it cites; `${...}` interpolations are resolved by hand for the few keys that use them.
"""
import os

import pytest

REF = "/root/reference/configs"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (build container only)")


def _y(rel):
    import yaml

    with open(os.path.join(REF, rel)) as f:
        return yaml.safe_load(f)


def _optim_matches(opt, model_yaml, trainer_yaml, exp_yaml):
    o, s = model_yaml["optimizer"], model_yaml["lr_scheduler"]["scheduler"]
    assert o["type"] == "AdamW" and s["type"] == "OneCycleLR" and s["anneal_strategy"] == "cos"
    assert s["max_lr"] == "${model.optimizer.lr}"
    assert opt["lr"] == o["lr"] and opt["weight_decay"] == o["weight_decay"]
    # `yaml_betas`: what the YAML asks for where the reference's builder does not forward it (bc/configs.py DP_OPTIM, tests/test_optim_ref.py)
    assert tuple(opt.get("yaml_betas", opt.get("betas", (0.9, 0.999)))) == tuple(o.get("betas", (0.9, 0.999)))
    assert opt["pct_start"] == s["pct_start"] and opt["div_factor"] == s["div_factor"] and opt["final_div_factor"] == s["final_div_factor"]
    assert opt["gradient_clip_val"] == trainer_yaml["gradient_clip_val"]
    want_acc = (exp_yaml.get("trainer") or {}).get("accumulate_grad_batches", trainer_yaml["accumulate_grad_batches"])
    assert opt["accumulate_grad_batches"] == want_acc


def _transformer_matches(c, pol):
    t, e = pol["transformer"], pol["encoder"]
    for k in ("dropout", "nhead", "dim_feedforward", "num_encoder_layers", "num_decoder_layers", "normalize_before", "return_intermediate_dec"):
        assert c[k] == t[k], k
    assert t["d_model"] == "${model.policy.hidden_dim}" and e["d_model"] == "${model.policy.hidden_dim}"
    assert e["num_layers"] == "${model.policy.transformer.num_encoder_layers}" and e["activation"] == "relu"
    for k in ("hidden_dim", "latent_dim", "kl_weight", "pcd_nsample"):
        assert c[k] == pol[k], k
    assert pol["action_loss"] == {"_target_": "torch.nn.MSELoss", "reduction": "none"}


def test_maniskill2_act_config():
    from pointcloudmatters_amd.bc.configs import ACT_MODEL, ACT_OPTIM, WORKLOADS

    model = _y("model/maniskill2_act_pcd_model.yaml")
    exp = _y("exp_maniskill2_act_policy/maniskill2_model/scratch_pointnet_pcd.yaml")
    task = _y("exp_maniskill2_act_policy/maniskill2_pcd_task/PickCube-v0.yaml")
    data = _y("data/maniskill2_act_pcd_dataset.yaml")
    ddp = _y("trainer/ddp.yaml")
    pol = model["policy"]
    assert pol["_target_"].endswith("act.ACTPCD")
    _transformer_matches(ACT_MODEL, pol)
    assert pol["num_queries"] == "${data.train.chunk_size}" and ACT_MODEL["num_queries"] == data["train"]["chunk_size"]
    for k in ("action_dim", "qpos_dim", "goal_cond_dim"):
        assert ACT_MODEL[k] == task["model"]["policy"][k], k
    bb = exp["model"]["policy"]["backbone"]
    assert bb["_target_"].endswith("pointnet.PointNet") and bb["num_classes"] == 0 and ACT_MODEL["in_channels"] == bb["in_channels"]
    _optim_matches(ACT_OPTIM, model, ddp, exp)
    assert ddp["sync_batchnorm"] is True and ddp["precision"] == "32-true" and ddp["strategy"] == "ddp"
    # the shipped shape: batch 8 (scratch_pointnet_pcd.yaml:10), 2048 tokens (maniskill2_act_pcd_model.yaml:68)
    assert WORKLOADS["REF"]["batch"] == exp["data"]["batch_size_train"] and WORKLOADS["REF"]["pcd_npoints"] == pol["pcd_npoints"]


@pytest.mark.parametrize("variant,cin", [("", 6), ("_wo_rgb", 3), ("_wo_xyz", 3)])
def test_presample_experiment_files(variant, cin):
    """The six PointNet `*_presample*` experiment files: what they set is what build_act_policy / build_dp_policy accept."""
    from pointcloudmatters_amd.bc import build_act_policy, build_dp_policy

    act = _y(f"exp_maniskill2_act_policy/maniskill2_model/scratch_pointnet_pcd_presample{variant}.yaml")["model"]["policy"]
    assert act["pre_sample"] is True and act["backbone"]["in_channels"] == cin and act["backbone"]["num_classes"] == 0
    pol = build_act_policy(pcd_npoints=2048, pre_sample=act["pre_sample"], in_channels=act["backbone"]["in_channels"], pointops=object())
    assert tuple(pol.linear.weight.shape) == (cin, 3 + cin) and pol.bn.num_features == cin and pol.backbone.num_channels == 512
    dp = _y(f"exp_maniskill2_diffusion_policy/maniskill2_model/scratch_pointnet_pcd_presample{variant}.yaml")["model"]["policy"]
    enc = dp["obs_encoder"]
    assert enc["pre_sample"] is True and enc["in_channel"] == cin and enc["pcd_model"]["in_channels"] == cin
    assert dp["shape_meta"]["obs"]["pcds"]["shape"] == [cin]
    pol = build_dp_policy(pcd_npoints=enc["pcd_npoints"], pre_sample=True, in_channels=cin, pcd_num_classes=enc["pcd_model"]["num_classes"],
                          pcd_hidden_dim=enc["pcd_hidden_dim"], projector_layers=enc["projector_layers"],
                          projector_channels=tuple(enc["projector_channels"]), down_dims=(32, 64), pointops=object())
    e = pol.obs_encoder
    assert tuple(e.linear.weight.shape) == (cin, 3 + cin) and e.projector[0].in_channels == enc["pcd_model"]["num_classes"]


def test_maniskill2_diffusion_policy_config():
    from pointcloudmatters_amd.bc.configs import DP_MODEL, DP_OPTIM, WORKLOADS

    model = _y("model/maniskill2_diffusion_policy_model.yaml")
    exp = _y("exp_maniskill2_diffusion_policy/maniskill2_model/scratch_pointnet_pcd.yaml")
    task = _y("exp_maniskill2_diffusion_policy/maniskill2_pcd_task/StackCube-v0.yaml")
    pol, enc = model["policy"], exp["model"]["policy"]["obs_encoder"]
    for k in ("n_action_steps", "n_obs_steps", "diffusion_step_embed_dim", "kernel_size", "n_groups", "cond_predict_scale"):
        assert DP_MODEL[k] == pol[k], k
    assert tuple(DP_MODEL["down_dims"]) == tuple(pol["down_dims"])
    ns = pol["noise_scheduler"]
    assert DP_MODEL["num_train_timesteps"] == ns["num_train_timesteps"] == pol["num_inference_steps"]
    assert ns["beta_schedule"] == "squaredcos_cap_v2" and ns["prediction_type"] == "epsilon" and ns["clip_sample"] is True
    assert ns["variance_type"] == "fixed_small"
    assert enc["_target_"].endswith("pcd_obs_encoder.PCDObsEncoder") and enc["share_pcd_model"] is True
    assert DP_MODEL["in_channels"] == enc["pcd_model"]["in_channels"] and DP_MODEL["pcd_num_classes"] == enc["pcd_model"]["num_classes"]
    for k in ("pcd_hidden_dim", "projector_layers", "pcd_nsample"):
        assert DP_MODEL[k] == enc[k], k
    assert tuple(DP_MODEL["projector_channels"]) == tuple(enc["projector_channels"])
    sm = task["model"]["policy"]["shape_meta"]
    assert DP_MODEL["action_dim"] == sm["action"]["shape"][0] and DP_MODEL["qpos_dim"] == sm["obs"]["qpos"]["shape"][0]
    assert pol["horizon"] == "${data.train.chunk_size}"
    assert DP_MODEL["horizon"] == _y("data/maniskill2_diffusion_policy_pcd_dataset.yaml")["train"]["chunk_size"]
    base = _y("exp_maniskill2_diffusion_policy/base.yaml")
    _optim_matches(DP_OPTIM, model, dict(_y("trainer/ddp.yaml"), **{k: v for k, v in (base.get("trainer") or {}).items()
                                                                     if k in ("gradient_clip_val", "accumulate_grad_batches")}), exp)
    assert WORKLOADS["C3"]["batch"] == task["data"]["batch_size_train"]


def test_rlbench_configs():
    from pointcloudmatters_amd.bc.configs import RLBENCH_ACT_MODEL, RLBENCH_ACT_OPTIM, RLBENCH_DP_MODEL, RLBENCH_DP_OPTIM, WORKLOADS

    model = _y("model/rlbench_act_pcd_model.yaml")
    data = _y("data/rlbench_act_pcd_dataset.yaml")["train"]
    exp = _y("exp_rlbench_act_policy/rlbench_model/scratch_pointnet_pcd.yaml")
    pol = model["policy"]
    assert pol["_target_"].endswith("act.ACTRLBenchPCD")
    _transformer_matches(RLBENCH_ACT_MODEL, pol)
    assert RLBENCH_ACT_MODEL["goal_cond_dim"] == pol["goal_cond_dim"] and RLBENCH_ACT_MODEL["position_loss_weight"] == pol["position_loss_weight"]
    assert pol["action_dim"] == "${data.train.action_dim}" == pol["qpos_dim"]
    assert RLBENCH_ACT_MODEL["action_dim"] == RLBENCH_ACT_MODEL["qpos_dim"] == data["action_dim"]
    assert RLBENCH_ACT_MODEL["num_queries"] == data["chunk_size"]
    assert RLBENCH_ACT_MODEL["rot_type"] == data["rot_type"] and RLBENCH_ACT_MODEL["collision"] == data["collision"]
    base = _y("exp_rlbench_act_policy/base.yaml")
    trainer = dict(_y("trainer/ddp.yaml"), **{k: v for k, v in (base.get("trainer") or {}).items()
                                              if k in ("gradient_clip_val", "accumulate_grad_batches")})
    _optim_matches(RLBENCH_ACT_OPTIM, model, trainer, exp)
    assert WORKLOADS["RLB"]["batch"] == exp["data"]["batch_size_train"]
    # Diffusion Policy
    model = _y("model/rlbench_diffusion_policy_model.yaml")
    data = _y("data/rlbench_diffusion_policy_pcd_dataset.yaml")["train"]
    exp = _y("exp_rlbench_diffusion_policy/rlbench_model/scratch_pointnet_pcd.yaml")
    pol = model["policy"]
    assert RLBENCH_DP_MODEL["goal_dim"] == pol["shape_meta"]["goal"]["task_emb"]["shape"][0]
    assert RLBENCH_DP_MODEL["action_dim"] == RLBENCH_DP_MODEL["qpos_dim"] == data["action_dim"] and RLBENCH_DP_MODEL["horizon"] == data["chunk_size"]
    assert tuple(RLBENCH_DP_MODEL["down_dims"]) == tuple(pol["down_dims"])
    base = _y("exp_rlbench_diffusion_policy/base.yaml")
    trainer = dict(_y("trainer/ddp.yaml"), **{k: v for k, v in (base.get("trainer") or {}).items()
                                              if k in ("gradient_clip_val", "accumulate_grad_batches")})
    _optim_matches(RLBENCH_DP_OPTIM, model, trainer, exp)
    assert WORKLOADS["RLBDP"]["batch"] == exp["data"]["batch_size_train"]
