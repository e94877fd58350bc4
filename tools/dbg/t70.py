import sys, torch
sys.path.insert(0, "/root/repo")
from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, build_dp_policy, build_rlbench_act_policy, clone_batch, make_act_batch, make_dp_batch, DP_OPTIM
dev = torch.device("cuda:0")
for kind in ("act", "dp", "rlb"):
    torch.manual_seed(0)
    if kind == "dp":
        pol = build_dp_policy(pcd_npoints=64, sa_impl="fused", down_dims=(32, 64, 128)).to(dev); optim = dict(DP_OPTIM); b = make_dp_batch(2, 256, device=dev)
    elif kind == "rlb":
        pol = build_rlbench_act_policy(pcd_npoints=64, sa_impl="fused", num_encoder_layers=1, num_decoder_layers=1).to(dev); optim = dict(accumulate_grad_batches=1)
        b = make_act_batch(2, 256, device=dev, action_dim=11, qpos_dim=11, goal_cond_dim=512)
    else:
        pol = build_act_policy(pcd_npoints=64, sa_impl="fused", num_encoder_layers=1, num_decoder_layers=1).to(dev); optim = dict(accumulate_grad_batches=1); b = make_act_batch(2, 256, device=dev)
    for mode in ("graph", "hybrid"):
        tr = BCTrainer(pol, total_steps=200, precision="bf16", device=dev, mode=mode, optim=optim)
        for i in range(70):
            tr.training_step(clone_batch(b))
        print(kind, mode, "ok", tr.metrics())
