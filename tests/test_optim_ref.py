"""CPU: `configure_optimizers` against tests/golden/optim_ref.npz, which the REFERENCE's own builders produced
(src/utils/optimizer.py build_optimizer / build_optimizer_v2 + src/utils/scheduler.py build_scheduler with the shipped YAML values; generator:
tests/golden/make_golden.py::golden_optim): which parameters are decayed, with what hyper-parameters, and the learning rate / beta1 in
effect at every step of a 200-step one-cycle schedule."""
import os

import numpy as np
import pytest

from tests.util import optimizer_zoo

FX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim_ref.npz"))


def test_act_recipe_is_one_group_with_every_trainable_parameter_decayed():
    # build_optimizer(cfg, policy, None) -> model.parameters() (optimizer.py:33-37): torch.optim.AdamW keeps frozen ones in the group too
    zoo = optimizer_zoo()
    assert int(FX["act.n_groups"]) == 1
    assert list(FX["act.group0.names"]) == [n for n, _ in zoo.named_parameters()]
    from pointcloudmatters_amd.bc.configs import ACT_OPTIM

    assert float(FX["act.group0.weight_decay"]) == ACT_OPTIM["weight_decay"] == 0.05
    assert not ACT_OPTIM.get("filter_bias_and_bn", False)  # the trainer then builds ONE decayed group (bc/trainer.py)
    assert float(FX["act.group0.eps"]) == 1e-8 and float(FX["act.group0.beta2"]) == 0.999


def test_dp_recipe_grouping_equals_the_reference_function():
    from pointcloudmatters_amd.bc.configs import DP_OPTIM
    from pointcloudmatters_amd.bc.trainer import undecayed_parameters

    zoo = optimizer_zoo()
    assert int(FX["dp.n_groups"]) == 2
    wd = [float(FX[f"dp.group{g}.weight_decay"]) for g in range(2)]
    assert wd == [0.0, DP_OPTIM["weight_decay"]] and DP_OPTIM["filter_bias_and_bn"]
    # build_optimizer_v2 does not forward the YAML's betas [0.9, 0.95] (optimizer.py:304-318): the reference trains with torch's defaults
    assert float(FX["dp.group0.beta2"]) == float(FX["dp.group1.beta2"]) == DP_OPTIM["betas"][1] == 0.999
    assert DP_OPTIM["yaml_betas"] == (0.9, 0.95)
    assert float(FX["dp.group0.eps"]) == float(FX["dp.group1.eps"]) == 1e-8  # FlatAdamW's default too (bc/flat_optim.py)
    nd = {id(p) for p in undecayed_parameters(zoo)}
    ours_nd = [n for n, p in zoo.named_parameters() if id(p) in nd]
    ours_d = [n for n, p in zoo.named_parameters() if p.requires_grad and id(p) not in nd]
    assert ours_nd == list(FX["dp.group0.names"])
    assert ours_d == list(FX["dp.group1.names"])
    assert not any(n.startswith("frozen.") for n in ours_nd + ours_d)  # optimizer.py:160-161: frozen parameters are in no group
    assert "odd.bias" in ours_nd and "pos_table" in ours_d and "emb.weight" in ours_d and "scale" in ours_nd


@pytest.mark.parametrize("tag,cfg_name", [("act", "ACT_OPTIM"), ("dp", "DP_OPTIM")])
def test_one_cycle_schedule_equals_the_reference_scheduler_at_every_step(tag, cfg_name):
    from pointcloudmatters_amd.bc import configs
    from pointcloudmatters_amd.bc.schedule import OneCycle

    o = getattr(configs, cfg_name)
    lr, b1 = FX[f"{tag}.lr"], FX[f"{tag}.beta1"]
    T = lr.shape[0]
    s = OneCycle(o["lr"], T, o["pct_start"], o["div_factor"], o["final_div_factor"])
    got = np.array([s.at(k) for k in range(T)])
    for g in range(lr.shape[1]):  # every group follows the same cycle (no lr_scale in the shipped configs, scheduler.py:120-121)
        np.testing.assert_allclose(got[:, 0], lr[:, g], rtol=1e-14, atol=0)
        np.testing.assert_allclose(got[:, 1], b1[:, g], rtol=1e-14, atol=0)
    assert lr[0, 0] == pytest.approx(o["lr"] / o["div_factor"]) and lr.max() == pytest.approx(o["lr"])
    assert b1[0, 0] == pytest.approx(0.95) and b1.min() == pytest.approx(0.85, abs=1e-4)  # beta1 is cycled too (torch default)
