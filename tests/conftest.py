import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


# Run order (matters under `pytest -x`, which the round-end GPU run uses): evidence first.  Tier 0 = parity against the
# oracle / the reference-generated fixtures, tier 1 = one kernel against a framework (fp32 / fp64) restatement, tier 2 =
# equivalence of execution modes (flat / graph / hybrid), resume, multi-process.  A failure in a later tier can no longer
# hide the parity results of an earlier one.  Inside a tier the collection order is kept.
_FILE_TIER = {
    "test_oracle": 0, "test_golden_cpu": 0, "test_capi": 0, "test_gridsample_cpu": 0, "test_rollout_cpu": 0, "test_rlbench_cpu": 0,
    "test_pointops_gpu": 0, "test_pointops_fuzz_gpu": 0, "test_pointops_misc_gpu": 0, "test_segsum_gpu": 0, "test_gridsample_gpu": 0,
    "test_rollout_gpu": 0, "test_rlbench_gpu": 0,
    "test_sa_fused_gpu": 1, "test_bn_relu_gpu": 1, "test_drln_gpu": 1, "test_tokens_gpu": 1, "test_small_attn_gpu": 1,
    "test_flash_attn_gpu": 1, "test_rows_linear_gpu": 1, "test_unet_ops_gpu": 1, "test_pointnet2_gpu": 1, "test_graphs_gpu": 1,
    "test_host_logic": 1,
    "test_policy_gpu": 2, "test_sync_bn_gpu": 2, "test_hybrid_two_ranks_gpu": 2, "test_bench_multirank_gpu": 2, "test_ddp_gloo": 2,
    "test_determinism_gpu": 2,
}
_TIER0_NAMES = ("matches_reference", "matches_cpu_oracle", "vs_torch")  # fixture / oracle parity tests inside test_policy_gpu.py


def _tier(item):
    mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    tier = _FILE_TIER.get(mod, 1)
    if mod == "test_policy_gpu" and any(s in item.name for s in _TIER0_NAMES):
        tier = 0
    return tier


def pytest_collection_modifyitems(config, items):
    order = {id(it): i for i, it in enumerate(items)}
    items.sort(key=lambda it: (_tier(it), order[id(it)]))
