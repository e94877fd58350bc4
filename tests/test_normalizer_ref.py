"""CPU: policy/diffusion.LinearNormalizer against tests/golden/normalizer_ref.npz, produced by the reference's own LinearNormalizer
(src/utils/diffusion_policy/normalizer.py) the two ways the Diffusion-Policy datasets build it -- `get_range_normalizer_from_stat` per key
(src/utils/normalize_utils.py:7-21) and `fit` -- with one constant action dimension (the `ignore_dim` branch)."""
import os

import numpy as np
import torch

FX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "normalizer_ref.npz"))


def _ours(tag):
    from pointcloudmatters_amd.policy.diffusion import LinearNormalizer

    n = LinearNormalizer()
    if tag == "fit":
        n.fit({"action": torch.from_numpy(FX["action"]), "qpos": torch.from_numpy(FX["qpos"])})
    else:
        for k in ("action", "qpos"):
            f = torch.from_numpy(FX[k]).reshape(-1, FX[k].shape[-1])
            n.set_range(k, f.min(0).values, f.max(0).values, f.mean(0), f.std(0))
    return n


def test_state_dict_keys_and_values_equal_the_reference():
    for tag in ("stat", "fit"):
        sd = _ours(tag).state_dict()
        assert sorted(sd) == list(FX[f"{tag}.keys"])
        for k, v in sd.items():
            assert not v.requires_grad
            # bit-equal on the machine that wrote the fixture; the mean / std reductions may round differently under another vector width
            np.testing.assert_allclose(v.numpy(), FX[f"{tag}.sd.{k}"], rtol=1e-6, atol=1e-7, err_msg=f"{tag} {k}")
    # the constant dimension: scale 1, offset -min (normalizer.py:236-242)
    sd = _ours("fit").state_dict()
    assert float(sd["params_dict.action.scale"][6]) == 1.0 and float(sd["params_dict.action.offset"][6]) == -1.0


def test_normalize_and_unnormalize_equal_the_reference():
    for tag in ("stat", "fit"):
        n = _ours(tag)
        out = n.normalize({"action": torch.from_numpy(FX["action"][:5]), "qpos": torch.from_numpy(FX["qpos"][:5])})
        np.testing.assert_array_equal(out["action"].numpy(), FX[f"{tag}.norm.action"])
        np.testing.assert_array_equal(out["qpos"].numpy(), FX[f"{tag}.norm.qpos"])
        np.testing.assert_array_equal(n["action"].unnormalize(torch.from_numpy(FX["y"])).numpy(), FX[f"{tag}.unnorm.action"])


def test_the_reference_state_dict_loads_strictly():
    from pointcloudmatters_amd.policy.diffusion import LinearNormalizer

    ref_sd = {k[len("stat.sd."):]: torch.from_numpy(FX[k]) for k in FX.files if k.startswith("stat.sd.")}
    n = _ours("fit")  # any fitted normaliser with the same keys (set_normalizer's route, diffusion_unet_image_policy.py:225-226)
    n.load_state_dict(ref_sd, strict=True)
    got = n.normalize(torch.from_numpy(FX["qpos"][:5]), "qpos")
    np.testing.assert_array_equal(got.numpy(), FX["stat.norm.qpos"])
