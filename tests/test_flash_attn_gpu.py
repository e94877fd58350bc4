"""csrc/attn_flash.hip (MFMA attention for long query sets: the encoder's self-attention over the point tokens) against an
fp32 restatement and against the framework's fp32 scaled_dot_product_attention, forward and backward, at the three token
counts of the BASELINE configs (515 / 1027 / 2051 = M + 3), with key-padding masks, the packed q|k views of the
in-projection, ragged lengths around the 64 / 128 tile edges, dropout re-derived from the counter hash, and run-to-run
bit reproducibility (the backward has no atomics).  Reference semantics:
/root/reference/src/models/components/act/transformer.py:221,244-262 (nn.MultiheadAttention)."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.test_small_attn_gpu import DEV, keep_mask, reference, run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S", [515, 1027, 2051])
@pytest.mark.parametrize("masked", [False, True])
def test_encoder_self_attention_shapes_match_fp32(S, masked):
    """B x 8 heads x S x S as the ACT encoder runs it (q | k packed side by side, dq | dk written into the same layout)."""
    run_case(2 if S > 1027 else 3, 8, S, S, masked, packed=True)


@pytest.mark.parametrize("B,H,L,S", [(1, 2, 129, 40), (1, 1, 300, 257), (2, 3, 130, 64), (1, 2, 192, 63), (1, 1, 257, 129),
                                     (2, 2, 640, 100), (1, 8, 1027, 130), (1, 1, 131, 1),
                                     # remainders of exactly 32 / 33 rows (rem mode on / off), a whole problem inside one remainder block
                                     (1, 2, 160, 160), (1, 1, 161, 288), (2, 2, 32, 20), (1, 2, 2051, 2051)])
@pytest.mark.parametrize("masked", [False, True])
def test_ragged_lengths_around_the_tile_edges(B, H, L, S, masked):
    run_case(B, H, L, S, masked, packed=False)


@pytest.mark.parametrize("L,S", [(515, 515), (300, 1027)])
def test_flash_attention_dropout_uses_the_counter_hash_consistently(L, S):
    run_case(2, 8, L, S, False, packed=False, p=0.1)


def test_flash_matches_the_framework_sdpa_in_fp32():
    """The same call through torch's scaled_dot_product_attention in fp32 (what nn.MultiheadAttention runs): forward and
    all three gradients at bf16 resolution."""
    from pointcloudmatters_amd.policy import small_attn

    torch.manual_seed(7)
    B, H, S = 2, 8, 1027
    q, k, v = (torch.randn(B, S, H * 64, device=DEV).bfloat16().requires_grad_(True) for _ in range(3))
    kpm = torch.zeros(B, S, dtype=torch.bool, device=DEV)
    kpm[1, 900:] = True
    g = torch.randn(B, S, H * 64, device=DEV).bfloat16()
    out = small_attn.small_attention(q, k, v, kpm, H, 0.0)
    heads = lambda t: t.float().view(B, S, H, 64).transpose(1, 2)  # noqa: E731
    want = F.scaled_dot_product_attention(heads(q), heads(k), heads(v), attn_mask=(~kpm)[:, None, None, :]).transpose(1, 2).reshape(B, S, H * 64)
    assert (out.float() - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 2e-2
    got = torch.autograd.grad(out, (q, k, v), g)
    exp = torch.autograd.grad(want, (q, k, v), g.float())
    for a, r in zip(got, exp):
        assert (a.float() - r.float()).norm().item() <= 1.5e-2 * r.float().norm().item()  # bf16 P / dS and bf16 outputs
        assert (a.float() - r.float()).abs().max().item() <= 3e-2 * r.float().abs().max().item() + 1e-2


def test_fully_masked_rows_and_batches_give_zeros_not_nans():
    """A batch entry whose keys are ALL padded: output 0, lse +inf, gradients 0 (the framework returns NaN there)."""
    from pointcloudmatters_amd.policy import small_attn

    torch.manual_seed(1)
    B, H, S = 2, 2, 200
    q, k, v = (torch.randn(B, S, H * 64, device=DEV).bfloat16().requires_grad_(True) for _ in range(3))
    kpm = torch.zeros(B, S, dtype=torch.bool, device=DEV)
    kpm[1] = True
    out = small_attn.small_attention(q, k, v, kpm, H, 0.0)
    assert torch.isfinite(out).all() and torch.count_nonzero(out[1]) == 0
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), torch.randn_like(out))
    for t in (dq, dk, dv):
        assert torch.isfinite(t).all() and torch.count_nonzero(t[1]) == 0
    want = reference(q[:1], k[:1], v[:1], None, H)
    assert (out[:1].float() - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 2e-2


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_flash_attention_is_bit_reproducible(p):
    from pointcloudmatters_amd.policy import fused_ops, small_attn

    torch.manual_seed(5)
    B, H, S = 2, 8, 515
    q, k, v = (torch.randn(B, S, H * 64, device=DEV).bfloat16().requires_grad_(True) for _ in range(3))
    g = torch.randn(B, S, H * 64, device=DEV).bfloat16()
    runs = []
    for _ in range(2):
        ctx = fused_ops.FusedContext(DEV)
        ctx.set_step(11)
        with fused_ops.activate(ctx):
            out = small_attn.small_attention(q, k, v, None, H, p)
        runs.append((out.detach().clone(),) + tuple(t.clone() for t in torch.autograd.grad(out, (q, k, v), g)))
    for a, b in zip(*runs):
        assert torch.equal(a, b)


def test_dropout_statistics_and_scaling_at_long_lengths():
    """keep rate and the 1 / (1 - p) rescale at 1027 x 1027: E[out] under dropout == the no-dropout output."""
    from pointcloudmatters_amd.policy import fused_ops, small_attn

    torch.manual_seed(2)
    B, H, S, p = 1, 2, 1027, 0.25
    q, k, v = (torch.randn(B, S, H * 64, device=DEV).bfloat16() for _ in range(3))
    ctx = fused_ops.FusedContext(DEV)
    ctx.set_step(1)
    with fused_ops.activate(ctx):
        out = small_attn.small_attention(q, k, v, None, H, p)
    keep = keep_mask(int(ctx.seed.item()), 1, B, H, S, S, p).float()
    assert abs(keep.mean().item() - (1 - p)) < 5e-3
    want = reference(q, k, v, None, H, keep, p)
    assert (out.float() - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 2e-2
    assert math.isfinite(out.float().sum().item())
