"""csrc/tokens.hip and the self-attention in-projection node (policy/fused_ops._SelfAttnInProj) against framework ops."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_add_cast2_and_add2_cast_are_exact():
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for b, s, e in ((3, 17, 64), (1, 5, 512), (4, 100, 256)):
        x = torch.randn(b, s, e, device=DEV)
        for pos in (torch.randn(b, s, e, device=DEV), torch.randn(1, s, e, device=DEV)):
            q = torch.empty(b, s, e, dtype=torch.bfloat16, device=DEV)
            v = torch.empty_like(q)
            assert L.pcm_add_cast2_hip(x.numel(), pos.numel(), x.data_ptr(), pos.data_ptr(), q.data_ptr(), v.data_ptr(), st) == 0
            assert torch.equal(q, (x + pos).bfloat16()) and torch.equal(v, x.bfloat16())
        a, c = torch.randn(b * s, e, device=DEV).bfloat16(), torch.randn(b * s, e, device=DEV).bfloat16()
        out = torch.empty(b * s, e, device=DEV)
        assert L.pcm_add2_cast_hip(out.numel(), a.data_ptr(), c.data_ptr(), out.data_ptr(), st) == 0
        assert torch.equal(out, a.float() + c.float())
    assert L.pcm_add_cast2_hip(6, 4, 0, 0, 0, 0, st) != 0  # not a multiple of 4 / pos does not divide n


@pytest.mark.parametrize("rows,c", [(4120, 512), (800, 512), (37, 64), (5000, 1024), (8, 4)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_colsum_matches_fp64_sum(rows, c, dt):
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    qk = torch.randn(rows, 2 * c, device=DEV).to(dt)
    v = torch.randn(rows, c, device=DEV).to(dt)
    partial = torch.empty(L.pcm_colsum_slots(rows, c) * 3 * c, device=DEV)
    for out_dt in (torch.float32, torch.bfloat16):
        out = torch.empty(3 * c, dtype=out_dt, device=DEV)
        rc = L.pcm_colsum_hip(rows, c, 3, int(dt == torch.bfloat16), qk.data_ptr(), 2 * c, qk.data_ptr() + c * qk.element_size(),
                              2 * c, v.data_ptr(), c, partial.data_ptr(), int(out_dt == torch.bfloat16), out.data_ptr(), st)
        assert rc == 0
        want = torch.cat([qk.double().sum(0), v.double().sum(0)])
        tol = 1e-5 if out_dt == torch.float32 else 1e-2
        torch.testing.assert_close(out.double(), want, rtol=tol, atol=tol * (rows ** 0.5))
        again = torch.empty_like(out)
        L.pcm_colsum_hip(rows, c, 3, int(dt == torch.bfloat16), qk.data_ptr(), 2 * c, qk.data_ptr() + c * qk.element_size(), 2 * c,
                         v.data_ptr(), c, partial.data_ptr(), int(out_dt == torch.bfloat16), again.data_ptr(), st)
        assert torch.equal(out, again)  # fixed reduction order


@pytest.mark.parametrize("pos_batch", [1, 3])
@pytest.mark.parametrize("pos_grad", [False, True])
def test_self_attn_in_proj_node_matches_framework_chain(pos_batch, pos_grad):
    from pointcloudmatters_amd.policy import fused_ops

    torch.manual_seed(0)
    b, s, e = 3, 700, 256  # 2100 rows: the split-K weight-gradient path is taken
    mha = nn.MultiheadAttention(e, 4).to(DEV)
    w16 = mha.in_proj_weight.detach().bfloat16().requires_grad_(True)
    b16 = mha.in_proj_bias.detach().bfloat16().requires_grad_(True)
    x = torch.randn(b, s, e, device=DEV, requires_grad=True)
    pos = torch.randn(pos_batch, s, e, device=DEV, requires_grad=pos_grad)
    gq, gk, gv = (torch.randn(b, s, e, device=DEV).bfloat16() for _ in range(3))

    def ref():
        qk_in = (x + pos).bfloat16()
        q, k = F.linear(qk_in, w16[: 2 * e], b16[: 2 * e]).unflatten(-1, (2, e)).unbind(-2)
        v = F.linear(x.bfloat16(), w16[2 * e:], b16[2 * e:])
        return q, k, v

    with fused_ops.activate(fused_ops.FusedContext(DEV)), torch.autocast("cuda", dtype=torch.bfloat16):
        assert fused_ops.self_attn_in_proj_supported(x, pos, mha)
        got = fused_ops._SelfAttnInProj.apply(x, pos, w16, b16)
    want = ref()
    for a, r in zip(got, want):
        torch.testing.assert_close(a.float(), r.float(), rtol=1e-2, atol=1e-2)
    ins = [x, w16, b16] + ([pos] if pos_grad else [])
    g_got = torch.autograd.grad(got, ins, (gq, gk, gv))
    g_ref = torch.autograd.grad(want, ins, (gq, gk, gv))
    for a, r, name in zip(g_got, g_ref, ("dx", "dw", "db", "dpos")):
        assert a.dtype == r.dtype and a.shape == r.shape, name
        scale = r.float().abs().max().item()
        assert (a.float() - r.float()).abs().max().item() <= 2e-2 * scale + 1e-3, name
