"""GPU: the feed-forward sub-layer on the matrix cores (csrc/ffn_mfma.hip, the bf16-autocast path) against

  * an fp32 framework restatement of LayerNorm(x + dropout(linear2(dropout(relu(linear1(x)))))) with the SAME roundings the
    autocast recipe has (bf16 operands, h and y rounded to bf16 as they leave their GEMMs; /root/reference/src/models/components/
    act/transformer.py:253-256, 342-345): outputs within 1e-4, gradients within 2e-3 of their largest magnitude (dy and dh enter
    the backward products as bf16, so a gradient carries one more bf16 rounding than a forward value);
  * the fp32 kernel (csrc/ffn.hip) on the same inputs: values at bf16 distance; with dropout on, the masks the backward re-derives
    from the counter hash are the ones the forward applied.
Row counts cover tiles with tails (rows % 32 != 0), one-row inputs, the decoder's 800 and the encoder's 4120 rows."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _enable_mfma_path(monkeypatch):
    """The matrix-core path is opt-in in the product (PCM_FFN_MFMA, see policy/fused_ops.py for the measurements): on for these tests."""
    from pointcloudmatters_amd.policy import fused_ops

    monkeypatch.setattr(fused_ops, "FFN_MFMA", True)
    monkeypatch.setattr(fused_ops, "FFN_MFMA_MIN_ROWS", 1)


def _bf(t):
    return t.to(torch.bfloat16).float()


def _reference(x, l1, l2, norm):
    """fp32 evaluation with the autocast recipe's roundings made explicit (straight-through: rounding has gradient 1)."""
    def rnd(t):
        return t + (_bf(t) - t).detach()

    h = rnd(torch.nn.functional.linear(rnd(x), rnd(l1.weight)) + l1.bias)  # bias added in the GEMM's fp32 epilogue, then bf16
    h = torch.relu(h)
    y = rnd(torch.nn.functional.linear(h, rnd(l2.weight)) + l2.bias)
    return norm(x + y)


def _modules(E, dev, seed=1):
    torch.manual_seed(seed)
    l1, l2, norm = nn.Linear(E, 32).to(dev), nn.Linear(32, E).to(dev), nn.LayerNorm(E).to(dev)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
        l1.bias.uniform_(-0.3, 0.3)
        l2.bias.uniform_(-0.3, 0.3)
    return l1, l2, norm


def _fused(x, l1, l2, norm, drop_a, drop_b, ctx, bf16=True, n_out=1):
    from pointcloudmatters_amd.policy import fused_ops

    with fused_ops.activate(ctx), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        assert fused_ops.ffn_ln_supported(x, l1, l2, norm)
        return fused_ops.ffn_ln(x, l1, l2, norm, drop_a, drop_b, n_out=n_out)


@pytest.mark.parametrize("E,rows", [(512, (5, 103)), (256, (3, 40)), (512, (1, 1)), (512, (8, 100)), (512, (8, 515)), (256, (2, 33)), (512, (1, 32))])
def test_ffn_mfma_matches_the_autocast_recipe_in_fp32(hip_device, E, rows):
    from pointcloudmatters_amd.policy import fused_ops

    l1, l2, norm = _modules(E, hip_device)
    x = torch.randn(*rows, E, device=hip_device, requires_grad=True)
    params = (x, l1.weight, l1.bias, l2.weight, l2.bias, norm.weight, norm.bias)
    ref = _reference(x, l1, l2, norm)
    gout = torch.randn_like(ref)
    want = torch.autograd.grad(ref, params, gout)
    seen = []
    from pointcloudmatters_amd import _lib

    orig = _lib.check
    _lib.check = lambda rc, what, *a, **k: (seen.append(what), orig(rc, what, *a, **k))[1]
    try:
        out = _fused(x, l1, l2, norm, nn.Dropout(0.0), nn.Dropout(0.0), fused_ops.FusedContext(hip_device))
        got = torch.autograd.grad(out, params, gout)
    finally:
        _lib.check = orig
    assert "pcm_ffn_ln_mfma_forward_hip" in seen and "pcm_ffn_ln_mfma_backward_hip" in seen, seen
    assert out.dtype == torch.float32
    # a pre-activation that sits within one bf16 ulp of a rounding boundary may round the other way in the kernel (fp32 sums in
    # another order): such an element moves by one bf16 step (4e-3 relative) -- allow a handful, hold the rest to 1e-4
    err = (out - ref).abs() / (ref.abs() + 1.0)
    assert (err > 1e-4).float().mean().item() < 2e-3 and err.max().item() < 2e-2, (err.max().item(), (err > 1e-4).float().mean().item())
    for g, w, name in zip(got, want, ("x", "w1", "b1", "w2", "b2", "gamma", "beta")):
        scale = w.abs().max().item() + 1e-9
        tol = 6e-3 if name in ("x", "w1", "b1") else 2e-3  # dh = bf16(dy) W2 rounded to bf16 once more before W1
        assert (g - w).abs().max().item() <= tol * scale + 1e-6, (name, (g - w).abs().max().item(), scale)


@pytest.mark.parametrize("E,R", [(512, 800), (512, 77), (256, 200)])
def test_ffn_mfma_dropout_masks_agree_between_forward_and_backward(hip_device, E, R):
    """With dropout on: the masks are read off the forward's own outputs (hd > 0: unit active and kept; s != x: channel kept), a
    framework evaluation with THOSE masks gives the reference gradients, and the backward kernel -- which re-derives the masks from
    the counter hash -- must reproduce them: dx, dy, dh and the four column sums.  Two output-gradient addends (dout2)."""
    from pointcloudmatters_amd import _lib
    from pointcloudmatters_amd._lib import raw_stream

    L = _lib.load()
    torch.manual_seed(4)
    f32 = dict(dtype=torch.float32, device=hip_device)
    l1, l2, norm = _modules(E, hip_device, seed=5)
    x = torch.randn(R, E, **f32)
    seed = torch.full((1,), 1234567, dtype=torch.int64, device=hip_device)
    pa, pb = 0.1, 0.1
    hd, s, out = torch.empty(R, 32, **f32), torch.empty(R, E, **f32), torch.empty(R, E, **f32)
    mean, rstd = torch.empty(R, **f32), torch.empty(R, **f32)
    rc = L.pcm_ffn_ln_mfma_forward_hip(R, E, 32, x.data_ptr(), l1.weight.data_ptr(), l1.bias.data_ptr(), l2.weight.data_ptr(), l2.bias.data_ptr(),
                                       norm.weight.data_ptr(), norm.bias.data_ptr(), 1e-5, pa, pb, seed.data_ptr(), 11, 12, hd.data_ptr(),
                                       s.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 0, 0, 0, 0, raw_stream())
    assert rc == 0
    ma, mb = hd > 0, (s - x) != 0
    assert abs(mb.float().mean().item() - (1 - pb)) < 0.01  # kept share of the output channels
    h_pre = _bf(torch.nn.functional.linear(_bf(x), _bf(l1.weight)) + l1.bias)
    active = h_pre > 0
    assert abs((ma & active).float().sum().item() / active.float().sum().item() - (1 - pa)) < 0.02  # kept share of the active units
    # reference with these masks (straight-through roundings as in _reference)
    xr = x.clone().requires_grad_(True)
    w1, b1, w2, b2, g, bt = (t.detach().clone().requires_grad_(True) for t in (l1.weight, l1.bias, l2.weight, l2.bias, norm.weight, norm.bias))

    def rnd(t):
        return t + (_bf(t) - t).detach()

    h = rnd(torch.nn.functional.linear(rnd(xr), rnd(w1)) + b1)
    hdr = rnd(torch.relu(h) * ma.float() / (1 - pa))
    y = rnd(torch.nn.functional.linear(hdr, rnd(w2)) + b2)
    sr = xr + rnd(y / (1 - pb)) * mb.float()
    outr = torch.nn.functional.layer_norm(sr, (E,), g, bt, 1e-5)
    assert (outr.detach() - out).abs().max().item() < 2e-2 and ((outr.detach() - out).abs() > 1e-3).float().mean().item() < 5e-3
    g1, g2 = torch.randn(R, E, **f32), torch.randn(R, E, **f32)
    want = torch.autograd.grad(outr, (xr, w1, b1, w2, b2, g, bt), g1 + g2)
    dx, dy, dh = torch.empty(R, E, **f32), torch.empty(R, E, **f32), torch.empty(R, 32, **f32)
    pw = 3 * E + 32
    part, sums = torch.empty(L.pcm_ffn_ln_mfma_blocks(R) * pw, **f32), torch.empty(pw, **f32)
    rc = L.pcm_ffn_ln_mfma_backward_hip(R, E, 32, g1.data_ptr(), g2.data_ptr(), x.data_ptr(), s.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                        hd.data_ptr(), l1.weight.data_ptr(), l2.weight.data_ptr(), norm.weight.data_ptr(), pa, pb, seed.data_ptr(),
                                        12, dx.data_ptr(), dy.data_ptr(), dh.data_ptr(), part.data_ptr(), sums.data_ptr(), raw_stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(dy == 0, ~mb)  # the backward dropped exactly the channels the forward dropped
    assert bool((dh[~ma] == 0).all())
    got = (dx, dh.t() @ x, sums[3 * E:], dy.t() @ hd, sums[2 * E:3 * E], sums[:E], sums[E:2 * E])
    for gk, wk, name in zip(got, want, ("x", "w1", "b1", "w2", "b2", "gamma", "beta")):
        rel = ((gk - wk).norm() / (wk.norm() + 1e-12)).item()
        assert rel < 1e-2, (name, rel)


def test_ffn_mfma_and_fp32_kernel_agree_without_dropout(hip_device):
    """The matrix-core path against the fp32 kernel of csrc/ffn.hip through the autograd node, two consumers (n_out = 2): values at
    bf16 distance.  A hidden pre-activation within a bf16 ulp of zero sits on the other side of the relu, which moves one row's
    gradient by a whole weight column -- rare, so the comparison is in the L2 norm."""
    from pointcloudmatters_amd.policy import fused_ops

    E = 512
    l1, l2, norm = _modules(E, hip_device, seed=3)
    x = torch.randn(8, 100, E, device=hip_device, requires_grad=True)
    params = (x, l1.weight, l1.bias, l2.weight, l2.bias, norm.weight, norm.bias)
    res = {}
    for bf16 in (True, False):
        o1, o2 = _fused(x, l1, l2, norm, nn.Dropout(0.0), nn.Dropout(0.0), fused_ops.FusedContext(hip_device), bf16=bf16, n_out=2)
        g1 = torch.randn(o1.shape, device=hip_device, generator=torch.Generator(hip_device).manual_seed(5))
        g2 = torch.randn(o1.shape, device=hip_device, generator=torch.Generator(hip_device).manual_seed(6))
        res[bf16] = (o1.detach(), torch.autograd.grad((o1, o2), params, (g1, g2)))
    om, of = res[True][0], res[False][0]
    assert ((om - of).norm() / of.norm()).item() < 1e-2
    for gm, gf, name in zip(res[True][1], res[False][1], ("x", "w1", "b1", "w2", "b2", "gamma", "beta")):
        rel = ((gm - gf).norm() / (gf.norm() + 1e-12)).item()
        assert rel <= 6e-2, (name, rel)  # w1 / b1 / x sit behind the relu gates: the most flip-sensitive


def test_ffn_mfma_is_bit_reproducible_and_emits_the_next_layers_operands(hip_device):
    from pointcloudmatters_amd import _lib
    from pointcloudmatters_amd._lib import raw_stream

    L = _lib.load()
    E, R = 512, 2 * 131
    torch.manual_seed(0)
    f32 = dict(dtype=torch.float32, device=hip_device)
    x, pos = torch.randn(R, E, **f32), torch.randn(131, E, **f32)
    w1, b1, w2, b2 = torch.randn(32, E, **f32) * 0.05, torch.randn(32, **f32) * 0.1, torch.randn(E, 32, **f32) * 0.05, torch.randn(E, **f32) * 0.1
    g, bt = torch.rand(E, **f32) + 0.5, torch.randn(E, **f32) * 0.1
    seed = torch.full((1,), 99, dtype=torch.int64, device=hip_device)
    outs = []
    for _ in range(2):
        hd, s, out = torch.empty(R, 32, **f32), torch.empty(R, E, **f32), torch.empty(R, E, **f32)
        mean, rstd = torch.empty(R, **f32), torch.empty(R, **f32)
        s16, o16 = torch.empty(R, E, dtype=torch.bfloat16, device=hip_device), torch.empty(R, E, dtype=torch.bfloat16, device=hip_device)
        rc = L.pcm_ffn_ln_mfma_forward_hip(R, E, 32, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), g.data_ptr(),
                                           bt.data_ptr(), 1e-5, 0.1, 0.1, seed.data_ptr(), 3, 4, hd.data_ptr(), s.data_ptr(), out.data_ptr(),
                                           mean.data_ptr(), rstd.data_ptr(), pos.data_ptr(), pos.numel(), s16.data_ptr(), o16.data_ptr(),
                                           raw_stream())
        assert rc == 0
        outs.append((hd, s, out, mean, rstd, s16, o16))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    hd, s, out, mean, rstd, s16, o16 = outs[0]
    assert torch.equal(o16, out.to(torch.bfloat16)) and torch.equal(s16, (out.view(2, 131, E) + pos).view(R, E).to(torch.bfloat16))
    torch.testing.assert_close(mean, s.mean(dim=1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rstd, torch.rsqrt(s.var(dim=1, unbiased=False) + 1e-5), rtol=1e-4, atol=1e-5)
    assert 0.05 < (hd == 0).float().mean().item() < 0.75  # relu + dropout zeros
