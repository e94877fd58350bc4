"""policy/rows_linear.py (split-K weight gradients) and fused_ops.proj_drln against plain framework ops."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("rows", [2048, 4120, 8192, 5003, 131072 + 17])  # exact splits, and leftovers handled by the tail GEMM
@pytest.mark.parametrize("m,k,bias", [(64, 6, False), (512, 128, True), (96, 512, True)])
def test_linear_rows_fp32_matches_f_linear(rows, m, k, bias):
    from pointcloudmatters_amd.policy.rows_linear import linear_rows

    torch.manual_seed(rows + m)
    x = torch.randn(rows, k, device=DEV, requires_grad=True)
    w = (torch.randn(m, k, device=DEV) * 0.1).requires_grad_(True)
    b = torch.randn(m, device=DEV, requires_grad=True) if bias else None
    y = linear_rows(x, w, b)
    want = F.linear(x, w, b)
    torch.testing.assert_close(y, want, rtol=1e-5, atol=1e-5)
    g = torch.randn_like(want)
    ins = [x, w] + ([b] if bias else [])
    for a, r in zip(torch.autograd.grad(y, ins, g), torch.autograd.grad(want, ins, g)):
        assert (a - r).norm().item() <= 1e-5 * r.norm().item() + 1e-6


def test_linear_rows_autocast_three_dim_and_no_input_grad():
    from pointcloudmatters_amd.policy.rows_linear import linear_rows

    torch.manual_seed(0)
    x = torch.randn(8, 515, 512, device=DEV)  # no grad on the input (first PointNet layer)
    w16 = (torch.randn(1024, 512, device=DEV) * 0.05).bfloat16().requires_grad_(True)  # bf16 shadow weights
    b16 = torch.randn(1024, device=DEV).bfloat16().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = linear_rows(x, w16, b16)
        want = F.linear(x, w16, b16)
    assert y.dtype == torch.bfloat16 and y.shape == (8, 515, 1024)
    torch.testing.assert_close(y.float(), want.float(), rtol=1e-2, atol=1e-2)
    g = torch.randn_like(want)
    gw, gb = torch.autograd.grad(y, (w16, b16), g)
    rw, rb = torch.autograd.grad(want, (w16, b16), g)
    assert gw.dtype == torch.bfloat16 and gb.dtype == torch.bfloat16
    # the split-K sum is accumulated in fp32 and rounded once: at least as close to the fp32 result as the framework's
    exact = g.float().reshape(-1, 1024).t() @ x.reshape(-1, 512)
    assert (gw.float() - exact).norm() <= 1.05 * (rw.float() - exact).norm() + 1e-3
    torch.testing.assert_close(gb.float(), rb.float(), rtol=2e-2, atol=2e-1)


def test_linear_rows_small_inputs_fall_back_to_f_linear():
    from pointcloudmatters_amd.policy.rows_linear import MIN_ROWS, linear_rows

    x = torch.randn(MIN_ROWS - 1, 32, device=DEV, requires_grad=True)
    w = torch.randn(16, 32, device=DEV, requires_grad=True)
    y = linear_rows(x, w)
    assert type(y.grad_fn).__name__ != "_LinearRowsBackward"
    xc = torch.randn(4000, 32, requires_grad=True)  # host tensors too
    assert torch.equal(linear_rows(xc, w.cpu()), F.linear(xc, w.cpu()))


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_proj_drln_matches_the_plain_chain(p):
    from pointcloudmatters_amd.policy import fused_ops

    torch.manual_seed(1)
    B, L, E = 4, 600, 512
    proj = nn.Linear(E, E).to(DEV)
    norm = nn.LayerNorm(E).to(DEV)
    drop = nn.Dropout(p)
    a = torch.randn(B, L, E, device=DEV, requires_grad=True)
    x = torch.randn(B, L, E, device=DEV, requires_grad=True)
    g = torch.randn(B, L, E, device=DEV)
    ctx = fused_ops.FusedContext(DEV)
    with fused_ops.activate(ctx):
        out = fused_ops.proj_drln(a, proj, x, norm, drop)
        if p == 0.0:
            want = norm(x + proj(a))
        else:  # same mask: run the un-fused drln on the projection's output at the same call site
            with fused_ops.activate(ctx):
                want = fused_ops.drln(x, proj(a), norm, drop)
    torch.testing.assert_close(out, want, rtol=1e-4, atol=1e-4)
    ins = [a, x, proj.weight, proj.bias, norm.weight, norm.bias]
    for got, ref, name in zip(torch.autograd.grad(out, ins, g), torch.autograd.grad(want, ins, g), "a x W b gamma beta".split()):
        assert (got - ref).norm().item() <= 1e-4 * ref.norm().item() + 1e-5, name


@pytest.mark.parametrize("rows,c,dt,odt", [(800, 512, torch.bfloat16, torch.bfloat16), (4120, 3584, torch.bfloat16, torch.bfloat16),
                                           (4120, 3584, torch.bfloat16, torch.float32), (37, 12, torch.float32, torch.float32),
                                           (2000, 2500, torch.float32, torch.float32), (5, 6, torch.float32, torch.float32)])
def test_bias_grad_column_sums(rows, c, dt, odt):
    """rows_linear.bias_grad (two-stage column sums in <= 1024-column pieces, up to three per launch) against the fp64 sum,
    immediately and with its closing stage left to policy/deferred.py's batch; odd widths fall back to the framework."""
    from pointcloudmatters_amd.policy import deferred
    from pointcloudmatters_amd.policy.rows_linear import bias_grad

    torch.manual_seed(0)
    go = (torch.randn(rows, c, device="cuda") + 0.3).to(dt)
    want = go.double().sum(0)
    tol = 1e-5 if odt == torch.float32 else 8e-3
    got = bias_grad(go, odt)
    assert got.dtype == odt and got.shape == (c,)
    torch.testing.assert_close(got.double(), want, rtol=tol, atol=tol * rows ** 0.5)
    assert deferred.begin()
    try:
        pending = bias_grad(go, odt, defer=True)
        n = deferred.flush()
        assert n >= (1 if c % 4 == 0 else 0)
    finally:
        deferred.end()
    assert torch.equal(pending, got)


def test_pending_results_reach_leaf_parameters_intact():
    """A result that is still pending in policy/deferred.py's window must not be COPIED by AccumulateGrad before it is
    written (it clones a gradient that anything else references -- e.g. a queue's view of it): fp32 leaf parameters,
    4120 rows (split-K weight gradient + two-stage bias sum, both with deferred closing stages)."""
    from pointcloudmatters_amd.policy import deferred
    from pointcloudmatters_amd.policy import rows_linear
    from pointcloudmatters_amd.policy.rows_linear import linear_rows

    torch.manual_seed(0)
    lin = nn.Linear(512, 256).to(DEV)
    x = torch.randn(4120, 512, device=DEV)
    go = torch.randn(4120, 256, device=DEV)
    old = rows_linear.DEFER_MAX_BYTES
    rows_linear.DEFER_MAX_BYTES = 1 << 30  # also leave the split-K closing sum pending
    try:
        assert deferred.begin()
        try:
            y = linear_rows(x, lin.weight, lin.bias)
            with deferred.backward_phase():
                y.backward(go)
            assert deferred.flush() >= 2
        finally:
            deferred.end()
    finally:
        rows_linear.DEFER_MAX_BYTES = old
    gw, gb = lin.weight.grad.clone(), lin.bias.grad.clone()
    torch.testing.assert_close(gb.double(), go.double().sum(0), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(gw.double(), go.double().t() @ x.double(), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("rows,cin,cout", [(8, 256, 1024), (8, 1024, 256), (64, 512, 7), (1, 128, 128)])
def test_rows_linear_module_matches_nn_linear(rows, cin, cout):
    """RowsLinear == nn.Linear: same state-dict keys, same fp32 results / gradients; under bf16 autocast the backward node is the
    library's (its bias gradient comes from the column-sum kernel of csrc/tokens.hip, not the framework's bf16 reduction)."""
    from pointcloudmatters_amd.policy.rows_linear import RowsLinear

    torch.manual_seed(rows + cout)
    ours, ref = RowsLinear(cin, cout).to(DEV), nn.Linear(cin, cout).to(DEV)
    assert list(ours.state_dict()) == list(ref.state_dict())
    ref.load_state_dict(ours.state_dict())
    x = torch.randn(rows, cin, device=DEV, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    g = torch.randn(rows, cout, device=DEV)
    ours(x).backward(g)
    ref(x2).backward(g)
    torch.testing.assert_close(ours(x), ref(x2), rtol=1e-5, atol=1e-5)
    for a, r in ((x.grad, x2.grad), (ours.weight.grad, ref.weight.grad), (ours.bias.grad, ref.bias.grad)):
        assert (a - r).norm().item() <= 1e-5 * r.norm().item() + 1e-6
    ours.zero_grad(); ref.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = ours(x.detach())
        want = ref(x.detach())
    assert type(y.grad_fn).__name__ == "_LinearRowsBackward"
    y.backward(g.bfloat16())
    want.backward(g.bfloat16())
    exact = g.bfloat16().float().sum(0)
    assert (ours.bias.grad - exact).norm() <= 1.05 * (ref.bias.grad - exact).norm() + 2e-2 * exact.norm()
    assert (ours.weight.grad - ref.weight.grad).norm() <= 2e-2 * ref.weight.grad.norm() + 1e-6
