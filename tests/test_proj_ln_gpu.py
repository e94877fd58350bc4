"""csrc/proj_ln.hip: out = LayerNorm(x + dropout(a W^T + b)) as one matrix-core kernel, against (i) the composition it replaces -- the
bf16 library product followed by csrc/drln.hip, same dropout masks bit for bit -- and (ii) an fp64 evaluation of the formula.
Called through the C ABI with the tensors' pointers, so the same test body also runs on the host wave64 model
(tests/test_wavesim_parity.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _call(a, W, bias, x, gamma, beta, p, seed, site, pos=None, want16=False, a_ls=None):
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    R, E = x.shape
    K = W.shape[1]
    f32 = dict(dtype=torch.float32, device=x.device)
    s, out = torch.empty(R, E, **f32), torch.empty(R, E, **f32)
    mean, rstd = torch.empty(R, **f32), torch.empty(R, **f32)
    sum16 = torch.empty(R, E, dtype=torch.bfloat16, device=x.device) if pos is not None else None
    out16 = torch.empty(R, E, dtype=torch.bfloat16, device=x.device) if want16 else None
    ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
    rc = L.pcm_proj_drln_mfma_forward_hip(R, E, K, ptr(a), a_ls if a_ls is not None else K, ptr(W), ptr(bias),
                                          int(bias is not None and bias.dtype == torch.bfloat16), ptr(x), ptr(gamma), ptr(beta), 1e-5, p,
                                          ptr(seed), site, ptr(s), ptr(out), ptr(mean), ptr(rstd), ptr(pos),
                                          pos.numel() if pos is not None else 0, ptr(sum16), ptr(out16), _lib.raw_stream())
    _lib.check(rc, "pcm_proj_drln_mfma_forward_hip")
    torch.cuda.synchronize()
    return s, out, mean, rstd, sum16, out16


def _drln(y16, x, gamma, beta, p, seed, site):
    """The kernel this one must agree with: csrc/drln.hip on a given bf16 product (same counter-based dropout mask)."""
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    R, E = x.shape
    f32 = dict(dtype=torch.float32, device=x.device)
    s, out = torch.empty(R, E, **f32), torch.empty(R, E, **f32)
    mean, rstd = torch.empty(R, **f32), torch.empty(R, **f32)
    rc = L.pcm_drln_forward_hip(R, E, 1, x.data_ptr(), y16.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, p,
                                seed.data_ptr() if seed is not None else 0, site, s.data_ptr(), out.data_ptr(), mean.data_ptr(),
                                rstd.data_ptr(), _lib.raw_stream())
    _lib.check(rc, "pcm_drln_forward_hip")
    torch.cuda.synchronize()
    return s, out, mean, rstd


def _inputs(R, E, K, seed, bias_dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn(R, K, generator=g) * 0.7).bfloat16().to(DEV)
    W = (torch.randn(E, K, generator=g) / K ** 0.5).bfloat16().to(DEV)
    bias = None if bias_dtype is None else (torch.randn(E, generator=g) * 0.1).to(bias_dtype).to(DEV)
    x = torch.randn(R, E, generator=g).to(DEV)
    gamma = (1 + 0.2 * torch.randn(E, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(E, generator=g)).to(DEV)
    return a, W, bias, x, gamma, beta


# decoder / CVAE-encoder / ragged row counts, every supported width, K != E, a tile with a single row, fp32 and missing bias
# ... and the 64-row tile taken from 2048 rows on (the encoder's 4120 rows, a ragged tail, a narrow and a wide reduction)
CASES = [(4120, 512, 512, torch.bfloat16), (2051, 256, 512, torch.float32), (2048, 512, 64, None), (800, 512, 512, torch.bfloat16), (816, 512, 512, torch.bfloat16), (37, 256, 256, torch.float32), (1, 512, 512, None),
         (100, 768, 256, torch.bfloat16), (48, 1024, 1024, torch.bfloat16), (515, 512, 64, torch.bfloat16), (16, 256, 32, torch.float32)]


@pytest.mark.parametrize("R,E,K,bias_dtype", CASES)
def test_projection_residual_norm_matches_the_formula(R, E, K, bias_dtype):
    a, W, bias, x, gamma, beta = _inputs(R, E, K, R + E + K, bias_dtype)
    s, out, mean, rstd, _, _ = _call(a, W, bias, x, gamma, beta, 0.0, None, 5)
    y64 = a.double() @ W.double().t() + (bias.double() if bias is not None else 0.0)
    y16 = y64.float().bfloat16()  # what a bf16 product hands on (fp64 accumulation: at most one bf16 step from any fp32 order)
    s_ref = x.double() + y16.double()
    step = y64.abs().clamp_min(1e-3) * 2.0 ** -7  # one bf16 step of the product
    assert ((s.double() - s_ref).abs() <= step + 1e-6).all()
    # the rest is a function of s: LayerNorm of the kernel's OWN s in fp64 must reproduce out / mean / rstd to fp32 accuracy
    mu = s.double().mean(1, keepdim=True)
    var = ((s.double() - mu) ** 2).mean(1, keepdim=True)
    want = (s.double() - mu) / (var + 1e-5).sqrt() * gamma.double() + beta.double()
    torch.testing.assert_close(out.double(), want, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(mean.double(), mu.squeeze(1), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rstd.double(), (var + 1e-5).rsqrt().squeeze(1), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("R,E,K", [(800, 512, 512), (70, 256, 512), (2115, 512, 512)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_same_masks_and_results_as_product_plus_drln(R, E, K, p):
    """Fed the product this kernel itself formed (recovered from s with dropout off), csrc/drln.hip must give the same s / out / mean /
    rstd -- with dropout on that pins the mask: (seed, site, element) -> keep is the same function, so pcm_drln_backward2_hip can
    serve as this kernel's backward."""
    a, W, bias, x, gamma, beta = _inputs(R, E, K, 7 * R + K)
    seed = torch.tensor([123456789012345], dtype=torch.int64, device=DEV)
    s0 = _call(a, W, bias, x, gamma, beta, 0.0, None, 9)[0]
    y16 = (s0 - x).bfloat16()  # exact: s0 = x + y with y a bf16 value ... up to the fp32 rounding of the sum
    got = _call(a, W, bias, x, gamma, beta, p, seed if p > 0 else None, 9)
    want = _drln(y16, x, gamma, beta, p, seed if p > 0 else None, 9)
    if p > 0:  # dropped elements are exactly x: the masks agree element for element
        dropped_got, dropped_want = got[0] == x, want[0] == x
        assert torch.equal(dropped_got, dropped_want)
        frac = dropped_got.float().mean().item()
        assert abs(frac - p) < 0.02, frac
    for g, w_, name in zip(got[:4], want, ("s", "out", "mean", "rstd")):
        torch.testing.assert_close(g, w_, rtol=2e-2, atol=2e-2, msg=name)  # y16 went through one more rounding (s0 - x)
    assert (got[0] - want[0]).abs().max().item() <= 2.0 ** -6 * (s0 - x).abs().max().item() + 1e-6


def test_consumer_operands_and_row_stride():
    """sum16 = bf16(out + pos) with pos broadcast over the leading rows, out16 = bf16(out); `a` as a strided view (row stride > K)."""
    _consumer_operands(200)
    _consumer_operands(2100)  # the 64-row tile


def _consumer_operands(R):
    E, K = 512, 512
    a, W, bias, x, gamma, beta = _inputs(R, E, K, 3)
    pos = torch.randn(100, E, generator=torch.Generator().manual_seed(4)).to(DEV)  # the decoder's query_pos: 100 queries, batch-major rows
    wide = torch.zeros(R, K + 64, dtype=torch.bfloat16, device=DEV)
    wide[:, :K] = a
    ref = _call(a, W, bias, x, gamma, beta, 0.0, None, 1)
    got = _call(wide, W, bias, x, gamma, beta, 0.0, None, 1, pos=pos, want16=True, a_ls=K + 64)
    for g, r in zip(got[:4], ref[:4]):
        assert torch.equal(g, r)
    out = got[1]
    assert torch.equal(got[5], out.bfloat16())
    assert torch.equal(got[4], (out + pos.repeat(R // 100, 1)).bfloat16())


def test_argument_contract():
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    assert L.pcm_proj_drln_mfma_supported(512, 512) == 1 and L.pcm_proj_drln_mfma_supported(512, 48) == 0
    assert L.pcm_proj_drln_mfma_supported(384, 512) == 0 and L.pcm_proj_drln_mfma_supported(1024, 1024) == 1
    z = [0] * 21
    assert L.pcm_proj_drln_mfma_forward_hip(-1, 512, 512, *z) == 1       # negative size
    assert L.pcm_proj_drln_mfma_forward_hip(0, 512, 512, 0, 512, *z[2:]) == 0  # empty call
    assert L.pcm_proj_drln_mfma_forward_hip(16, 384, 512, 0, 512, *z[2:]) == 2  # unsupported width


@pytest.mark.parametrize("B", [8, 24])  # 800 rows: 16-row tiles; 2400 rows: the 64-row tile (PROJ_MFMA_LONG)
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_fused_node_with_and_without_the_matrix_core_kernel(monkeypatch, p, B):
    """fused_ops.proj_drln (the autograd node of every attention sub-layer's tail) with PROJ_MFMA on against the same node with the
    library product + csrc/drln.hip: same dropout masks, outputs and all gradients equal up to the bf16 rounding of the product."""
    import torch.nn as nn

    from pointcloudmatters_amd.policy import fused_ops

    torch.manual_seed(5)
    E, Lq = 512, 100
    monkeypatch.setattr(fused_ops, "PROJ_MFMA_LONG", True)
    lin, norm, drop = nn.Linear(E, E).to(DEV), nn.LayerNorm(E).to(DEV), nn.Dropout(p)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5), norm.bias.uniform_(-0.2, 0.2)
    a0 = torch.randn(Lq, B, E, device=DEV).bfloat16()
    x0 = torch.randn(Lq, B, E, device=DEV)
    g = torch.randn(Lq, B, E, device=DEV)
    runs, called = {}, []
    orig = fused_ops._lib.check

    def check(rc, what, *args, **kw):
        called.append(what)
        return orig(rc, what, *args, **kw)

    monkeypatch.setattr(fused_ops._lib, "check", check)
    for flag in (False, True):
        monkeypatch.setattr(fused_ops, "PROJ_MFMA", flag)
        a, x = a0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
        for q in list(lin.parameters()) + list(norm.parameters()):
            q.grad = None
        ctx = fused_ops.FusedContext(torch.device(DEV))
        ctx.set_step(3)
        called.clear()
        with fused_ops.activate(ctx), torch.autocast("cuda", dtype=torch.bfloat16):
            assert fused_ops.drln_supported(x, None, norm, y_dtype=torch.bfloat16)
            out = fused_ops.proj_drln(a, lin, x, norm, drop)
        assert ("pcm_proj_drln_mfma_forward_hip" in called) == flag, called
        out.backward(g)
        torch.cuda.synchronize()
        runs[flag] = [out.detach(), a.grad.float(), x.grad, lin.weight.grad.float(), lin.bias.grad.float(), norm.weight.grad, norm.bias.grad]
    for got, want, name in zip(runs[True], runs[False], ("out", "da", "dx", "dW", "db", "dgamma", "dbeta")):
        scale = want.abs().max().item()
        assert (got - want).abs().max().item() <= 2e-2 * scale + 1e-6, (name, (got - want).abs().max().item(), scale)
    # the residual stream is where a different mask would show: dropped positions pass x's gradient only
    assert (runs[True][2] - runs[False][2]).abs().mean().item() <= 2e-3 * runs[False][2].abs().mean().item() + 1e-7


# ------------------------------------------------------------------------------------------------------- pcm_linear_mfma
def _linear(a, W, bias, pos=None, pos_cols=0, out_dtype=torch.bfloat16, a_ls=None, out=None, a_alt=None, emit=(None, None)):
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    R, K = a.shape[0], W.shape[1]
    N = W.shape[0]
    if out is None:
        out = torch.empty(R, N, dtype=out_dtype, device=a.device)
    ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
    rc = L.pcm_linear_mfma_forward_hip(R, N, K, ptr(a), int(a.dtype == torch.float32), a_ls if a_ls is not None else a.stride(0), ptr(a_alt),
                                       ptr(pos), pos.numel() if pos is not None else 0, pos_cols, ptr(W), ptr(bias),
                                       int(bias is not None and bias.dtype == torch.bfloat16), ptr(out), int(out.dtype == torch.bfloat16),
                                       out.stride(0), ptr(emit[0]), ptr(emit[1]), _lib.raw_stream())
    _lib.check(rc, "pcm_linear_mfma_forward_hip")
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("R,N,K", [(800, 1536, 512), (816, 512, 512), (37, 264, 64), (1, 8, 32), (100, 768, 1024),
                                   (4120, 1536, 512), (2051, 264, 64), (2048, 512, 1024)])  # the last three: 64-row tiles
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_linear_from_bf16_rows(R, N, K, out_dtype):
    a, W, bias, *_ = _inputs(R, N, K, 11 * R + N)
    got = _linear(a, W, bias, out_dtype=out_dtype)
    want = a.double() @ W.double().t() + bias.double()
    if out_dtype == torch.float32:
        torch.testing.assert_close(got.double(), want, rtol=2e-5, atol=2e-5 * K ** 0.5)
    else:  # one rounding to bf16 of an fp32 sum: within one bf16 step of the exact value
        assert ((got.double() - want).abs() <= want.abs().clamp_min(1e-2) * 2.0 ** -7).all()


@pytest.mark.parametrize("B", [8, 24])  # 800 rows / 2400 rows (64-row tiles)
def test_in_projection_with_position_embedding_fused(B):
    """q | k | v = in_proj([x + pos ; x]): what csrc/tokens.hip's add + cast launch and the doubled-row product did -- columns below
    pos_cols (q, k) see bf16(x + pos), the others (v) bf16(x); pos is the decoder's (100, 1, E) query_pos broadcast over the batch."""
    E, Lq = 512, 100
    g = torch.Generator().manual_seed(21)
    x = torch.randn(Lq * B, E, generator=g).to(DEV)
    pos = torch.randn(Lq, E, generator=g).to(DEV)  # rows (query, batch)-major would need an expanded pos; here batch-major rows: block of Lq
    W = (torch.randn(3 * E, E, generator=g) / E ** 0.5).bfloat16().to(DEV)
    bias = (0.1 * torch.randn(3 * E, generator=g)).bfloat16().to(DEV)
    got = _linear(x, W, bias, pos=pos, pos_cols=2 * E)
    xp = (x + pos.repeat(B, 1)).bfloat16().double()
    want_qk = xp @ W[: 2 * E].double().t() + bias[: 2 * E].double()
    want_v = x.bfloat16().double() @ W[2 * E:].double().t() + bias[2 * E:].double()
    want = torch.cat([want_qk, want_v], 1)
    assert ((got.double() - want).abs() <= want.abs().clamp_min(1e-2) * 2.0 ** -7).all()
    # the query projection of the cross-attention: every column sees x + pos; fp32 operand without pos = a plain cast
    q = _linear(x, W[:E].contiguous(), bias[:E].contiguous(), pos=pos, pos_cols=E)
    assert torch.equal(q, got[:, :E].contiguous())
    v = _linear(x, W[2 * E:].contiguous(), bias[2 * E:].contiguous())
    assert torch.equal(v, got[:, 2 * E:].contiguous())
    # the bf16 operands written out by the same launch (the backward's weight-gradient operands) ...
    e_pos, e_x = torch.empty(Lq * B, E, dtype=torch.bfloat16, device=DEV), torch.empty(Lq * B, E, dtype=torch.bfloat16, device=DEV)
    again = _linear(x, W, bias, pos=pos, pos_cols=2 * E, emit=(e_pos, e_x))
    assert torch.equal(again, got)
    assert torch.equal(e_pos, (x + pos.repeat(B, 1)).bfloat16()) and torch.equal(e_x, x.bfloat16())
    # ... and the same product from operands a producer emitted: two bf16 matrices
    assert torch.equal(_linear(e_pos, W, bias, pos_cols=2 * E, a_alt=e_x), got)


def test_linear_strided_operands_and_contract():
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    R, N, K = 50, 512, 256
    a, W, bias, *_ = _inputs(R, N, K, 5)
    wide_a = torch.zeros(R, K + 32, dtype=torch.bfloat16, device=DEV)
    wide_a[:, :K] = a
    wide_o = torch.full((R, N + 16), 7.0, dtype=torch.bfloat16, device=DEV)
    ref = _linear(a, W, bias)
    _linear(wide_a[:, :K], W, bias, out=wide_o[:, :N])
    assert torch.equal(wide_o[:, :N], ref) and bool((wide_o[:, N:] == 7.0).all())  # nothing written past a row's N columns
    assert L.pcm_linear_mfma_supported(1536, 512, 1024) == 1 and L.pcm_linear_mfma_supported(1536, 512, 100) == 0
    assert L.pcm_linear_mfma_supported(12, 512, 0) == 0 and L.pcm_linear_mfma_supported(512, 48, 0) == 0
    assert L.pcm_linear_mfma_forward_hip(-1, 512, 512, 0, 0, 512, 0, 0, 0, 0, 0, 0, 0, 0, 1, 512, 0, 0, 0) == 1
    assert L.pcm_linear_mfma_forward_hip(0, 512, 512, 0, 0, 512, 0, 0, 0, 0, 0, 0, 0, 0, 1, 512, 0, 0, 0) == 0


@pytest.mark.parametrize("B", [8, 24])
@pytest.mark.parametrize("batch_first_pos", [False, True])
def test_in_projection_nodes_with_and_without_the_matrix_core_kernel(monkeypatch, batch_first_pos, B):
    """fused_ops.self_attn_in_proj and add_pos_linear (one autograd node each) with LINEAR_MFMA on against the add + cast launch and the
    library product(s): q / k / v and every gradient equal up to one bf16 rounding of the products."""
    import torch.nn as nn

    from pointcloudmatters_amd.policy import fused_ops

    torch.manual_seed(8)
    E, Lq = 512, 100
    monkeypatch.setattr(fused_ops, "PROJ_MFMA_LONG", True)
    mha = nn.MultiheadAttention(E, 8).to(DEV)
    lin = nn.Linear(E, E).to(DEV)
    x0 = torch.randn(B, Lq, E, device=DEV)
    pos0 = torch.randn(1 if batch_first_pos else B, Lq, E, device=DEV)
    gq, gk, gv, gy = (torch.randn(B, Lq, E, device=DEV).bfloat16() for _ in range(4))
    called, orig = [], fused_ops._lib.check

    def check(rc, what, *args, **kw):
        called.append(what)
        return orig(rc, what, *args, **kw)

    monkeypatch.setattr(fused_ops._lib, "check", check)
    runs = {}
    for flag in (False, True):
        monkeypatch.setattr(fused_ops, "LINEAR_MFMA", flag)
        x, pos = x0.clone().requires_grad_(True), pos0.clone().requires_grad_(True)
        for p_ in list(mha.parameters()) + list(lin.parameters()):
            p_.grad = None
        ctx = fused_ops.FusedContext(torch.device(DEV))
        called.clear()
        with fused_ops.activate(ctx), torch.autocast("cuda", dtype=torch.bfloat16):
            assert fused_ops.self_attn_in_proj_supported(x, pos, mha) and fused_ops.add_pos_linear_supported(x, pos, lin.weight, lin.bias)
            q, k, v, xr = fused_ops.self_attn_in_proj(x, pos, mha)
            y = fused_ops.add_pos_linear(xr, pos, lin.weight, lin.bias)
        assert (called.count("pcm_linear_mfma_forward_hip") == 2) == flag, called
        assert ("pcm_add_cast2_hip" in called) != flag, called
        ((q * gq).float().sum() + (k * gk).float().sum() + (v * gv).float().sum() + (y * gy).float().sum()).backward()
        torch.cuda.synchronize()
        runs[flag] = [q.float(), k.float(), v.float(), y.float(), x.grad, pos.grad, mha.in_proj_weight.grad.float(), mha.in_proj_bias.grad.float(),
                      lin.weight.grad.float(), lin.bias.grad.float()]
    for got, want, name in zip(runs[True], runs[False], ("q", "k", "v", "y", "dx", "dpos", "dW_in", "db_in", "dW", "db")):
        scale = want.abs().max().item()
        assert (got - want).abs().max().item() <= 2e-2 * scale + 1e-6, (name, (got - want).abs().max().item(), scale)
    for i in range(4):  # the forward values differ only by the summation order inside one bf16 rounding
        assert (runs[True][i] - runs[False][i]).abs().mean().item() <= 2e-3 * runs[False][i].abs().mean().item()


# ------------------------------------------------------------------------------------- pcm_proj_drln_mfma_backward (round 6)
def _bwd_inputs(R, E, K, seed):
    g = torch.Generator().manual_seed(seed)
    dout = torch.randn(R, E, generator=g).to(DEV)
    dout2 = (0.5 * torch.randn(R, E, generator=g)).to(DEV)
    s = (torch.randn(R, E, generator=g) * 1.3 + 0.2).to(DEV)
    mean = s.mean(1).contiguous()
    rstd = (s.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    gamma = (1 + 0.2 * torch.randn(E, generator=g)).to(DEV)
    W = (torch.randn(E, K, generator=g) / E ** 0.5).bfloat16().to(DEV)
    return dout, dout2, s, mean, rstd, gamma, W


def _row_kernel(dout, dout2, s, mean, rstd, gamma, p, seed, site):
    """csrc/drln.hip's backward: what the chain kernel's row phase must reproduce bit for bit."""
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    R, E = s.shape
    dx = torch.empty_like(s)
    dy = torch.empty(R, E, dtype=torch.bfloat16, device=s.device)
    partial = torch.empty(L.pcm_drln_blocks(R) * 3 * E, dtype=torch.float32, device=s.device)
    sums = torch.empty(3, E, dtype=torch.float32, device=s.device)
    db16 = torch.empty(E, dtype=torch.bfloat16, device=s.device)
    rc = L.pcm_drln_backward2_hip(R, E, 1, dout.data_ptr(), dout2.data_ptr() if dout2 is not None else 0, s.data_ptr(), mean.data_ptr(),
                                  rstd.data_ptr(), gamma.data_ptr(), p, seed.data_ptr() if seed is not None else 0, site, dx.data_ptr(),
                                  dy.data_ptr(), partial.data_ptr(), sums.data_ptr(), db16.data_ptr(), _lib.raw_stream())
    _lib.check(rc, "pcm_drln_backward2_hip")
    torch.cuda.synchronize()
    return dx, dy, sums, db16


def _chain_bwd(dout, dout2, s, mean, rstd, gamma, W, p, seed, site, da_ls=None, defer=False):
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    R, E = s.shape
    K = W.shape[1]
    dx = torch.empty_like(s)
    dy = torch.empty(R, E, dtype=torch.bfloat16, device=s.device)
    ls = K if da_ls is None else da_ls
    da = torch.full((R, ls), 7.0, dtype=torch.bfloat16, device=s.device)
    blocks = L.pcm_proj_drln_mfma_backward_blocks(R)
    partial = torch.empty(blocks * 3 * E, dtype=torch.float32, device=s.device)
    sums = torch.empty(3, E, dtype=torch.float32, device=s.device)
    db16 = torch.empty(E, dtype=torch.bfloat16, device=s.device)
    rc = L.pcm_proj_drln_mfma_backward_hip(R, E, K, dout.data_ptr(), dout2.data_ptr() if dout2 is not None else 0, s.data_ptr(), mean.data_ptr(),
                                           rstd.data_ptr(), gamma.data_ptr(), p, seed.data_ptr() if seed is not None else 0, site, W.data_ptr(),
                                           dx.data_ptr(), dy.data_ptr(), da.data_ptr(), ls, partial.data_ptr(), 0 if defer else sums.data_ptr(),
                                           0 if defer else db16.data_ptr(), _lib.raw_stream())
    _lib.check(rc, "pcm_proj_drln_mfma_backward_hip")
    torch.cuda.synchronize()
    if defer:
        sums = partial.view(blocks, 3, E).double().sum(0).float()
    return dx, dy, sums, db16, da


@pytest.mark.parametrize("R,E,K", [(800, 512, 512), (816, 512, 512), (37, 256, 256), (1, 512, 512), (100, 1024, 1024), (50, 768, 256),
                                   (129, 256, 1024)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_backward_chain_equals_row_kernel_plus_product(R, E, K, p):
    """dx and dy are the row kernel's BITS (same arithmetic per row, same dropout hash); the column sums agree up to the fp32 grouping of
    the per-workgroup partial rows; da = bf16(dy W) with the products accumulated in fp32 on the matrix cores: against the fp64 product
    of the SAME bf16 dy and W, within one bf16 rounding of the result."""
    dout, dout2, s, mean, rstd, gamma, W = _bwd_inputs(R, E, K, 7 * R + E)
    seed = torch.tensor([12345], dtype=torch.int64, device=DEV) if p > 0 else None
    for d2 in (None, dout2):
        dx0, dy0, sums0, db0 = _row_kernel(dout, d2, s, mean, rstd, gamma, p, seed, 3)
        dx, dy, sums, db16, da = _chain_bwd(dout, d2, s, mean, rstd, gamma, W, p, seed, 3)
        assert torch.equal(dx, dx0) and torch.equal(dy.view(torch.int16), dy0.view(torch.int16))
        scale = sums0.abs().max().item()
        assert (sums - sums0).abs().max().item() <= 2e-5 * scale + 1e-6
        assert (db16.float() - db0.float()).abs().max().item() <= 1e-2 * db0.float().abs().max().item() + 1e-6
        want = (dy0.double() @ W.double())
        err = (da.double() - want).abs().max().item()
        assert err <= 2 ** -8 * want.abs().max().item() + 1e-6, (err, want.abs().max().item())
    if p > 0:  # the mask really drops: a share p of dy is exactly zero where dx is not
        dropped = ((dy.float() == 0) & (dx != 0)).float().mean().item()
        assert abs(dropped - p) < 0.02, dropped


def test_backward_chain_row_stride_partial_rows_and_contract():
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    R, E, K = 100, 512, 512
    dout, dout2, s, mean, rstd, gamma, W = _bwd_inputs(R, E, K, 3)
    ref = _chain_bwd(dout, None, s, mean, rstd, gamma, W, 0.0, None, 0)
    wide = _chain_bwd(dout, None, s, mean, rstd, gamma, W, 0.0, None, 0, da_ls=K + 64)   # da into a wider buffer: the columns past K untouched
    assert torch.equal(wide[4][:, :K], ref[4]) and bool((wide[4][:, K:] == 7.0).all())
    left = _chain_bwd(dout, None, s, mean, rstd, gamma, W, 0.0, None, 0, defer=True)      # NULL result pointer: partial rows only
    assert (left[2] - ref[2]).abs().max().item() <= 2e-5 * ref[2].abs().max().item() + 1e-6
    assert L.pcm_proj_drln_mfma_backward_supported(512, 512) == 1 and L.pcm_proj_drln_mfma_backward_supported(512, 384) == 0
    assert L.pcm_proj_drln_mfma_backward_supported(384, 512) == 0 and L.pcm_proj_drln_mfma_backward_blocks(800) == 50
    z = [0] * 18
    assert L.pcm_proj_drln_mfma_backward_hip(-1, 512, 512, *z) == 1          # negative size
    assert L.pcm_proj_drln_mfma_backward_hip(0, 512, 512, *z[:13], 512, *z[14:]) == 0   # empty call
    assert L.pcm_proj_drln_mfma_backward_hip(16, 512, 384, *z[:13], 384, *z[14:]) == 2  # unsupported width


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_fused_node_backward_with_and_without_the_chain_kernel(monkeypatch, p):
    """fused_ops.proj_drln with PROJ_MFMA_BWD on against the same node with pcm_drln_backward2 + the library product: dx / dgamma / dbeta /
    db / dW as before (dy is the same tensor), da up to the bf16 rounding of the product."""
    import torch.nn as nn

    from pointcloudmatters_amd.policy import fused_ops

    torch.manual_seed(9)
    E, Lq, B = 512, 100, 8
    lin, norm, drop = nn.Linear(E, E).to(DEV), nn.LayerNorm(E).to(DEV), nn.Dropout(p)
    a0 = torch.randn(Lq, B, E, device=DEV).bfloat16()
    x0 = torch.randn(Lq, B, E, device=DEV)
    g = torch.randn(Lq, B, E, device=DEV)
    runs, called = {}, []
    orig = fused_ops._lib.check

    def check(rc, what, *args, **kw):
        called.append(what)
        return orig(rc, what, *args, **kw)

    monkeypatch.setattr(fused_ops._lib, "check", check)
    for flag in (False, True):
        monkeypatch.setattr(fused_ops, "PROJ_MFMA_BWD", flag)
        a, x = a0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
        for q in list(lin.parameters()) + list(norm.parameters()):
            q.grad = None
        ctx = fused_ops.FusedContext(torch.device(DEV))
        ctx.set_step(3)
        with fused_ops.activate(ctx), torch.autocast("cuda", dtype=torch.bfloat16):
            out = fused_ops.proj_drln(a, lin, x, norm, drop)
        called.clear()
        out.backward(g)
        torch.cuda.synchronize()
        assert ("pcm_proj_drln_mfma_backward_hip" in called) == flag and ("pcm_drln_backward2_hip" in called) == (not flag), called
        runs[flag] = [a.grad.float(), x.grad, lin.weight.grad.float(), lin.bias.grad.float(), norm.weight.grad, norm.bias.grad]
    for got, want, name in zip(runs[True], runs[False], ("da", "dx", "dW", "db", "dgamma", "dbeta")):
        scale = want.abs().max().item()
        tol = 2e-2 if name == "da" else (1e-2 if name == "db" else 2e-5)
        assert (got - want).abs().max().item() <= tol * scale + 1e-6, (name, (got - want).abs().max().item(), scale)
    assert torch.equal(runs[True][1], runs[False][1])  # dx: the same bits


# ------------------------------------------------------------------------------------- pcm_linear_mfma_backward (round 6)
def _lin_bwd(dy, W, dres, pos_cols, want_dpos, N=None):
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    R, K = dy.shape[0], W.shape[1]
    N = W.shape[0] if N is None else N
    dx = torch.full((R, K), 7.0, dtype=torch.float32, device=dy.device)
    dpos = torch.full((R, K), 7.0, dtype=torch.float32, device=dy.device) if want_dpos else None
    rc = L.pcm_linear_mfma_backward_hip(R, N, K, dy.data_ptr(), dy.stride(0), W.data_ptr(), dres.data_ptr() if dres is not None else 0,
                                        dx.data_ptr(), dpos.data_ptr() if dpos is not None else 0, pos_cols, _lib.raw_stream())
    _lib.check(rc, "pcm_linear_mfma_backward_hip")
    torch.cuda.synchronize()
    return dx, dpos


@pytest.mark.parametrize("R,N,K,pos_cols", [(800, 1536, 512, 1024), (816, 1536, 512, 1024), (800, 512, 512, 512), (37, 768, 256, 512), (1, 96, 256, 32),
                                            (100, 3072, 1024, 2048), (129, 64, 512, 64), (50, 1536, 512, 0)])
@pytest.mark.parametrize("with_res", [False, True])
def test_linear_backward_matches_the_fp64_products(R, N, K, pos_cols, with_res):
    """dx = dy W + dres and dpos = dy[:, :pos_cols] W[:pos_cols] against fp64 products of the same bf16 operands: the matrix cores accumulate
    in fp32 (no intermediate bf16 rounding, unlike the batched library product + add kernel this replaces), so the bound is fp32 summation
    noise over N terms."""
    g = torch.Generator().manual_seed(R + N)
    wide = torch.randn(R, N + 64, generator=g).bfloat16().to(DEV)      # dy as a strided view of a wider buffer (dq | dk | dv inside more)
    dy = wide[:, :N]
    W = (torch.randn(N, K, generator=g) / N ** 0.5).bfloat16().to(DEV)
    dres = torch.randn(R, K, generator=g).to(DEV) if with_res else None
    dx, dpos = _lin_bwd(dy, W, dres, pos_cols, True)
    want = dy.double() @ W.double()
    wpos = dy[:, :pos_cols].double() @ W[:pos_cols].double() if pos_cols < N else want
    tol = 3e-6 * N ** 0.5 * max(1.0, want.abs().max().item())
    assert (dx.double() - (want + (dres.double() if with_res else 0))).abs().max().item() <= tol
    assert (dpos.double() - wpos).abs().max().item() <= tol
    dx2, none = _lin_bwd(dy, W, dres, pos_cols, False)
    assert none is None and torch.equal(dx2, dx)


def test_linear_backward_contract():
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    assert L.pcm_linear_mfma_backward_supported(1536, 512, 1024) == 1 and L.pcm_linear_mfma_backward_supported(1536, 384, 1024) == 0
    assert L.pcm_linear_mfma_backward_supported(1530, 512, 1024) == 0 and L.pcm_linear_mfma_backward_supported(512, 512, 500) == 0
    assert L.pcm_linear_mfma_backward_supported(512, 512, 512) == 1 and L.pcm_linear_mfma_backward_supported(4096, 512, 0) == 0
    assert L.pcm_linear_mfma_backward_hip(-1, 512, 512, 0, 512, 0, 0, 0, 0, 512, 0) == 1   # negative size
    assert L.pcm_linear_mfma_backward_hip(0, 512, 512, 0, 512, 0, 0, 0, 0, 512, 0) == 0    # empty call
    assert L.pcm_linear_mfma_backward_hip(16, 512, 384, 0, 512, 0, 0, 0, 0, 512, 0) == 2   # unsupported width
    assert L.pcm_linear_mfma_backward_hip(16, 512, 512, 0, 256, 0, 0, 0, 0, 512, 0) == 1   # row stride below N


class _JointGrads(torch.autograd.Function):
    """Hands q, k, v the three column blocks of ONE (rows, 3E) gradient buffer, the layout csrc/attn_small.hip's backward writes."""

    @staticmethod
    def forward(ctx, q, k, v, g3):
        ctx.save_for_backward(g3)
        return (q.float().sum() + k.float().sum() + v.float().sum()) * 0.0

    @staticmethod
    def backward(ctx, g):
        (g3,) = ctx.saved_tensors
        E = g3.shape[-1] // 3
        return g3[..., :E], g3[..., E:2 * E], g3[..., 2 * E:], None


@pytest.mark.parametrize("batch_first_pos", [True, False])  # pos (1, Lq, E): broadcast over the batch, its gradient is summed; (B, Lq, E): x's shape
def test_in_projection_backward_with_and_without_the_matrix_core_kernel(monkeypatch, batch_first_pos):
    """The self-attention in-projection node's and the query-projection node's backward with LINEAR_MFMA_BWD on against the batched library
    product + add kernel / the library product: dx (residual gradient folded in), dpos, and the weight / bias gradients (unchanged path)."""
    import torch.nn as nn

    from pointcloudmatters_amd.policy import fused_ops

    torch.manual_seed(11)
    E, Lq, B = 512, 100, 8
    mha = nn.MultiheadAttention(E, 8).to(DEV)
    lin = nn.Linear(E, E).to(DEV)
    x0 = torch.randn(B, Lq, E, device=DEV)
    pos0 = torch.randn(1 if batch_first_pos else B, Lq, E, device=DEV)
    g3 = torch.randn(B, Lq, 3 * E, device=DEV).bfloat16()
    gy, gres = torch.randn(B, Lq, E, device=DEV).bfloat16(), torch.randn(B, Lq, E, device=DEV)
    called, orig = [], fused_ops._lib.check

    def check(rc, what, *args, **kw):
        called.append(what)
        return orig(rc, what, *args, **kw)

    monkeypatch.setattr(fused_ops._lib, "check", check)
    runs = {}
    for flag in (False, True):
        monkeypatch.setattr(fused_ops, "LINEAR_MFMA_BWD", flag)
        x, pos = x0.clone().requires_grad_(True), pos0.clone().requires_grad_(True)
        for p_ in list(mha.parameters()) + list(lin.parameters()):
            p_.grad = None
        ctx = fused_ops.FusedContext(torch.device(DEV))
        with fused_ops.activate(ctx), torch.autocast("cuda", dtype=torch.bfloat16):
            q, k, v, xr = fused_ops.self_attn_in_proj(x, pos, mha)
            y = fused_ops.add_pos_linear(x.detach().requires_grad_(True), pos, lin.weight, lin.bias)
        called.clear()
        (_JointGrads.apply(q, k, v, g3) + (xr * gres).sum() + (y * gy).float().sum()).backward()
        torch.cuda.synchronize()
        assert (called.count("pcm_linear_mfma_backward_hip") == 2) == flag, called
        assert ("pcm_add3_cast2_hip" in called) != flag, called
        runs[flag] = [x.grad, pos.grad, mha.in_proj_weight.grad.float(), mha.in_proj_bias.grad.float(), lin.weight.grad.float(), lin.bias.grad.float()]
    for got, want, name in zip(runs[True], runs[False], ("dx", "dpos", "dW_in", "db_in", "dW", "db")):
        scale = want.abs().max().item()
        tol = 2e-2 if name in ("dx", "dpos") else 1e-6
        assert (got - want).abs().max().item() <= tol * scale + 1e-6, (name, (got - want).abs().max().item(), scale)
