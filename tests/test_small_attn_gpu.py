"""csrc/attn_small.hip (MFMA attention for <= 128 queries) against an fp32 framework reference, forward and backward,
including key-padding masks, strided (packed) q / k, and dropout with the mask re-derived from the counter hash."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
M32 = 0xFFFFFFFF


def mix32(h):
    h = h ^ (h >> 16)
    h = (h * 0x7FEB352D) & M32
    h = h ^ (h >> 15)
    h = (h * 0x846CA68B) & M32
    return h ^ (h >> 16)


def mixp(x):
    """mixp of csrc/pcm_attn.hpp: xorshift, one 24-bit multiply (v_mul_u32_u24 keeps the low 32 bits of the product), xorshift."""
    x = x ^ (x >> 16)
    x = ((x & 0xFFFFFF) * 0xD35A2D) & M32
    return x ^ (x >> 12)


def keep_mask(seed, site, B, H, L, S, p):
    """The attention-dropout mask of csrc/attn_small.hip, as int64 tensor arithmetic: one hash per pair of adjacent keys,
    low / high 16 bits against thr16 = round(p * 65536)."""
    rowid = torch.arange(B * H * L, dtype=torch.int64, device=DEV)
    k = (seed & M32) ^ (((seed >> 32) * 0x9E3779B9) & M32) ^ ((site * 0x85EBCA6B) & M32)
    rowbase = mix32((k ^ rowid) & M32)
    key = torch.arange(S, dtype=torch.int64, device=DEV)
    bits = mixp((rowbase[:, None] + (key >> 1)[None, :] * 0x9E3779B1) & M32)
    half = torch.where((key & 1)[None, :] == 1, bits >> 16, bits & 0xFFFF)
    return (half >= int(p * 65536.0 + 0.5)).view(B, H, L, S)


def reference(q, k, v, kpm, heads, keep=None, p=0.0):
    B, L, E = q.shape
    S = k.shape[1]
    hd = E // heads
    qh, kh, vh = (t.float().view(B, -1, heads, hd).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / math.sqrt(hd)
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    pr = torch.softmax(s, dim=-1)
    if keep is not None:
        pr = pr * keep / (1 - p)
    return (pr @ vh).transpose(1, 2).reshape(B, L, E)


def run_case(B, H, L, S, masked, packed, p=0.0):
    from pointcloudmatters_amd.policy import fused_ops, small_attn

    torch.manual_seed(L * 1000 + S)
    E = H * 64
    if packed:  # q | k interleaved like the in-projection's (B, L, 2, E) buffer
        assert L == S
        qk = torch.randn(B, L, 2, E, device=DEV).bfloat16().requires_grad_(True)
        q, k = qk[:, :, 0], qk[:, :, 1]
    else:
        q = torch.randn(B, L, E, device=DEV).bfloat16().requires_grad_(True)
        k = torch.randn(B, S, E, device=DEV).bfloat16().requires_grad_(True)
    v = torch.randn(B, S, E, device=DEV).bfloat16().requires_grad_(True)
    kpm = None
    if masked:
        kpm = torch.zeros(B, S, dtype=torch.bool, device=DEV)
        for b in range(B):
            kpm[b, S - 1 - (7 * b) % max(1, S // 2):] = True
        kpm[:, 0] = False
    g = torch.randn(B, L, E, device=DEV).bfloat16()
    assert small_attn.supported(q, k, v, H, 0.0) == (L > small_attn.MAX_QUERIES or S <= small_attn.MAX_KEYS)
    keep = None
    if p > 0:
        ctx = fused_ops.FusedContext(DEV)
        ctx.set_step(3)
        torch.cuda.synchronize()
        with fused_ops.activate(ctx):
            out = small_attn.small_attention(q, k, v, kpm, H, p)
        keep = keep_mask(int(ctx.seed.item()), 1, B, H, L, S, p).float()
        if keep.numel() >= 100_000:  # a statistical check: only meaningful with enough samples
            assert abs(keep.mean().item() - (1 - p)) < 0.02
    else:
        out = small_attn.small_attention(q, k, v, kpm, H, 0.0)
    want = reference(q, k, v, kpm, H, keep, p)
    assert out.shape == (B, L, E) and out.dtype == torch.bfloat16
    err = (out.float() - want).abs().max().item()
    assert err <= 2e-2 * want.abs().max().item() + 2e-2, err
    leaves = [qk, v] if packed else [q, k, v]
    got = torch.autograd.grad(out, leaves, g)
    exp = torch.autograd.grad(want, leaves, g.float())
    for a, r in zip(got, exp):
        scale = r.float().abs().max().item()
        assert (a.float() - r.float()).abs().max().item() <= 3e-2 * scale + 1e-2, ((a.float() - r.float()).abs().max().item(), scale)


@pytest.mark.parametrize("B,H,L,S", [(2, 8, 100, 100), (3, 8, 102, 102), (2, 8, 100, 515), (1, 2, 1, 5), (2, 4, 128, 33),
                                     (8, 8, 100, 100), (2, 1, 33, 64), (1, 8, 128, 128), (2, 8, 515, 515), (1, 2, 129, 40), (1, 1, 300, 257), (2, 8, 100, 2051)])
@pytest.mark.parametrize("masked", [False, True])
def test_small_attention_matches_fp32_reference(B, H, L, S, masked):
    run_case(B, H, L, S, masked, packed=False)


def test_small_attention_with_packed_qk_views():
    run_case(2, 8, 102, 102, True, packed=True)


@pytest.mark.parametrize("L,S", [(100, 100), (100, 515)])
def test_small_attention_dropout_uses_the_counter_hash_consistently(L, S):
    run_case(2, 8, L, S, False, packed=False, p=0.1)


def test_small_attention_rejects_what_it_does_not_cover():
    from pointcloudmatters_amd.policy import small_attn

    q = torch.randn(2, 2051, 512, device=DEV).bfloat16()
    assert small_attn.supported(q, q, q, 8)              # long query sets: csrc/attn_flash.hip behind the same interface
    assert not small_attn.supported(q[:, :100].float(), q.float(), q.float(), 8)  # fp32
    assert not small_attn.supported(q[:, :100], q, q, 4)  # head_dim 128


def test_zero_upstream_gradient_takes_the_fast_path_and_returns_exact_zeros():
    from pointcloudmatters_amd.policy import small_attn

    torch.manual_seed(3)
    B, H, L, S = 2, 8, 100, 515
    q = torch.randn(B, L, 512, device=DEV).bfloat16().requires_grad_(True)
    k = torch.randn(B, S, 512, device=DEV).bfloat16().requires_grad_(True)
    v = torch.randn(B, S, 512, device=DEV).bfloat16().requires_grad_(True)
    out = small_attn.small_attention(q, k, v, None, H, 0.0)
    for g in torch.autograd.grad(out, (q, k, v), torch.zeros_like(out), retain_graph=True):
        assert torch.count_nonzero(g) == 0
    g = torch.zeros_like(out)
    g[1, 37, 200] = 1.0  # one non-zero element anywhere re-enables the full computation for that (batch, head)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), g)
    assert torch.count_nonzero(dq[1]) > 0 and torch.count_nonzero(dk[1]) > 0 and torch.count_nonzero(dv[1]) > 0
    assert torch.count_nonzero(dq[0]) == 0 and torch.count_nonzero(dv[0]) == 0


@pytest.mark.parametrize("first_only", ["keep", "prune_backward"])
def test_decoder_memory_gradients_through_the_shared_buffer(first_only):
    """TransformerDecoder._project_memory hands every layer a (B, S, E) slice of ONE key / value projection; the layers'
    cross-attention backward writes dK / dV straight into the shared (B, S, n, E) gradient buffer (transformer.GradArena)
    instead of autograd stacking n tensors.  Same kernels either way: the gradients must be bit-identical to the stacked
    path, and the no-copy path must actually have been taken."""
    from pointcloudmatters_amd.policy import fused_ops, transformer as tr

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    dec = tr.TransformerDecoder(tr.TransformerDecoderLayer(128, 2, dim_feedforward=64, dropout=0.0), 3, norm=torch.nn.LayerNorm(128),
                                return_intermediate=True).to(dev)
    dec.first_only = first_only
    B, S, L = 3, 333, 20
    mem = torch.randn(B, S, 128, device=dev)
    pos = torch.randn(B, S, 128, device=dev)
    tgt = torch.zeros(B, L, 128, device=dev)
    qpos = torch.randn(B, L, 128, device=dev)

    def run(shared):
        orig = tr.shared_unbind
        if not shared:
            tr.shared_unbind = lambda y, n: y.unflatten(-1, (n, y.shape[-1] // n)).unbind(-2)
        try:
            m = mem.clone().requires_grad_(True)
            for p in dec.parameters():
                p.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                hs = dec(tgt, m, pos=pos, query_pos=qpos)
            hs[0].float().square().sum().backward()
            return m.grad.clone(), {n: p.grad.clone() for n, p in dec.named_parameters() if p.grad is not None}
        finally:
            tr.shared_unbind = orig

    before = tr.GradArena.hits
    g_mem, g_par = run(True)
    assert tr.GradArena.hits == before + 2  # keys and values
    r_mem, r_par = run(False)
    assert tr.GradArena.hits == before + 2
    assert torch.equal(g_mem, r_mem)
    assert g_par.keys() == r_par.keys()
    for n in g_par:
        assert torch.equal(g_par[n], r_par[n]), n
