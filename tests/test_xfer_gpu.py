"""GPU: pcm_xfer_batch_hip (csrc/optim.hip) -- the gradient hand-off into the flat fp32 buffer and the packed-gradient assembly copies,
one table-driven launch per 96 jobs -- against the framework's per-tensor operations, bit for bit: every kind, sizes around the 8192-element
chunk and the 8-element vector width, misaligned sources / destinations, more jobs than one launch takes; and FlatAdamW.collect through it
against the multi-tensor route it replaced (first micro-batch: copies + zero runs; later ones: adds)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _jobs(dev, gen, sizes, kind, misalign=False):
    from pointcloudmatters_amd import _lib

    out = []
    for i, n in enumerate(sizes):
        off_d, off_s = ((i % 3) + 1, (i % 5) + 1) if misalign else (0, 0)
        if kind == _lib.XFER_COPY_2B:
            src = torch.randn(n + 8, device=dev, generator=gen).to(torch.bfloat16)[off_s: off_s + n]
            dst = torch.zeros(n + 8, device=dev, dtype=torch.bfloat16)[off_d: off_d + n]
            want = src.clone()
        else:
            dst = torch.randn(n + 8, device=dev, generator=gen)[off_d: off_d + n]
            bf = kind in (_lib.XFER_SET_BF16, _lib.XFER_ADD_BF16)
            src = torch.randn(n + 8, device=dev, generator=gen)
            src = (src.to(torch.bfloat16) if bf else src)[off_s: off_s + n]
            if kind == _lib.XFER_ZERO:
                want = torch.zeros_like(dst)
            elif kind in (_lib.XFER_SET_BF16, _lib.XFER_SET_F32):
                want = src.float()
            else:
                want = dst + src.float()
        out.append((dst, src, want))
    return out


@pytest.mark.parametrize("misalign", [False, True])
def test_xfer_batch_every_kind_bit_exact(hip_device, misalign):
    from pointcloudmatters_amd import _lib

    gen = torch.Generator(device=hip_device).manual_seed(3)
    sizes = [1, 7, 8, 9, 63, 2047, 8191, 8192, 8193, 3 * 8192 + 5, 512 * 512, 100_003]
    cases, jobs = [], []
    for kind in range(6):
        for dst, src, want in _jobs(hip_device, gen, sizes, kind, misalign):
            cases.append((dst, want, kind))
            jobs.append((dst.data_ptr(), 0 if kind == _lib.XFER_ZERO else src.data_ptr(), dst.numel(), kind))
            cases[-1] += (src,)  # keep the source alive
    assert len(jobs) < 96
    _lib.xfer_batch(jobs)
    torch.cuda.synchronize()
    for dst, want, kind, _ in cases:
        assert torch.equal(dst, want), (kind, dst.numel())


def test_xfer_batch_more_jobs_than_one_launch_and_empty_jobs(hip_device):
    from pointcloudmatters_amd import _lib

    gen = torch.Generator(device=hip_device).manual_seed(5)
    flat = torch.randn(300 * 1000, device=hip_device, generator=gen)
    before = flat.clone()
    srcs = [torch.randn(1000 - (i % 7), device=hip_device, generator=gen).to(torch.bfloat16) for i in range(300)]
    jobs = [(flat.data_ptr() + 4000 * i, s.data_ptr(), s.numel(), _lib.XFER_ADD_BF16) for i, s in enumerate(srcs)]
    jobs.insert(17, (flat.data_ptr(), srcs[0].data_ptr(), 0, _lib.XFER_SET_BF16))  # an empty job is skipped
    _lib.xfer_batch(jobs)
    _lib.xfer_batch([])
    torch.cuda.synchronize()
    for i, s in enumerate(srcs):
        n = s.numel()
        assert torch.equal(flat[1000 * i: 1000 * i + n], before[1000 * i: 1000 * i + n] + s.float())
        assert torch.equal(flat[1000 * i + n: 1000 * (i + 1)], before[1000 * i + n: 1000 * (i + 1)])  # the gap between jobs is untouched


def test_xfer_batch_rejects_bad_arguments(hip_device):
    import ctypes

    from pointcloudmatters_amd import _lib

    L = _lib.load()
    t = torch.zeros(16, device=hip_device)
    P, Lg, I = ctypes.c_void_p * 1, ctypes.c_long * 1, ctypes.c_int * 1
    assert L.pcm_xfer_batch_hip(1, P(t.data_ptr()), P(t.data_ptr()), Lg(16), I(9), None) == 1       # unknown kind
    assert L.pcm_xfer_batch_hip(1, P(t.data_ptr()), P(None), Lg(16), I(_lib.XFER_SET_F32), None) == 1  # missing source
    assert L.pcm_xfer_batch_hip(1, P(t.data_ptr()), P(None), Lg(-1), I(0), None) == 1
    assert L.pcm_xfer_batch_hip(0, None, None, None, None, None) == 0


@pytest.mark.parametrize("first", [True, False])
def test_collect_matches_the_multi_tensor_route(hip_device, first):
    """FlatAdamW.collect (bf16 mirror mode): gradients of every kind -- bf16 stash, fp32 .grad, none at all (zero run), a
    non-contiguous one (fallback) -- land in flat_g exactly as copies / adds per tensor would put them."""
    from pointcloudmatters_amd.bc.flat_optim import FlatAdamW

    torch.manual_seed(0)
    shapes = [(512, 512), (1536, 512), (512,), (7,), (32, 512), (3, 5), (100_001,), (64, 64), (9,)]
    params = [torch.nn.Parameter(torch.randn(*s, device=hip_device)) for s in shapes]
    opt = FlatAdamW(params, None, weight_decay=0.0)
    opt._fixed_lr = 1e-3
    opt.enable_bf16_mirror({id(params[i]) for i in (0, 1, 4, 6)})
    opt.flat_g.normal_()
    before = opt.flat_g.clone()
    want = before.clone() if not first else torch.zeros_like(before)
    if first:
        want.copy_(before)  # slots outside every parameter (alignment padding) keep their contents
    grads = {}
    for k, p in enumerate(params):
        if k in (3, 5):  # no gradient this step
            if first:
                want[opt.offsets[k]: opt.offsets[k] + p.numel()] = 0
            continue
        if opt.shadow[k] is not None:
            g = torch.randn_like(p).to(torch.bfloat16)
            opt.stash_grad(k, g)
        elif k == 7:  # non-contiguous fp32 gradient
            g = torch.randn(64, 64, device=hip_device).t()
            p.grad = g
        else:
            g = torch.randn_like(p)
            p.grad = g
        grads[k] = g
        sl = slice(opt.offsets[k], opt.offsets[k] + p.numel())
        want[sl] = g.float().reshape(-1) if first else before[sl] + g.float().reshape(-1)
    opt.collect(first=first)
    torch.cuda.synchronize()
    assert torch.equal(opt.flat_g, want)
    assert all(s is None for s in opt._stash) and all(p.grad is None for p in params)
