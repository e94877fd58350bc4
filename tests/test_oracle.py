"""CPU: the C oracle against its independent Python twin (bit-exact) and against properties that
hold for any correct implementation of the reference kernels."""
import numpy as np
import pytest
import torch

from oracle import lib as olib
from oracle import pointops_cpu as po
from oracle import py_twin as tw
from tests.util import make_clouds, new_offsets

CASES = [
    ([100, 37, 260], [20, 50, 64], "uniform"),
    ([5, 3], [8, 2], "uniform"),          # M > N
    ([300], [64], "lattice"),             # forced ties
    ([130, 64, 1], [30, 30, 1], "dup"),
]


def test_opt_n_threads_matches_reference_formula():
    L = olib.load()
    for n in list(range(1, 70)) + [127, 128, 129, 255, 256, 511, 512, 513, 1023, 1024, 1025, 4096, 8192, 100000]:
        assert L.pcm_opt_n_threads_cpu(n) == tw.opt_n_threads(n)
    for k in range(0, 11):  # exact powers of two must not fall one short
        assert L.pcm_opt_n_threads_cpu(1 << k) == (1 << k)
    assert L.pcm_opt_n_threads_cpu(5000) == 1024


@pytest.mark.parametrize("sizes,ms,mode", CASES)
def test_fps_c_equals_twin(sizes, ms, mode):
    xyz, off = make_clouds(sizes, seed=1, mode=mode, lattice=0.05)
    noff = new_offsets(ms)
    a = po.farthest_point_sampling(xyz, off, noff).numpy()
    b = tw.fps(xyz.numpy(), off.numpy(), noff.numpy())
    assert np.array_equal(a, b)


@pytest.mark.parametrize("sizes,ms,mode", CASES)
def test_queries_c_equal_twin(sizes, ms, mode):
    xyz, off = make_clouds(sizes, seed=2, mode=mode, lattice=0.05)
    noff = new_offsets(ms)
    sel = po.farthest_point_sampling(xyz, off, noff).long()
    q = xyz[sel].contiguous()
    i1, d1 = po.knn_query_raw(16, xyz, off, q, noff)
    i2, d2 = tw.knn(16, xyz.numpy(), q.numpy(), off.numpy(), noff.numpy())
    assert np.array_equal(i1.numpy(), i2) and np.array_equal(d1.numpy(), d2)
    i1, d1 = po.ball_query_raw(8, 0.2, 0.01, xyz, off, q, noff)
    i2, d2 = tw.ball(8, 0.01, 0.2, xyz.numpy(), q.numpy(), off.numpy(), noff.numpy())
    assert np.array_equal(i1.numpy(), i2) and np.array_equal(d1.numpy(), d2)
    order = po.make_random_order(off, generator=torch.Generator().manual_seed(0))
    i1, d1 = po.random_ball_query_raw(8, 0.2, 0.01, xyz, off, q, noff, order)
    i2, d2 = tw.random_ball(8, 0.01, 0.2, order.numpy(), xyz.numpy(), q.numpy(), off.numpy(), noff.numpy())
    assert np.array_equal(i1.numpy(), i2) and np.array_equal(d1.numpy(), d2)


def test_fps_properties():
    sizes, ms = [600, 350, 1024], [100, 80, 200]
    xyz, off = make_clouds(sizes, seed=3)
    noff = new_offsets(ms)
    idx = po.farthest_point_sampling(xyz, off, noff).numpy()
    starts = [0] + off.tolist()[:-1]
    mstarts = [0] + noff.tolist()[:-1]
    pts = xyz.numpy().astype(np.float64)
    for c in range(len(sizes)):
        sel = idx[mstarts[c]: noff[c]]
        assert sel[0] == starts[c]                                  # first pick = first point
        assert sel.min() >= starts[c] and sel.max() < off[c]        # stays inside its cloud
        assert len(set(sel.tolist())) == len(sel)                   # no repeats while M <= N (distinct points)
        cloud = pts[starts[c]: off[c]]
        chosen = [sel[0] - starts[c]]
        mind = np.full(len(cloud), np.inf)
        for j in range(1, len(sel)):                                # each pick maximises the min-distance
            mind = np.minimum(mind, ((cloud - cloud[chosen[-1]]) ** 2).sum(1))
            got = sel[j] - starts[c]
            assert mind[got] >= mind.max() * (1 - 1e-5)
            chosen.append(got)


def test_knn_equals_bruteforce_without_ties():
    xyz, off = make_clouds([500, 300], seed=4)
    idx, d2 = po.knn_query_raw(16, xyz, off)
    starts = [0, 500]
    for q in range(0, 800, 37):
        c = 0 if q < 500 else 1
        cloud = xyz[starts[c]: off[c]]
        dd = ((cloud - xyz[q]) ** 2).sum(1)
        order = torch.argsort(dd, stable=True)[:16] + starts[c]
        assert idx[q].tolist() == order.tolist()
        assert torch.all(d2[q][1:] >= d2[q][:-1])
        assert idx[q, 0] == q


def test_knn_padding_when_cloud_smaller_than_nsample():
    xyz, off = make_clouds([5, 40], seed=5)
    idx, dist = po.knn_query(16, xyz, off)
    assert (idx[:5, 5:] == -1).all() and (idx[:5, :5] >= 0).all()
    assert torch.allclose(dist[:5, 5:], torch.full((5, 11), 1e5))     # sqrt(1e10)


def test_ball_query_quirks():
    xyz, off = make_clouds([400], seed=6)
    idx, d2 = po.ball_query_raw(8, 0.5, 0.0, xyz, off)              # many candidates -> subsample branch
    assert torch.equal(d2, idx.float())                             # dist2 := index (ball_query_cuda_kernel.cu:120)
    idx, d2 = po.ball_query_raw(64, 0.03, 0.0, xyz, off)            # few candidates -> padded
    assert ((idx == -1) == (d2 == 1e10)).all() and (idx[:, 0] >= 0).all()


def test_ball_query_overflow_is_reported():
    L = olib.load()
    xyz, off = make_clouds([2100], seed=7)
    idx = torch.zeros(2100, 4, dtype=torch.int32)
    d2 = torch.zeros(2100, 4)
    rc = L.pcm_ball_query_cpu(2100, 4, 0.0, 10.0, xyz.data_ptr(), xyz.data_ptr(), off.data_ptr(), off.data_ptr(),
                              idx.data_ptr(), d2.data_ptr())
    assert rc == 2 and (idx == -1).all()


def test_gather_ops_against_torch():
    g = torch.Generator().manual_seed(0)
    n, m, k, c = 40, 25, 6, 7
    feat = torch.randn(n, c, generator=g, requires_grad=True)
    idx = torch.randint(0, n, (m, k), generator=g, dtype=torch.int32)
    out = po.grouping2(feat, idx)
    want = feat[idx.long()]
    assert torch.equal(out, want)
    gout = torch.randn(m, k, c, generator=g)
    (g1,) = torch.autograd.grad(out, feat, gout)
    (g2,) = torch.autograd.grad(want, feat, gout)
    torch.testing.assert_close(g1, g2, rtol=1e-5, atol=1e-6)
    # subtraction
    a = torch.randn(m, c, generator=g, requires_grad=True)
    b = torch.randn(m, c, generator=g, requires_grad=True)
    sidx = torch.randint(0, m, (m, k), generator=g, dtype=torch.int32)
    out = po.subtraction(a, b, sidx)
    want = a[:, None, :] - b[sidx.long()]
    assert torch.equal(out, want)
    gout = torch.randn(m, k, c, generator=g)
    ga = torch.autograd.grad(out, (a, b), gout)
    gb = torch.autograd.grad(want, (a, b), gout)
    for x, y in zip(ga, gb):
        torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-5)


def test_interpolation_and_aggregation_against_torch():
    xyz, off = make_clouds([60, 50], seed=8)
    new_xyz, noff = make_clouds([30, 20], seed=9)
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(110, 5, generator=g, requires_grad=True)
    out = po.interpolation2(xyz, new_xyz, feat, off, noff, 3)
    want = po.interpolation(xyz, new_xyz, feat, off, noff, 3)
    torch.testing.assert_close(out, want, rtol=1e-6, atol=1e-6)
    gout = torch.randn(out.shape, generator=g)
    (g1,) = torch.autograd.grad(out, feat, gout)
    (g2,) = torch.autograd.grad(want, feat, gout)
    torch.testing.assert_close(g1, g2, rtol=1e-5, atol=1e-6)
    # aggregation: out[n,c] = sum_s (in[idx] + pos) * w[.., c % w_c]
    n, s, c, wc = 20, 4, 6, 3
    inp = torch.randn(n, c, generator=g, requires_grad=True)
    pos = torch.randn(n, s, c, generator=g, requires_grad=True)
    w = torch.randn(n, s, wc, generator=g, requires_grad=True)
    idx = torch.randint(0, n, (n, s), generator=g, dtype=torch.int32)
    out = po.aggregation(inp, pos, w, idx)
    want = ((inp[idx.long()] + pos) * w.repeat(1, 1, c // wc)).sum(1)
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
    gout = torch.randn(n, c, generator=g)
    ga = torch.autograd.grad(out, (inp, pos, w), gout)
    gb = torch.autograd.grad(want, (inp, pos, w), gout)
    for x, y in zip(ga, gb):
        torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-5)


def test_attention_steps_against_torch():
    g = torch.Generator().manual_seed(2)
    n, m, gg, c = 15, 40, 2, 4
    q = torch.randn(n, gg, c, generator=g, requires_grad=True)
    k = torch.randn(n, gg, c, generator=g, requires_grad=True)
    w = torch.ones(c)
    it = torch.randint(0, n, (m,), generator=g, dtype=torch.int32)
    ir = torch.randint(0, n, (m,), generator=g, dtype=torch.int32)
    out = po.attention_relation_step(q, k, w, it, ir)
    want = (q[it.long()] * k[ir.long()] * w).sum(-1)
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-6)
    gout = torch.randn(m, gg, generator=g)
    for x, y in zip(torch.autograd.grad(out, (q, k), gout), torch.autograd.grad(want, (q, k), gout)):
        torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-5)
    aw = torch.randn(m, gg, generator=g, requires_grad=True)
    v = torch.randn(n, gg, c, generator=g, requires_grad=True)
    out = po.attention_fusion_step(aw, v, it, ir)
    want = torch.zeros(n, gg, c).index_add(0, it.long(), aw[:, :, None] * v[ir.long()])
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
    gout = torch.randn(n, gg, c, generator=g)
    for x, y in zip(torch.autograd.grad(out, (aw, v), gout), torch.autograd.grad(want, (aw, v), gout)):
        torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-5)


def test_oracle_equals_twin_on_a_random_sweep():
    """A seeded slice of tools/fuzz_oracle_twin.py (1007 cases without a mismatch in round 4): ragged clouds, M > N, exact ties (lattice,
    duplicates), k in 1..32, inner radii, random-order ball query -- C oracle == Python twin, indices AND distances, bit for bit."""
    rng = np.random.default_rng(11)
    for _ in range(24):
        b = int(rng.integers(1, 4))
        sizes = [int(rng.choice([1, 2, 3, 7, 31, 33, 64, 65, 100, 129, 200])) for _ in range(b)]
        ms = [int(rng.integers(1, max(2, s + 3))) for s in sizes]
        mode = str(rng.choice(["uniform", "lattice", "dup"]))
        seed = int(rng.integers(0, 1 << 30))
        xyz, off = make_clouds(sizes, seed=seed, mode=mode, lattice=float(rng.choice([0.05, 0.1, 0.2])))
        noff = new_offsets(ms)
        sel = po.farthest_point_sampling(xyz, off, noff)
        assert np.array_equal(sel.numpy(), tw.fps(xyz.numpy(), off.numpy(), noff.numpy())), (sizes, ms, mode, seed)
        q = xyz[sel.long()].contiguous()
        k = int(rng.choice([1, 3, 8, 16, 32]))
        r, rmin = float(rng.choice([0.05, 0.1, 0.2, 0.5])), float(rng.choice([0.0, 0.01, 0.03]))
        order = po.make_random_order(off, generator=torch.Generator().manual_seed(seed & 0xFFFF))
        for got, want in (
                (po.knn_query_raw(k, xyz, off, q, noff), tw.knn(k, xyz.numpy(), q.numpy(), off.numpy(), noff.numpy())),
                (po.ball_query_raw(k, r, rmin, xyz, off, q, noff), tw.ball(k, rmin, r, xyz.numpy(), q.numpy(), off.numpy(), noff.numpy())),
                (po.random_ball_query_raw(k, r, rmin, xyz, off, q, noff, order),
                 tw.random_ball(k, rmin, r, order.numpy(), xyz.numpy(), q.numpy(), off.numpy(), noff.numpy()))):
            assert np.array_equal(got[0].numpy(), want[0]) and np.array_equal(got[1].numpy(), want[1]), (sizes, ms, mode, seed, k, r, rmin)
