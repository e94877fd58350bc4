"""Seeded synthetic clouds in the packed (n,3)+offset layout."""
import numpy as np
import torch


def make_clouds(sizes, seed=0, mode="uniform", dup_frac=0.05, lattice=0.005):
    """mode: 'uniform' table-top extent; 'lattice' snaps to a grid (forces exact distance ties);
    'dup' additionally duplicates dup_frac of the points."""
    rng = np.random.default_rng(seed)
    parts = []
    for n in sizes:
        p = np.empty((n, 3), dtype=np.float32)
        p[:, 0:2] = rng.uniform(-0.4, 0.4, (n, 2))
        p[:, 2] = rng.uniform(0.005, 0.4, n)
        if mode in ("lattice", "dup"):
            p = (np.round(p / lattice) * lattice).astype(np.float32)
        if mode == "dup" and n > 1:
            k = max(1, int(n * dup_frac))
            src = rng.integers(0, n, k)
            dst = rng.integers(0, n, k)
            p[dst] = p[src]
        parts.append(p.astype(np.float32))
    xyz = np.concatenate(parts) if parts else np.zeros((0, 3), np.float32)
    offset = np.cumsum(sizes).astype(np.int32)
    return torch.from_numpy(xyz).contiguous(), torch.from_numpy(offset)


def new_offsets(ms):
    return torch.from_numpy(np.cumsum(ms).astype(np.int32))


def seeded_fill(module, seed, bn_stats=False, skip=("normalizer.",)):
    """Deterministic weights WITHOUT storing them: every parameter is filled from its own NumPy stream, seeded by
    (seed, crc32 of the parameter's state-dict name), so two modules with the same parameter names -- the reference class in
    the build container (tests/golden/make_golden.py) and the product class on the GPU box -- end up with identical
    weights whatever their construction order.  Matrices ~ N(0, 1/fan_in), norm weights +-(1 + 0.1 N), biases / vectors 0.05 N,
    embeddings N(0, 0.5).  With `bn_stats` the BatchNorm running statistics get seeded non-trivial values too (eval tests).
    Returns a float64 checksum (sum over parameters of sum |w| * (1 + index mod 7)) that fixtures store and tests re-check."""
    import zlib

    check = 0.0
    with torch.no_grad():
        items = list(module.named_parameters())
        if bn_stats:
            items += [(n, b) for n, b in module.named_buffers() if n.endswith(("running_mean", "running_var"))]
        for name, p in items:
            if p.numel() == 0 or name.startswith(tuple(skip)):  # the Diffusion Policy's data normaliser is fitted, not trained
                continue
            rng = np.random.default_rng([int(seed), zlib.crc32(name.encode())])
            v = rng.standard_normal(tuple(p.shape))
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "running_var":
                v = 0.5 + rng.random(tuple(p.shape))
            elif leaf == "running_mean":
                v = 0.1 * v
            elif p.dim() >= 2 and "embed" in name and leaf == "weight":
                v = 0.5 * v
            elif p.dim() >= 2:
                fan_in = int(np.prod(p.shape[1:]))
                v = v / np.sqrt(fan_in)
            elif leaf == "weight":  # BatchNorm / LayerNorm / GroupNorm scale; one in five negative (the fused SA layer's min branch)
                v = (1.0 + 0.1 * v) * np.where(rng.random(tuple(p.shape)) < 0.2, -1.0, 1.0)
            else:
                v = 0.05 * v
            p.copy_(torch.from_numpy(v.astype(np.float32)).to(p.device))
            flat = np.abs(v.astype(np.float32).astype(np.float64)).ravel()
            check += float((flat * (1 + (np.arange(flat.size) % 7))).sum())
    return check


def grad_digest(name, g):
    """A gradient in a form small enough to commit: tensors of <= 16384 elements whole; larger matrices as their first four rows
    and columns, two seeded random projections (g v and u g, float64), the Frobenius norm and the largest magnitude.  Any
    structured error (a missing term, a sign, a transposed block, a wrong row) moves the projections; element noise does not."""
    import zlib

    g = np.asarray(g, dtype=np.float32)
    if g.size <= 16384:
        return {"full": g}
    g2 = g.reshape(g.shape[0], -1)
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    u, v = rng.standard_normal(g2.shape[0]), rng.standard_normal(g2.shape[1])
    g64 = g2.astype(np.float64)
    return {"rows": g2[:4].copy(), "cols": g2[:, :4].copy(), "right": g64 @ v, "left": u @ g64,
            "fro": np.array(np.sqrt((g64 * g64).sum())), "absmax": np.array(np.abs(g2).max())}


def check_grad_digest(name, got, ref, rtol=1e-4, atol=1e-6):
    """`got`: the full gradient (ndarray); `ref`: dict of the digest's parts as stored.  Element parts are held to
    rtol * max|g| (+ atol); a projection over d elements to rtol * max|g| * sqrt(d) (element errors of random sign add up like
    that; a systematic error grows like d); the norm to rtol.  Returns the worst ratio error / bound (for reporting)."""
    dig = grad_digest(name, got)
    worst = 0.0
    if "full" in ref:
        scale = float(np.abs(ref["full"]).max())
        err = float(np.abs(dig["full"] - ref["full"]).max()) if ref["full"].size else 0.0
        bound = rtol * scale + atol
        assert err <= bound, (name, "full", err, bound)
        return err / bound
    scale = float(ref["absmax"])
    rows, cols = ref["cols"].shape[0], ref["rows"].shape[1]
    for part, dim in (("rows", 1), ("cols", 1), ("right", cols), ("left", rows)):
        err = float(np.abs(dig[part] - ref[part]).max())
        bound = (rtol * scale + atol) * np.sqrt(dim)
        assert err <= bound, (name, part, err, bound)
        worst = max(worst, err / bound)
    assert abs(float(dig["fro"]) - float(ref["fro"])) <= rtol * float(ref["fro"]) + atol, (name, "fro")
    return worst


def digest_rel_error(name, got, ref):
    """max |got - ref| / max |ref| over the parts of a grad_digest (projections scaled by 1 / sqrt(d)); (error, max |ref|)."""
    dig = grad_digest(name, got)
    if "full" in ref:
        s = float(np.abs(ref["full"]).max()) if ref["full"].size else 0.0
        e = float(np.abs(dig["full"] - ref["full"]).max()) if ref["full"].size else 0.0
        return e / (s + 1e-30), s
    s = float(ref["absmax"])
    rows, cols = ref["cols"].shape[0], ref["rows"].shape[1]
    e = max(float(np.abs(dig["rows"] - ref["rows"]).max()), float(np.abs(dig["cols"] - ref["cols"]).max()),
            float(np.abs(dig["right"] - ref["right"]).max()) / np.sqrt(cols), float(np.abs(dig["left"] - ref["left"]).max()) / np.sqrt(rows))
    return e / (s + 1e-30), s


def optimizer_zoo():
    """A module whose parameter NAMES and shapes cover every case of the reference's weight-decay grouping rule
    (src/utils/optimizer.py:152-170): matrices, biases, 1-d norm weights, convolution kernels, an embedding table, bare
    parameters of 1 and 3 dimensions, a parameter whose name merely ENDS in '.bias' but is a matrix, and a frozen one.
    Used by tests/golden/make_golden.py::golden_optim (through the reference's build_optimizer / build_optimizer_v2) and by
    tests/test_optim_ref.py (through the trainer's grouping)."""
    import torch.nn as nn

    class Odd(nn.Module):
        def __init__(self):
            super().__init__()
            self.bias = nn.Parameter(torch.zeros(4, 4))  # name 'odd.bias', 2-d: the name rule wins

    class Zoo(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(6, 8)
            self.lin_nobias = nn.Linear(8, 8, bias=False)
            self.ln = nn.LayerNorm(8)
            self.bn = nn.BatchNorm1d(8)
            self.gn = nn.GroupNorm(2, 8)
            self.conv = nn.Conv1d(8, 8, 3)
            self.emb = nn.Embedding(5, 8)
            self.pos_table = nn.Parameter(torch.zeros(1, 3, 8))
            self.scale = nn.Parameter(torch.ones(8))
            self.odd = Odd()
            self.frozen = nn.Linear(8, 2)
            for p in self.frozen.parameters():
                p.requires_grad_(False)

    torch.manual_seed(0)
    return Zoo()
