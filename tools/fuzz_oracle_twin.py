"""C oracle (oracle/pcm_oracle.c) against its independent Python twin (oracle/py_twin.py) on random layouts -- ragged clouds of 1..300
points, M > N, lattice / duplicate clouds (exact distance ties), k in 1..32, ball radii with and without an inner radius, random-order
ball query -- for a wall-clock budget (default 240 s).  CPU only.  Round 4: 1007 cases, 0 mismatches.  A 24-case seeded slice of the same
sweep runs in the suite (tests/test_oracle.py::test_oracle_equals_twin_on_a_random_sweep).
    python tools/fuzz_oracle_twin.py [seconds]"""
import numpy as np, torch, time, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pointops_cpu as po, py_twin as tw
from tests.util import make_clouds, new_offsets
rng = np.random.default_rng(7)
t0 = time.time(); n = 0; bad = 0
BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
while time.time() - t0 < BUDGET:
    b = int(rng.integers(1, 4))
    sizes = [int(rng.choice([1, 2, 3, 7, 31, 33, 64, 65, 100, 129, 200, 257, 300])) for _ in range(b)]
    ms = [int(rng.integers(1, max(2, s + 3))) for s in sizes]
    mode = str(rng.choice(["uniform", "lattice", "dup"]))
    seed = int(rng.integers(0, 1 << 30))
    xyz, off = make_clouds(sizes, seed=seed, mode=mode, lattice=float(rng.choice([0.05, 0.1, 0.2])))
    noff = new_offsets(ms)
    a = po.farthest_point_sampling(xyz, off, noff).numpy()
    bb = tw.fps(xyz.numpy(), off.numpy(), noff.numpy())
    ok = np.array_equal(a, bb)
    q = xyz[torch.from_numpy(a).long()].contiguous()
    k = int(rng.choice([1, 3, 8, 16, 32]))
    i1, d1 = po.knn_query_raw(k, xyz, off, q, noff)
    i2, d2 = tw.knn(k, xyz.numpy(), q.numpy(), off.numpy(), noff.numpy())
    ok &= np.array_equal(i1.numpy(), i2) and np.array_equal(d1.numpy(), d2)
    r = float(rng.choice([0.05, 0.1, 0.2, 0.5])); rmin = float(rng.choice([0.0, 0.01, 0.03]))
    i1, d1 = po.ball_query_raw(k, r, rmin, xyz, off, q, noff)
    i2, d2 = tw.ball(k, rmin, r, xyz.numpy(), q.numpy(), off.numpy(), noff.numpy())
    ok &= np.array_equal(i1.numpy(), i2) and np.array_equal(d1.numpy(), d2)
    order = po.make_random_order(off, generator=torch.Generator().manual_seed(seed & 0xffff))
    i1, d1 = po.random_ball_query_raw(k, r, rmin, xyz, off, q, noff, order)
    i2, d2 = tw.random_ball(k, rmin, r, order.numpy(), xyz.numpy(), q.numpy(), off.numpy(), noff.numpy())
    ok &= np.array_equal(i1.numpy(), i2) and np.array_equal(d1.numpy(), d2)
    n += 1
    if not ok:
        bad += 1; print("MISMATCH", sizes, ms, mode, seed, k, r, rmin, flush=True)
print("cases", n, "bad", bad)
