"""CPU ORACLE twin (test infrastructure): literal pure-Python / NumPy-fp32 emulation of the
reference's K1-K4 CUDA kernels, written independently of pcm_oracle.c and deliberately in the
reference's *thread* order (thread-outer loops, explicit shared-memory arrays).  Small cases only.

tests/test_oracle.py requires this twin and the C oracle to agree bit-for-bit, which is the only
pinning available for the index kernels (the reference has no golden vectors).
Citations are relative to /root/reference/libs/pointops/src/.
"""
import math

import numpy as np

F = np.float32


def opt_n_threads(work_size):
    """cuda_utils.h:11-14."""
    pow_2 = int(math.log(float(work_size)) / math.log(2.0))
    return max(min(1 << pow_2, 1024), 1)


def _sqdist(ax, ay, az, bx, by, bz):
    """(a-b)*(a-b) summed left to right in fp32 without fusion."""
    dx, dy, dz = F(ax - bx), F(ay - by), F(az - bz)
    return F(F(F(dx * dx) + F(dy * dy)) + F(dz * dz))


def fps(xyz, offset, new_offset):
    """sampling/sampling_cuda_kernel.cu:15-129 + functions/sampling.py:8-24."""
    xyz = np.asarray(xyz, dtype=F)
    b = len(offset)
    n_max = int(offset[0])
    for i in range(1, b):
        n_max = max(int(offset[i] - offset[i - 1]), n_max)
    bs = opt_n_threads(n_max)
    idx = np.zeros(int(new_offset[b - 1]), dtype=np.int32)
    tmp = np.full(xyz.shape[0], F(1e10), dtype=F)
    for bid in range(b):
        start_n = 0 if bid == 0 else int(offset[bid - 1])
        end_n = int(offset[bid])
        start_m = 0 if bid == 0 else int(new_offset[bid - 1])
        end_m = int(new_offset[bid])
        if end_m <= start_m:
            continue
        old = start_n
        idx[start_m] = start_n
        for j in range(start_m + 1, end_m):
            dists = [F(-1)] * bs
            dists_i = [start_n] * bs
            x1, y1, z1 = xyz[old]
            for tid in range(bs):  # one CUDA thread at a time
                besti, best = start_n, F(-1)
                for k in range(start_n + tid, end_n, bs):
                    x2, y2, z2 = xyz[k]
                    d = _sqdist(x2, y2, z2, x1, y1, z1)
                    d2 = min(d, tmp[k])
                    tmp[k] = d2
                    if d2 > best:
                        besti, best = k, d2
                dists[tid], dists_i[tid] = best, besti
            s = bs // 2
            while s >= 1:  # the block_size >= 2s ladder, :64-123
                for tid in range(s):
                    v1, v2 = dists[tid], dists[tid + s]
                    i1, i2 = dists_i[tid], dists_i[tid + s]
                    dists[tid] = max(v1, v2)
                    dists_i[tid] = i2 if v2 > v1 else i1
                s //= 2
            old = dists_i[0]
            idx[j] = old
    return idx


def _reheap(dist, idx, k):
    root, child = 0, 1
    while child < k:
        if child + 1 < k and dist[child + 1] > dist[child]:
            child += 1
        if dist[root] > dist[child]:
            return
        dist[root], dist[child] = dist[child], dist[root]
        idx[root], idx[child] = idx[child], idx[root]
        root = child
        child = root * 2 + 1


def _heap_sort(dist, idx, k):
    for i in range(k - 1, 0, -1):
        dist[0], dist[i] = dist[i], dist[0]
        idx[0], idx[i] = idx[i], idx[0]
        _reheap(dist, idx, i)


def _bt(q, off):
    i = 0
    while not q < off[i]:
        i += 1
    return i


def knn(nsample, xyz, new_xyz, offset, new_offset):
    """knn_query/knn_query_cuda_kernel.cu:60-104.  Returns (idx, dist2)."""
    xyz, new_xyz = np.asarray(xyz, dtype=F), np.asarray(new_xyz, dtype=F)
    m = new_xyz.shape[0]
    out_i = np.zeros((m, nsample), dtype=np.int32)
    out_d = np.zeros((m, nsample), dtype=F)
    for q in range(m):
        bt = _bt(q, new_offset)
        start = 0 if bt == 0 else int(offset[bt - 1])
        end = int(offset[bt])
        bd, bi = [F(1e10)] * nsample, [-1] * nsample
        qx, qy, qz = new_xyz[q]
        for i in range(start, end):
            x, y, z = xyz[i]
            d2 = _sqdist(qx, qy, qz, x, y, z)
            if d2 < bd[0]:
                bd[0], bi[0] = d2, i
                _reheap(bd, bi, nsample)
        _heap_sort(bd, bi, nsample)
        out_i[q], out_d[q] = bi, bd
    return out_i, out_d


def _in_ball(d2, min_r2, max_r2):
    return float(d2) <= 1e-5 or (d2 >= min_r2 and d2 < max_r2)


def ball(nsample, min_radius, max_radius, xyz, new_xyz, offset, new_offset):
    """ball_query/ball_query_cuda_kernel.cu:58-123.  Returns (idx, dist2)."""
    xyz, new_xyz = np.asarray(xyz, dtype=F), np.asarray(new_xyz, dtype=F)
    max_r2 = F(F(max_radius) * F(max_radius))
    min_r2 = F(F(min_radius) * F(min_radius))
    m = new_xyz.shape[0]
    out_i = np.zeros((m, nsample), dtype=np.int32)
    out_d = np.zeros((m, nsample), dtype=F)
    for q in range(m):
        bt = _bt(q, new_offset)
        start = 0 if bt == 0 else int(offset[bt - 1])
        end = int(offset[bt])
        cd, ci = [], []
        qx, qy, qz = new_xyz[q]
        for i in range(start, end):
            x, y, z = xyz[i]
            d2 = _sqdist(qx, qy, qz, x, y, z)
            if _in_ball(d2, min_r2, max_r2):
                cd.append(d2)
                ci.append(i)
        num = len(cd)
        assert num <= 2048
        _heap_sort(cd, ci, num)
        if num <= nsample:
            for i in range(num):
                out_i[q, i], out_d[q, i] = ci[i], cd[i]
            for i in range(num, nsample):
                out_i[q, i], out_d[q, i] = -1, F(1e10)
        else:
            sep = F(F(num) / F(nsample))
            for i in range(nsample):
                index = int(F(sep * F(i)))
                out_i[q, i] = ci[index]
                out_d[q, i] = F(ci[index])
    return out_i, out_d


def random_ball(nsample, min_radius, max_radius, order, xyz, new_xyz, offset, new_offset):
    """random_ball_query/random_ball_query_cuda_kernel.cu:58-108.  Returns (idx, dist2)."""
    xyz, new_xyz = np.asarray(xyz, dtype=F), np.asarray(new_xyz, dtype=F)
    max_r2 = F(F(max_radius) * F(max_radius))
    min_r2 = F(F(min_radius) * F(min_radius))
    m = new_xyz.shape[0]
    out_i = np.zeros((m, nsample), dtype=np.int32)
    out_d = np.zeros((m, nsample), dtype=F)
    for q in range(m):
        bt = _bt(q, new_offset)
        start = 0 if bt == 0 else int(offset[bt - 1])
        end = int(offset[bt])
        qx, qy, qz = new_xyz[q]
        cnt = 0
        for i in range(start, end):
            o = int(order[i])
            x, y, z = xyz[o]
            d2 = _sqdist(qx, qy, qz, x, y, z)
            if _in_ball(d2, min_r2, max_r2):
                out_d[q, cnt], out_i[q, cnt] = d2, o
                cnt += 1
                if cnt >= nsample:
                    break
        for i in range(cnt, nsample):
            out_i[q, i], out_d[q, i] = -1, F(1e10)
    return out_i, out_d
