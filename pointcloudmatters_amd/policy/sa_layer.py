"""The set-abstraction ("pcd_sampling") layer shared by ACTPCD and PCDObsEncoder, and the 3-D sine
position embedding of the tokens.

Reference: /root/reference/src/models/components/act/act.py:384-465 (twin:
diffusion_policy/vision/pcd_obs_encoder.py:123-198) and act.py:467-506.

    FPS(p, o -> n_o)  ->  n_p = p[idx]  ->  kNN(n_p in p, K)  ->  group [rel xyz | feat]  (m, K, 3+C)
    -> Linear(3+C -> H, no bias) -> BatchNorm1d(H) over the m*K rows -> ReLU -> max over K  -> (m, H)

`owner` is the module that holds ``linear``, ``bn``, ``pool``, ``relu``, ``pcd_nsample`` under the
reference's attribute names (so state-dict keys match).  Implementations:

  "reference"  the reference's literal op sequence (materialises (m,K,3+C) and (m,H,K));
  "torch"      same op order, BN applied on the (m*K, H) view (identical statistics) and a plain
               max over K -- skips the transpose/contiguous copy;
  "fused"      HIP kernels of policy/sa_fused.py: one (n,3+C)x(3+C,H) GEMM + gather/BN/ReLU/max
               passes that never materialise the grouped tensor (fp32 sums re-associated: within
               1e-4 relative of "reference", not bit-identical).
"""
import torch


def coord_embedding_sine(coord, hidden_dim, temperature=10000, normalize=False, scale=None):
    """act.py:467-506: per axis H//3 features -- with the default arguments the sine BLOCK followed by the cosine block (the
    reference's x_embed is (m, 1) there; with `normalize` it is (m,) and the two interleave) --, zero-padded to H.
    On the HIP device the default-argument form is one kernel (csrc/tokens.hip) instead of ~18 framework launches."""
    npf = hidden_dim // 3
    pad = hidden_dim - npf * 3
    if scale is not None and normalize is False:
        raise ValueError("normalize should be True if scale is passed")
    if scale is None:
        scale = 2 * torch.pi
    if coord.is_cuda and not normalize and npf > 0 and npf % 2 == 0 and coord.dtype == torch.float32 and coord.dim() == 2 \
            and coord.shape[1] == 3 and not (torch.is_grad_enabled() and coord.requires_grad) \
            and ((npf, float(temperature), coord.device) in _DIM_T or not torch.cuda.is_current_stream_capturing()):
        return _coord_embedding_sine_hip(coord, hidden_dim, npf, temperature)
    axes = [coord[:, 0:1], coord[:, 1:2], coord[:, 2:3]]
    if normalize:
        eps = 1e-6
        axes = [coord[:, i] / (coord[:, i].max() + eps) * scale for i in range(3)]
    dim_t = torch.arange(npf, dtype=torch.float32, device=coord.device)
    dim_t = temperature ** (2 * (dim_t // 2) / npf)
    parts = []
    for a in axes:
        ang = a[..., None] / dim_t  # (m, 1, npf)
        parts.append(torch.stack((ang[..., 0::2].sin(), ang[..., 1::2].cos()), dim=2).flatten(1))
    pos = torch.cat(parts, dim=1)
    if pad:
        pos = torch.cat((pos, pos.new_zeros(pos.shape[0], pad)), dim=1)
    return pos


_DIM_T = {}


def _coord_embedding_sine_hip(coord, hidden_dim, npf, temperature):
    from .. import _lib

    key = (npf, float(temperature), coord.device)
    dim_t = _DIM_T.get(key)
    if dim_t is None:  # the reference's own expression, evaluated once per (npf, temperature, device)
        dim_t = torch.arange(npf, dtype=torch.float32, device=coord.device)
        dim_t = _DIM_T[key] = (temperature ** (2 * (dim_t // 2) / npf)).contiguous()
    coord = coord.contiguous()
    out = torch.empty(coord.shape[0], hidden_dim, dtype=torch.float32, device=coord.device)
    with torch.cuda.device(coord.device):
        rc = _lib.load().pcm_coord_embed_sine_hip(coord.shape[0], hidden_dim, npf, coord.data_ptr(), dim_t.data_ptr(), out.data_ptr(),
                                                  _lib.raw_stream())
    _lib.check(rc, "pcm_coord_embed_sine_hip")
    return out


def masked_fps(owner, pointops, p, o, n_o, mask):
    """The ``use_mask`` branch of ``pcd_sampling`` (reference act.py:394-442, pcd_obs_encoder.py:131-180), literally:
    FPS runs over the foreground subset ``p[mask]`` (and, with ``bg_ratio`` > 0, over the background subset for the
    last ``int(npoints * bg_ratio)`` samples of every cloud), and the returned indices are LOCAL TO THOSE SUBSETS,
    foreground block first -- exactly what the reference then uses to index the unmasked cloud.  The per-cloud subset
    sizes come from one cumulative sum instead of the reference's per-cloud ``.item()`` loop; the boolean gather still
    needs its size on the host, so this branch cannot be captured into a graph."""
    if p.is_cuda and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("use_mask sampling has data-dependent sizes and cannot run under graph capture")
    mask = mask.bool()
    b = o.shape[0]
    last = o.long() - 1
    n_bg = int(owner.pcd_npoints * owner.bg_ratio) if owner.bg_ratio > 0.0 else 0
    steps = torch.arange(1, b + 1, device=o.device, dtype=torch.int32)
    fg_n_o = steps * (owner.pcd_npoints - n_bg) if n_bg else n_o
    fg_o = torch.cumsum(mask, 0, dtype=torch.int32)[last]
    idx = pointops.farthest_point_sampling(p[mask], fg_o, fg_n_o)
    if n_bg:
        bg_o = torch.cumsum(~mask, 0, dtype=torch.int32)[last]
        idx = torch.cat([idx, pointops.farthest_point_sampling(p[~mask], bg_o, steps * n_bg)], dim=0)
    return idx


def sample_and_query(owner, pointops, p, o, n_o, overlap=False, mask=None):
    """Coordinate-only part of the layer: FPS indices, sampled centres, kNN lists.

    With ``overlap`` the three launches go to a side HIP stream so they run concurrently with
    whatever the caller enqueues next on the current stream (the PointNet MLP); ``wait()`` joins.
    """
    nsample = owner.pcd_nsample
    masked = bool(getattr(owner, "use_mask", False)) and mask is not None
    static = owner.__dict__.get("_static_pre")
    if static is not None and static["key"] == _key(p, o):
        return static["pre"]  # graph mode: the indices live in static buffers that the trainer fills before each replay
    ready = owner.__dict__.get("_prefetched")
    if ready:  # indices computed ahead of time for exactly these tensors (prefetch_sampling)
        hit = ready.pop((p.data_ptr(), tuple(p.shape), o.data_ptr()), None)
        if hit is not None:
            return hit[0]

    want_stats = getattr(owner, "sa_impl", None) == "fused" and p.is_cuda

    def run():
        with torch.no_grad():
            if masked:
                idx = masked_fps(owner, pointops, p, o, n_o, mask)
            else:
                idx = pointops.farthest_point_sampling(p, o, n_o)  # (m) int32
            n_p = p[idx.long(), :]  # (m, 3)
            knn_idx, _ = pointops.knn_query(nsample, p, o, n_p, n_o)
            istats = None
            if want_stats:  # index-only half of the fused layer's backward: coordinates only, so it rides along here
                from .sa_fused import index_stats, layout_of

                istats = index_stats(p, n_p, knn_idx, *layout_of(o, n_o))
        return idx, n_p, knn_idx, istats

    if overlap and p.is_cuda:
        from .. import _graphs

        overlap = not _graphs.cuts()  # no forked stream across the cuts of a segmented capture (see ACTPCD.forward)
    if not (overlap and p.is_cuda):
        idx, n_p, knn_idx, istats = run()
        return {"idx": idx, "n_p": n_p, "knn_idx": knn_idx, "istats": istats, "event": None}
    main = torch.cuda.current_stream(p.device)
    side = getattr(owner, "_side_stream", None)
    if side is None:
        side = torch.cuda.Stream(device=p.device)
        owner._side_stream = side
    side.wait_stream(main)  # coordinates must be materialised
    with torch.cuda.stream(side):
        idx, n_p, knn_idx, istats = run()
        event = side.record_event()
    if not torch.cuda.is_current_stream_capturing():
        for t in (idx, n_p, knn_idx) + (istats or ()):
            t.record_stream(main)
    return {"idx": idx, "n_p": n_p, "knn_idx": knn_idx, "istats": istats, "event": event}


def _key(p, o):
    return (p.data_ptr(), tuple(p.shape), o.data_ptr())


def _pre_tensors(pre):
    return [pre["idx"], pre["n_p"], pre["knn_idx"]] + list(pre.get("istats") or ())


def install_static(owner, p, o, pre):
    """Graph mode with the sampling kept OUTSIDE the captured graph: `pre` (computed for the static coordinate buffers
    p, o) becomes the set of static index buffers that every later sample_and_query(owner, p, o) returns."""
    owner._static_pre = {"key": _key(p, o), "pre": dict(pre, event=None)}


def load_static(owner, pre):
    """Copy another batch's sampling result into the static index buffers (one table-driven copy launch on the current
    stream, after waiting for the side stream that produced it)."""
    static = owner._static_pre["pre"]
    if pre is static:
        return
    if pre.get("event") is not None:
        torch.cuda.current_stream(static["idx"].device).wait_event(pre["event"])
    from .. import _lib

    _lib.copy_batch(list(zip(_pre_tensors(static), _pre_tensors(pre))))


def prefetch_sampling(owner, pointops, p, o, n_o, mask=None):
    """FPS + kNN depend on the coordinates only, i.e. on the INPUT batch, not on any weight: like a data-loader worker they
    can run one batch ahead.  Launches them now on the side stream for the coordinates of a FUTURE batch; the next
    ``sample_and_query`` called with the same tensors picks the result up instead of computing it on the critical path.
    Every batch's indices are still computed exactly once."""
    if not p.is_cuda or torch.cuda.is_current_stream_capturing():
        return
    key = (p.data_ptr(), tuple(p.shape), o.data_ptr())
    ready = owner.__dict__.setdefault("_prefetched", {})
    if key not in ready:
        ready[key] = (sample_and_query(owner, pointops, p, o, n_o, overlap=True, mask=mask), p, o)  # p, o kept alive with the result


def set_abstraction(owner, pointops, p, x, o, n_o, impl="reference", pre=None):
    """Returns (n_p (m,3), tokens (m,H), fps_idx (m))."""
    if pre is None:
        pre = sample_and_query(owner, pointops, p, o, n_o, overlap=False)
    if pre["event"] is not None:
        torch.cuda.current_stream(p.device).wait_event(pre["event"])
    idx, n_p, knn_idx = pre["idx"], pre["n_p"], pre["knn_idx"]
    m, k = knn_idx.shape
    if impl == "fused":
        from .sa_fused import sa_fused_forward, supports

        if supports(owner, x):
            tokens = sa_fused_forward(owner, p, x, n_p, idx, knn_idx, o, n_o, istats=pre.get("istats"))
            return n_p, tokens, idx
        if getattr(owner.bn, "_pcm_sync", False) and owner.training:
            raise RuntimeError("this set-abstraction layer's BatchNorm is marked for synchronised statistics, which only the "
                               "fused HIP path implements (policy/sync_bn.py)")
        impl = "torch"  # eval mode / SyncBatchNorm / CPU: same maths through framework ops
    if x.dtype != torch.float32:
        x = x.float()  # pointops is fp32 (bf16 autocast applies to GEMM / attention only)
    grouped, _ = pointops.knn_query_and_group(x, p, offset=o, new_xyz=n_p, new_offset=n_o, idx=knn_idx,
                                              nsample=k, with_xyz=True)  # (m, K, 3+C)
    y = owner.linear(grouped)  # (m, K, H)
    if impl == "reference":
        y = owner.relu(owner.bn(y.transpose(1, 2).contiguous()))  # (m, H, K)
        tokens = owner.pool(y).squeeze(-1)  # (m, H)
    elif impl == "torch":
        h = y.shape[-1]
        y = owner.relu(owner.bn(y.reshape(m * k, h))).view(m, k, h)
        tokens = y.max(dim=1).values
    else:
        raise ValueError(impl)
    return n_p, tokens, idx


set_abstraction.sample_and_query = sample_and_query
set_abstraction.prefetch_sampling = prefetch_sampling
set_abstraction.install_static = install_static
set_abstraction.load_static = load_static
