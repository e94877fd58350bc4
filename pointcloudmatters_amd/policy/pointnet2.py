"""Multi-stage set-abstraction encoders composed ONLY from the pointops API and the fused SA kernel.

**No reference counterpart.**  BASELINE.json's configs name "PointNet++ / PointNeXt / PointBERT" backbones, but
/root/reference ships none of them (its point backbone is the per-point PointNet plus ONE
`pcd_sampling` layer).  These modules stack that very layer -- FPS -> kNN -> group [rel xyz | feat] -> Linear ->
BatchNorm -> ReLU -> max over the neighbourhood (act.py:384-465) -- into the hierarchical encoders the configs
mention, so parity here means: the fused HIP path equals the same composition evaluated with the CPU oracle's
pointops (tests/test_pointnet2_gpu.py), nothing more.

  SAStage           one pcd_sampling layer with its own (linear, bn); halves (or sets) the token count
  PointNet2Encoder  PointNet++-SSG style: stages of increasing width; returns the last stage's tokens
  FeaturePropagation  PointNet++ decoder step: inverse-distance interpolation (pointops.interpolation, k = 3)
                      of coarse features onto the fine points + a per-point Linear/BN/ReLU
  PatchTokenizer    PointBERT-style patch tokens: FPS centres + kNN patches encoded by one SA layer, with the
                    centres' 3-D sine position embedding
  SAStageMSG        PointNet++ multi-scale grouping: ONE set of FPS centres, several neighbourhoods (kNN or ball query, each
                    with its own Linear/BN), channel-concatenated
  InvResMLP         PointNeXt's inverted-residual block at fixed resolution: neighbourhood aggregation (the same SA layer with
                    every point as a query) -> pointwise expansion MLP -> residual
  PointNeXtBackbone stem + InvResMLP blocks + head: a per-point backbone for ACTPCD ("PointNeXt + ACT", BASELINE configs[3])
  PatchBertObsEncoder  patch tokens -> transformer encoder -> pooled feature: an observation encoder for the Diffusion
                    Policy ("PointBERT encoder + DiffusionPolicy", BASELINE configs[4])
"""
import torch
import torch.nn as nn

from .pointnet import PointNet
from .rows_linear import linear_rows
from .sa_layer import coord_embedding_sine, set_abstraction
from .rows_linear import RowsLinear


def _query_offsets(owner, o, npoints):
    """n_o = [M, 2M, ...] with its host copy, cached per (clouds, device): building it inside forward would be a pageable
    host-to-device copy in every step, which a hipGraph capture rejects."""
    b = int(o.shape[0])
    cache = owner.__dict__.setdefault("_n_o_cache", {})
    key = (b, o.device)
    if key not in cache:
        host = [npoints * (i + 1) for i in range(b)]
        t = torch.tensor(host, dtype=torch.int32, device=o.device)
        t._pcm_host = host
        cache[key] = t
    return cache[key]


class SAStage(nn.Module):
    def __init__(self, in_channels, out_channels, npoints, nsample=16, pointops=None, sa_impl="fused"):
        super().__init__()
        if pointops is None:
            from .. import pointops as _hip_pointops

            pointops = _hip_pointops
        self._pointops = [pointops]
        self.sa_impl = sa_impl
        self.pcd_nsample, self.pcd_npoints = nsample, npoints
        self.linear = nn.Linear(3 + in_channels, out_channels, bias=False)
        self.bn = nn.BatchNorm1d(out_channels)
        self.pool = nn.MaxPool1d(nsample)
        self.relu = nn.ReLU(inplace=True)

    def new_offsets(self, o):
        return _query_offsets(self, o, self.pcd_npoints)

    def forward(self, p, x, o):
        """(p (n,3), x (n,C), o (b)) -> (n_p (m,3), tokens (m,H), n_o (b))."""
        n_o = self.new_offsets(o)
        n_p, tokens, _ = set_abstraction(self, self._pointops[0], p, x, o, n_o, impl=self.sa_impl)
        return n_p, tokens, n_o


class PointNet2Encoder(nn.Module):
    """stages = [(npoints, nsample, width), ...]; the input features go through a shared per-point stem first."""

    def __init__(self, in_channels=6, stem=32, stages=((512, 16, 64), (128, 16, 128), (32, 16, 256)), pointops=None,
                 sa_impl="fused"):
        super().__init__()
        self.stem = nn.Sequential(nn.Linear(in_channels, stem, bias=False), nn.BatchNorm1d(stem), nn.ReLU())
        c = stem
        self.stages = nn.ModuleList()
        for npoints, nsample, width in stages:
            self.stages.append(SAStage(c, width, npoints, nsample, pointops=pointops, sa_impl=sa_impl))
            c = width
        self.num_channels = c

    def forward(self, pcd_dict, return_all=False):
        p, o = pcd_dict["coord"], pcd_dict["offset"]
        x = PointNet._layer(self.stem, pcd_dict["feat"])
        levels = [(p, x, o)]
        for stage in self.stages:
            p, x, o = stage(p, x, o)
            levels.append((p, x, o))
        return levels if return_all else (p, x, o)


class FeaturePropagation(nn.Module):
    def __init__(self, coarse_channels, skip_channels, out_channels, pointops=None):
        super().__init__()
        if pointops is None:
            from .. import pointops as _hip_pointops

            pointops = _hip_pointops
        self._pointops = [pointops]
        self.mlp = nn.Sequential(nn.Linear(coarse_channels + skip_channels, out_channels, bias=False),
                                 nn.BatchNorm1d(out_channels), nn.ReLU())

    def forward(self, fine, coarse):
        """fine = (p (n,3), x (n,Cs), o), coarse = (p (m,3), x (m,Cc), o) -> (n, out)."""
        (p1, x1, o1), (p2, x2, o2) = fine, coarse
        up = self._pointops[0].interpolation(p2, p1, x2.float(), o2, o1, k=3)
        return PointNet._layer(self.mlp, torch.cat([x1.float(), up], dim=1))


class PatchTokenizer(nn.Module):
    """(B clouds) -> (B, G, H) patch tokens + (B, G, H) position embedding of the patch centres."""

    def __init__(self, in_channels=6, num_groups=64, group_size=32, hidden_dim=384, pointops=None, sa_impl="fused"):
        super().__init__()
        self.stage = SAStage(in_channels, hidden_dim, num_groups, group_size, pointops=pointops, sa_impl=sa_impl)
        self.hidden_dim = hidden_dim

    def forward(self, pcd_dict):
        p, o = pcd_dict["coord"], pcd_dict["offset"]
        centres, tokens, _ = self.stage(p, pcd_dict["feat"], o)
        b = o.shape[0]
        pos = coord_embedding_sine(centres, self.hidden_dim)
        return tokens.view(b, -1, self.hidden_dim), pos.view(b, -1, self.hidden_dim)


def _neighbour_pre(owner, pointops, p, o, n_p, n_o, fps_idx, nsample, radius=None):
    """The `pre` record set_abstraction() consumes, for given centres and one neighbourhood definition."""
    with torch.no_grad():
        if radius is None:
            knn_idx, _ = pointops.knn_query(nsample, p, o, n_p, n_o)
        else:
            knn_idx, _ = pointops.ball_query(nsample, float(radius), 0.0, p, o, n_p, n_o)  # -1 = fewer than nsample in range
        istats = None
        if getattr(owner, "sa_impl", None) == "fused" and p.is_cuda:
            from .sa_fused import index_stats, layout_of

            istats = index_stats(p, n_p, knn_idx, *layout_of(o, n_o))
    return {"idx": fps_idx, "n_p": n_p, "knn_idx": knn_idx, "istats": istats, "event": None}


class _Branch(nn.Module):
    """(linear, bn) of one neighbourhood scale, under the attribute names set_abstraction() expects."""

    def __init__(self, in_channels, out_channels, nsample, sa_impl):
        super().__init__()
        self.sa_impl, self.pcd_nsample = sa_impl, nsample
        self.linear = nn.Linear(3 + in_channels, out_channels, bias=False)
        self.bn = nn.BatchNorm1d(out_channels)
        self.pool = nn.MaxPool1d(nsample)
        self.relu = nn.ReLU(inplace=True)


class SAStageMSG(nn.Module):
    """scales = [(nsample, radius or None, width), ...] -> tokens of width sum(widths)."""

    def __init__(self, in_channels, npoints, scales=((16, 0.05, 32), (32, 0.1, 64)), pointops=None, sa_impl="fused"):
        super().__init__()
        if pointops is None:
            from .. import pointops as _hip_pointops

            pointops = _hip_pointops
        self._pointops = [pointops]
        self.pcd_npoints = npoints
        self.scales = [(int(k), r) for k, r, _ in scales]
        self.branches = nn.ModuleList([_Branch(in_channels, w, k, sa_impl) for k, _, w in scales])
        self.out_channels = sum(w for _, _, w in scales)

    def forward(self, p, x, o):
        po = self._pointops[0]
        n_o = _query_offsets(self, o, self.pcd_npoints)
        with torch.no_grad():
            fps_idx = po.farthest_point_sampling(p, o, n_o)
            n_p = p[fps_idx.long(), :]
        outs = []
        for br, (k, radius) in zip(self.branches, self.scales):
            pre = _neighbour_pre(br, po, p, o, n_p, n_o, fps_idx, k, radius)
            outs.append(set_abstraction(br, po, p, x, o, n_o, impl=br.sa_impl, pre=pre)[1])
        return n_p, torch.cat([t.float() for t in outs], dim=1), n_o


class InvResMLP(nn.Module):
    """x -> relu( x + MLP( LocalAgg(x) ) ): LocalAgg = the SA layer with EVERY point as a query (no down-sampling),
    MLP = Linear(C, e*C) - BN - ReLU - Linear(e*C, C) - BN  (PointNeXt's inverted bottleneck, expansion e = 4)."""

    def __init__(self, channels, nsample=16, expansion=4, pointops=None, sa_impl="fused"):
        super().__init__()
        if pointops is None:
            from .. import pointops as _hip_pointops

            pointops = _hip_pointops
        self._pointops = [pointops]
        self.agg = _Branch(channels, channels, nsample, sa_impl)
        self.fc1 = nn.Sequential(nn.Linear(channels, expansion * channels, bias=False), nn.BatchNorm1d(expansion * channels), nn.ReLU())
        self.fc2 = nn.Linear(expansion * channels, channels, bias=False)
        self.bn2 = nn.BatchNorm1d(channels)

    def forward(self, p, x, o, pre=None):
        po = self._pointops[0]
        if pre is None:
            ident = torch.arange(p.shape[0], dtype=torch.int32, device=p.device)  # every point is its own query
            pre = _neighbour_pre(self.agg, po, p, o, p, o, ident, self.agg.pcd_nsample)
        y = set_abstraction(self.agg, po, p, x, o, o, impl=self.agg.sa_impl, pre=pre)[1]
        y = PointNet._layer(self.fc1, y if y.dtype == x.dtype else y.to(x.dtype))
        y = self.bn2(linear_rows(y, self.fc2.weight))
        return torch.relu(x.float() + y.float()), pre


class PointNeXtBackbone(nn.Module):
    """Per-point backbone (n, in) -> (n, out): stem Linear/BN/ReLU, `blocks` InvResMLP blocks sharing ONE kNN graph at full
    resolution, head Linear/BN/ReLU.  Drop-in for ACTPCD's `backbone` (exposes num_channels)."""

    def __init__(self, in_channels=6, width=64, blocks=2, nsample=16, out_channels=512, pointops=None, sa_impl="fused"):
        super().__init__()
        self.stem = nn.Sequential(nn.Linear(in_channels, width, bias=False), nn.BatchNorm1d(width), nn.ReLU())
        self.blocks = nn.ModuleList([InvResMLP(width, nsample, pointops=pointops, sa_impl=sa_impl) for _ in range(blocks)])
        self.head = nn.Sequential(nn.Linear(width, out_channels, bias=False), nn.BatchNorm1d(out_channels), nn.ReLU())
        self.num_channels = out_channels

    def forward(self, pcd_dict):
        p, o = pcd_dict["coord"], pcd_dict["offset"]
        x = PointNet._layer(self.stem, pcd_dict["feat"])
        pre = None
        for blk in self.blocks:
            x, pre = blk(p, x, o, pre)  # the neighbour lists depend on the coordinates only: computed once
        return PointNet._layer(self.head, x)

    def fused_batchnorms(self):
        return [self.stem[1], self.head[1]] + [m for blk in self.blocks for m in ([blk.agg.bn, blk.fc1[1]] if blk.agg.sa_impl == "fused" else [blk.fc1[1]])]


class PatchBertObsEncoder(nn.Module):
    """PointBERT-style observation encoder for the Diffusion Policy: patch tokens (PatchTokenizer) + position embedding ->
    post-norm transformer encoder -> [max | mean] pooling -> Linear.  Same call contract as PCDObsEncoder (obs dict with
    `pcds` + low-dim keys -> (B*To, out + low-dim)); statistics-free (LayerNorm only) behind the tokenizer."""

    def __init__(self, shape_meta, num_groups=128, group_size=32, hidden_dim=384, depth=4, nhead=6, out_channels=128, n_obs_step=2,
                 pointops=None, sa_impl="fused"):
        super().__init__()
        from .transformer import TransformerEncoder

        obs_meta = shape_meta["obs"]
        self.pcd_keys = [k for k, a in obs_meta.items() if a.get("type") == "pcd"]
        self.low_dim_keys = [k for k, a in obs_meta.items() if a.get("type", "low_dim") == "low_dim"]
        self.key_shape_map = {k: tuple(a["shape"]) for k, a in obs_meta.items()}
        in_channels = self.key_shape_map[self.pcd_keys[0]][0]
        self.tokenizer = PatchTokenizer(in_channels, num_groups, group_size, hidden_dim, pointops=pointops, sa_impl=sa_impl)
        self.encoder = TransformerEncoder(d_model=hidden_dim, nhead=nhead, dim_feedforward=4 * hidden_dim, dropout=0.0,
                                          num_layers=depth)
        self.proj = RowsLinear(2 * hidden_dim, out_channels)
        self.n_obs_step, self._out_channels = n_obs_step, out_channels
        self.overlap_sampling = False

    def pcd_features(self, pcd_dict):
        tokens, pos = self.tokenizer(pcd_dict)
        h = self.encoder(tokens.float(), pos=pos)
        return self.proj(torch.cat([h.max(dim=1).values, h.mean(dim=1)], dim=-1))

    def forward(self, obs_dict):
        feats, batch = [], None
        for key in self.pcd_keys:
            pcd = obs_dict[key]
            f = pcd["pcd_feat"] if "pcd_feat" in pcd else self.pcd_features(pcd)
            batch = f.shape[0]
            feats.append(f)
        for key in self.low_dim_keys:
            data = obs_dict[key]
            assert batch is None or batch == data.shape[0], (key, batch, data.shape)
            batch = data.shape[0]
            feats.append(data)
        return torch.cat(feats, dim=-1)

    def output_shape(self):
        return (self._out_channels + sum(self.key_shape_map[k][0] for k in self.low_dim_keys),)

    def fused_batchnorms(self):
        return [self.tokenizer.stage.bn] if self.tokenizer.stage.sa_impl == "fused" else []
