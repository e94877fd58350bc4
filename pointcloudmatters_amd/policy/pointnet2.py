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
"""
import torch
import torch.nn as nn

from .pointnet import PointNet
from .sa_layer import coord_embedding_sine, set_abstraction


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
        b = int(o.shape[0])
        host = [self.pcd_npoints * (i + 1) for i in range(b)]
        t = torch.tensor(host, dtype=torch.int32, device=o.device)
        t._pcm_host = host
        return t

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
