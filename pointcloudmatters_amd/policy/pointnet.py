"""Per-point MLP encoder ("PointNet") -- behavioural counterpart of
/root/reference/src/models/components/pcd_encoder/pointnet.py:16-85.

The reference writes the five layers as spconv ``SubMConv3d(kernel_size=1, bias=False)`` +
``BatchNorm1d(eps=1e-3, momentum=0.01)`` + ReLU.  A 1x1x1 submanifold convolution touches only the
active voxel itself, i.e. it is a per-point ``Linear`` -- the voxel hash / indice-pair build that
spconv performs for it is pure overhead, so here the layers ARE ``nn.Linear`` (hipBLASLt GEMMs).
Module names are kept (``conv1.0`` = linear, ``conv1.1`` = BN) so state-dict keys line up; only the
spconv weight layout differs (see ``load_reference_state_dict``).  spconv itself is a third-party
CUDA library absent from the reference tree: parity of this restatement is unpinned.
"""
import torch.nn as nn

from .rows_linear import linear_rows
from .rows_linear import RowsLinear


def _block(cin, cout):
    return nn.Sequential(nn.Linear(cin, cout, bias=False), nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01), nn.ReLU())


class PointNet(nn.Module):
    def __init__(self, in_channels, num_classes=0, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.conv1 = _block(in_channels, 64)
        self.conv2 = _block(64, 64)
        self.conv3 = _block(64, 64)
        self.conv4 = _block(64, 128)
        self.conv5 = _block(128, 512)
        # pointnet.py:57-61: a biased 512 -> num_classes projection without BN/ReLU, or identity
        self.final = RowsLinear(512, num_classes, bias=True) if num_classes > 0 else nn.Identity()
        self.num_channels = num_classes if num_classes > 0 else 512

    def forward(self, input_dict):
        """input_dict: feat (n, in_channels) [+ grid_coord, offset, unused by a k=1 conv] -> (n, C)."""
        x = input_dict["feat"]
        for block in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):
            x = self._layer(block, x)
        return linear_rows(x, self.final.weight, self.final.bias) if isinstance(self.final, nn.Linear) else x

    @staticmethod
    def _layer(block, x):
        """Linear -> BatchNorm1d -> ReLU; on the GPU the BN + ReLU tail is the fused kernel pair of csrc/bnrelu.hip."""
        y = linear_rows(x, block[0].weight, block[0].bias)
        if y.is_cuda:
            from . import bn_relu as fused

            if fused.supported(y, block[1]):
                return fused.bn_relu(y, block[1])
        if getattr(block[1], "_pcm_sync", False) and block[1].training:
            raise RuntimeError("this BatchNorm is marked for synchronised statistics, which only the fused HIP path implements "
                               "(policy/sync_bn.py); its input does not qualify for that path")
        return block[2](block[1](y))

    def fused_batchnorms(self):
        """BatchNorm layers whose training-mode forward / backward run in the fused kernels (they exchange their statistics
        themselves when synchronised BatchNorm is on, see policy/sync_bn.py)."""
        return [blk[1] for blk in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5)]

    def load_reference_state_dict(self, state_dict, strict=True):
        """Accept a reference checkpoint: spconv stores SubMConv3d weights as (1,1,1,Cin,Cout)
        (spconv 2.x KRSC off) or (Cout,1,1,1,Cin) (KRSC); both reduce to Linear's (Cout, Cin)."""
        own = self.state_dict()
        fixed = {}
        for k, v in state_dict.items():
            if k in own and v.dim() == 5:
                if tuple(v.shape[:3]) == (1, 1, 1):      # (k, k, k, Cin, Cout)
                    v = v.reshape(v.shape[3], v.shape[4]).t().contiguous()
                elif tuple(v.shape[1:4]) == (1, 1, 1):   # (Cout, k, k, k, Cin)
                    v = v.reshape(v.shape[0], v.shape[4])
            fixed[k] = v
        return self.load_state_dict(fixed, strict=strict)
