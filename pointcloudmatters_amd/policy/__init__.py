"""Policies and point-cloud tokenizers of the BC path (reference: src/models/components/)."""
from .act import ACTPCD, ACTRLBenchPCD
from .losses import KLDivergence
from .pointnet import PointNet
from .transformer import Transformer, TransformerEncoder

__all__ = ["ACTPCD", "KLDivergence", "PointNet", "Transformer", "TransformerEncoder"]
