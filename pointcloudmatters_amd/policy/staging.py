"""Backward staging: marks where a policy's backward pass may be cut into stages.

Data-parallel training wants the gradient exchange to overlap the backward pass.  The trainer therefore runs backward
in STAGES -- most recently used parameters first -- and starts the all-reduce of a stage's (contiguous) gradient slab
while the next stage is still computing (bc/trainer.py).  A stage boundary is a set of activations ("cut"): everything
the later part of the forward pass consumes from the earlier part.  The policy marks them with

    memory, pos = staging.cut("transformer.decoder", memory, pos)

which is the identity when no recorder is active.  With a recorder, every tensor is passed through a fresh view node so
that the cut tensors are consumed ONLY downstream of the cut (a tensor used on both sides -- the position embedding feeds
encoder and decoder -- would otherwise be a root and an input of the same partial backward), and the views are recorded
as (tensor, root_at, requested_at):

  root_at       the boundary BELOW which the tensor is produced: it seeds the partial backward of the stage under it;
  requested_at  the boundary directly below the region that CONSUMES it: the partial backward of that region asks for its
                gradient.  Equal to root_at for a tensor handed from one region to the next; different for one that skips
                regions (the CVAE's mu / logvar are produced at the bottom and consumed by the loss at the top).

Partial backward of the stage between boundaries (upper U, lower L):
    roots  = recorded tensors with root_at == U that hold a gradient (stage 0: the loss)
    inputs = recorded tensors with requested_at == L  +  the stage's parameters
Every path from the loss to an earlier stage must pass through a recorded tensor.
"""
import contextlib

import torch

_REC = None


class Recorder(list):
    def roots(self, boundary):
        return [t for t, root_at, _ in self if root_at == boundary and t.grad is not None]

    def requested(self, boundary):
        return [t for t, _, req in self if req == boundary]


@contextlib.contextmanager
def record():
    global _REC
    prev, _REC = _REC, Recorder()
    try:
        yield _REC
    finally:
        _REC = prev


def cut(name, *tensors, consumed_above=None):
    if _REC is None:
        return tensors if len(tensors) != 1 else tensors[0]
    out = tuple(t.view_as(t) if (torch.is_tensor(t) and t.requires_grad) else t for t in tensors)
    _REC.extend((t, name, consumed_above or name) for t in out if torch.is_tensor(t) and t.requires_grad)
    return out if len(out) != 1 else out[0]
