"""DETR-style post/pre-norm transformer used by ACT, written batch-first around fused attention.

Behavioural counterpart of /root/reference/src/models/components/act/transformer.py:16-425 with the
SAME parameter names (``encoder.layers.N.self_attn.in_proj_weight`` ..., ``decoder.norm.weight``) so
reference checkpoints load unchanged.  Differences in *how* it runs, not in what it computes:

* activations stay (batch, tokens, dim) -- no (tokens, batch, dim) permutes;
* attention goes through ``F.scaled_dot_product_attention`` (flash kernel on ROCm, bf16 under
  autocast) instead of ``nn.MultiheadAttention.forward`` with its default ``need_weights=True``
  that materialises an (S x S) probability matrix per head (S = 2051 on the shipped config);
* q and k share one (2E x E) projection GEMM when they have the same input
  (``q = k = x + pos`` in every self-attention of this model).

``nn.MultiheadAttention`` modules are kept as parameter containers (packed ``in_proj_weight``,
``out_proj``) for state-dict compatibility; their ``forward`` is never called.
"""
import copy
from typing import Optional

import torch
import torch.nn.functional as F

from .rows_linear import linear_rows
from torch import Tensor, nn


def _split_heads(x: Tensor, nhead: int) -> Tensor:
    b, l, e = x.shape
    return x.view(b, l, nhead, e // nhead).transpose(1, 2)  # (B, H, L, hd)


def _residual(x: Tensor, y: Tensor) -> Tensor:
    """x + y with the sub-layer output brought to the residual stream's dtype first: the mixed-dtype
    (fp32 + bf16) elementwise kernel runs at <1 TB/s, a cast followed by a same-dtype add at ~5 TB/s."""
    return x + (y if y.dtype == x.dtype else y.to(x.dtype))


def _add_norm(norm: nn.Module, dropout: nn.Module, x: Tensor, y: Tensor) -> Tensor:
    """norm(x + dropout(y)): one fused HIP kernel each way when a FusedContext is active (training loop on
    the GPU), the framework's dropout / cast / add / layer_norm chain otherwise."""
    from . import fused_ops

    if fused_ops.drln_supported(x, y, norm):
        return fused_ops.drln(x, y, norm, dropout)
    return norm(_residual(x, dropout(y)))


def _attn_add_norm(norm: nn.Module, dropout: nn.Module, x: Tensor, mha: nn.MultiheadAttention, *args, n_out: int = 1,
                   emit_pos: Optional[Tensor] = None, **kwargs):
    """norm(x + dropout(mha(...))): with a FusedContext active the output projection, the residual add and the norm
    are one autograd node (fused_ops.proj_drln); otherwise the plain chain.  n_out = 2: the result twice, for two consumers --
    the fused node hands out two aliases and sums their gradients inside its backward kernel (no add launch)."""
    from . import fused_ops

    if n_out > 1:
        ctx_on = fused_ops.current() is not None and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
        if ctx_on:
            aux = {}
            a = attention(mha, *args, project=False, aux=aux, **kwargs)
            x = aux.get("residual", x)
            ydt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else a.dtype
            if fused_ops.drln_supported(x, None, norm, y_dtype=ydt) and mha.out_proj.bias is not None:
                # emit_pos: the query projection that consumes this output with that pos finds its bf16 input ready
                emit = fused_ops.emit_for(emit_pos, x.shape, False, ("add_pos.s", None)) if emit_pos is not None else None
                return fused_ops.proj_drln(a, mha.out_proj, x, norm, dropout, n_out=n_out, emit=emit)
            out = _add_norm(norm, dropout, x, linear_rows(a, mha.out_proj.weight, mha.out_proj.bias))
        else:
            out = _add_norm(norm, dropout, x, attention(mha, *args, **kwargs))
        return (out,) * n_out
    ctx_on = fused_ops.current() is not None and x.is_cuda and x.dtype == torch.float32
    if ctx_on:
        aux = {}
        a = attention(mha, *args, project=False, aux=aux, **kwargs)
        x = aux.get("residual", x)  # the in-projection node's alias of x (fused_ops._SelfAttnInProj): one consumer of x
        ydt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else a.dtype
        if fused_ops.drln_supported(x, None, norm, y_dtype=ydt) and mha.out_proj.bias is not None:
            return fused_ops.proj_drln(a, mha.out_proj, x, norm, dropout)
        return _add_norm(norm, dropout, x, linear_rows(a, mha.out_proj.weight, mha.out_proj.bias))
    return _add_norm(norm, dropout, x, attention(mha, *args, **kwargs))


def _ffn_norm(layer, norm: nn.Module, dropout_out: nn.Module, x: Tensor, n_out: int = 1, emit_pos: Optional[Tensor] = None):
    """norm(x + dropout_out(linear2(dropout(act(linear1(x)))))): one fused HIP kernel each way for the shipped
    relu / dim_feedforward = 32 layers when a FusedContext is active, framework ops otherwise.  n_out: see _attn_add_norm."""
    from . import fused_ops

    if layer.activation is F.relu and fused_ops.ffn_ln_supported(x, layer.linear1, layer.linear2, norm):
        if n_out > 1 and not torch.is_grad_enabled():
            return (fused_ops.ffn_ln(x, layer.linear1, layer.linear2, norm, layer.dropout, dropout_out),) * n_out
        # emit_pos: the NEXT layer's in-projection (q = k = out + pos, v = out) finds its bf16 inputs ready
        emit = fused_ops.emit_for(emit_pos, x.shape, True, ("in_proj.qkv", None)) if emit_pos is not None else None
        return fused_ops.ffn_ln(x, layer.linear1, layer.linear2, norm, layer.dropout, dropout_out, n_out=n_out, emit=emit)
    out = _add_norm(norm, dropout_out, x, layer._ffn(x))
    return out if n_out == 1 else (out,) * n_out


def attention(
    mha: nn.MultiheadAttention,
    query: Tensor,
    key: Optional[Tensor],
    value: Optional[Tensor],
    key_padding_mask: Optional[Tensor] = None,
    training: bool = False,
    kv: Optional[tuple] = None,
    project: bool = True,
    qk_parts: Optional[tuple] = None,
    q_parts: Optional[tuple] = None,
    aux: Optional[dict] = None,
) -> Tensor:
    """Multi-head attention with the parameters of ``mha``; inputs (B, L, E) / (B, S, E).
    ``key_padding_mask`` (B, S) bool, True = ignore (nn.MultiheadAttention convention).
    ``kv``: (k, v, w_q, b_q) with already-projected k, v (B, S, E) -- the decoder projects the memory for all
    of its layers with two GEMMs and splits every packed in_proj weight exactly once
    (TransformerDecoder._project_memory)."""
    e, h = mha.embed_dim, mha.num_heads
    w, b = mha.in_proj_weight, mha.in_proj_bias
    # torch.split / unbind instead of slicing: their backward is ONE cat / stack kernel, whereas every
    # slice's backward allocates a zero tensor of the full size and copies into it
    projected = False
    if qk_parts is not None:  # self-attention with q = k = x + pos, v = x
        from . import fused_ops

        x_in, pos_in = qk_parts
        if fused_ops.self_attn_in_proj_supported(x_in, pos_in, mha):
            q, k, v, x_res = fused_ops.self_attn_in_proj(x_in, pos_in, mha)  # one autograd node (csrc/tokens.hip)
            if aux is not None:
                aux["residual"] = x_res  # the caller's residual branch reads x through this alias
            query, projected = x_in, True
        else:
            query = key = _add_pos(x_in, pos_in)
            value = x_in
    if projected:
        pass
    elif kv is not None:
        k, v, w_q, b_q = kv  # projected memory + this layer's query projection (split once by the decoder)
        if q_parts is not None:
            from . import fused_ops

            if fused_ops.add_pos_linear_supported(q_parts[0], q_parts[1], w_q, b_q):
                q = fused_ops.add_pos_linear(q_parts[0], q_parts[1], w_q, b_q)  # add + cast + GEMM, one autograd node
            else:
                q = linear_rows(_add_pos(*q_parts), w_q, b_q)
        else:
            q = linear_rows(query, w_q, b_q)
    elif query is key:
        w_qk, w_v = torch.split(w, [2 * e, e], dim=0)
        b_qk, b_v = torch.split(b, [2 * e, e], dim=0)
        q, k = linear_rows(query, w_qk, b_qk).unflatten(-1, (2, e)).unbind(-2)
        v = linear_rows(value, w_v, b_v)
    else:
        w_q, w_k, w_v = torch.split(w, [e, e, e], dim=0)
        b_q, b_k, b_v = torch.split(b, [e, e, e], dim=0)
        q = linear_rows(query, w_q, b_q)
        k = linear_rows(key, w_k, b_k)
        v = linear_rows(value, w_v, b_v)
    from . import small_attn

    p_attn = mha.dropout if training else 0.0
    if small_attn.supported(q, k, v, h, p_attn):  # <= 128 queries: MFMA kernel of csrc/attn_small.hip, output already (B, L, E)
        out = small_attn.small_attention(q, k, v, key_padding_mask, h, p_attn)
        if not project:
            return out
        return linear_rows(out, mha.out_proj.weight, mha.out_proj.bias)
    mask = None
    if key_padding_mask is not None:
        mask = (~key_padding_mask)[:, None, None, :]  # True = attend
    out = F.scaled_dot_product_attention(
        _split_heads(q, h), _split_heads(k, h), _split_heads(v, h), attn_mask=mask,
        dropout_p=mha.dropout if training else 0.0,
    )
    out = out.transpose(1, 2).reshape(query.shape[0], query.shape[1], e)
    if not project:
        return out  # the caller fuses out_proj with the residual add + norm (_attn_add_norm)
    return linear_rows(out, mha.out_proj.weight, mha.out_proj.bias)


class GradArena:
    """One (B, S, n, E) buffer for the gradients of the n tensors a `shared_unbind` hands out.  A consumer whose backward
    can write through strides (policy/small_attn.py) asks for `slot(l)` and writes its gradient there; when every
    gradient that comes back to the unbind IS its slot, the stacked gradient is the buffer itself and the n-way copy
    (235 MB of traffic per projection at the 2051-token shape) never happens."""

    hits = 0  # backward passes that returned the shared buffer without a copy (tests read it)

    def __init__(self, n, shape, dtype, device):
        self.n, self.shape, self.dtype, self.device = n, tuple(shape), dtype, device
        self.buf = None

    def slot(self, l):
        if self.buf is None:
            b, s, e = self.shape
            self.buf = torch.empty(b, s, self.n, e, dtype=self.dtype, device=self.device)
        return self.buf[:, :, l]

    def is_slot(self, g, l):
        if self.buf is None or g is None or g.dtype != self.buf.dtype:
            return False
        ref = self.buf[:, :, l]
        return g.data_ptr() == ref.data_ptr() and g.shape == ref.shape and g.stride() == ref.stride()


class _SharedUnbind(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, n, arena):
        ctx.arena, ctx.n = arena, n
        ctx.set_materialize_grads(False)
        return y.unflatten(-1, (n, y.shape[-1] // n)).unbind(-2)

    @staticmethod
    def backward(ctx, *grads):
        arena, n = ctx.arena, ctx.n
        hits = [arena.is_slot(g, l) for l, g in enumerate(grads)]
        if any(hits) and all(h or g is None for h, g in zip(hits, grads)):
            for l, g in enumerate(grads):
                if g is None:  # a layer autograd never visited ("prune_backward")
                    arena.slot(l).zero_()
            buf, arena.buf = arena.buf, None
            GradArena.hits += 1
            return buf.flatten(-2), None, None
        arena.buf = None
        ref = next(g for g in grads if g is not None)
        return torch.stack([g if g is not None else torch.zeros_like(ref) for g in grads], dim=-2).flatten(-2), None, None


class _SplitPacked(torch.autograd.Function):
    """w (3E, ...) -> its three row blocks, like torch.split(w, [E, E, E]): views forward; backward assembles the packed
    gradient.  Inside a deferral window (policy/deferred.py) the three block copies of every layer's weight and bias join ONE
    multi-tensor copy at the end of the backward stage instead of a cat launch per tensor (14 in ACT's decoder)."""

    @staticmethod
    def forward(ctx, w):
        from . import deferred

        e = w.shape[0] // 3
        ctx.defer = deferred.targets(w)
        ctx.meta = (w.shape, w.dtype, w.device, e)
        return w[:e], w[e: 2 * e], w[2 * e:]

    @staticmethod
    def backward(ctx, *gs):
        from . import deferred

        shape, dtype, dev, e = ctx.meta
        if all(g is None for g in gs):
            return None
        gs = [g if g is not None else torch.zeros((e,) + tuple(shape[1:]), dtype=dtype, device=dev) for g in gs]
        if all(g.dtype == dtype for g in gs) and deferred.clear(*ctx.defer):
            out = torch.empty(shape, dtype=dtype, device=dev)
            if deferred.push_copies([(out[i * e: (i + 1) * e], g) for i, g in enumerate(gs)]):
                return deferred.handout(out)
        return torch.cat([g.to(dtype) for g in gs], dim=0)


class _PackRows(torch.autograd.Function):
    """cat(group, dim=0) for several groups of equally shaped tensors at once: ONE multi-tensor copy forward (the decoder's 7
    key weights, 7 value weights and their biases: four cat launches otherwise); backward hands out views."""

    @staticmethod
    def forward(ctx, sizes, *ts):
        outs, pairs, k = [], [], 0
        for n in sizes:
            grp = ts[k: k + n]
            k += n
            rows = grp[0].shape[0]
            out = torch.empty((rows * n,) + tuple(grp[0].shape[1:]), dtype=grp[0].dtype, device=grp[0].device)
            pairs += [(out[i * rows: (i + 1) * rows], t) for i, t in enumerate(grp)]
            outs.append(out)
        if pairs and pairs[0][0].is_cuda:
            from .. import _lib

            _lib.copy_pairs(pairs)  # one table-driven launch (csrc/optim.hip pcm_xfer_batch_hip)
        else:
            by = {}
            for d, s_ in pairs:
                by.setdefault(d.dtype, []).append((d, s_))
            for grp in by.values():
                torch._foreach_copy_([d for d, _ in grp], [s_ for _, s_ in grp])
        ctx.sizes = sizes
        ctx.rows = [t.shape[0] for t in ts]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        res, k = [None], 0
        for n, g in zip(ctx.sizes, gs):
            for i in range(n):
                r = ctx.rows[k]
                res.append(None if g is None else g[i * r: (i + 1) * r])
                k += 1
        return tuple(res)


_ZEROS = {}


def _zeros_like_cached(ref):
    """zeros_like(ref) without a fill launch per step: the decoder's all-zero input (transformer.py:99) is read-only, so one
    buffer per (shape, dtype, device) serves every step.  Never cached while a graph is being captured (the buffer would live
    in that graph's pool); the trainer's eager warm-up runs create it first."""
    if not ref.is_cuda:
        return torch.zeros_like(ref)
    key = (tuple(ref.shape), ref.dtype, ref.device)
    z = _ZEROS.get(key)
    if z is None:
        z = torch.zeros(ref.shape, dtype=ref.dtype, device=ref.device)
        if torch.cuda.is_current_stream_capturing():
            return z
        # entries are NEVER evicted: a captured hipGraph reads its decoder input by raw address and does not keep the tensor
        # alive, so freeing one (round-3 code cleared the cache past 16 shapes) could hand its memory to somebody else under a
        # replaying training graph.  Past 64 shapes (rollouts with many batch sizes) new shapes are simply not cached.
        if len(_ZEROS) < 64:
            _ZEROS[key] = z
    return z


def split_packed(w):
    """The three row blocks of a packed (3E, ...) parameter; their gradients may stay pending inside a deferral window (the
    node copies them into the packed gradient at the end of the stage, after everything that produces them)."""
    from .rows_linear import goes_to_optimizer

    parts = _SplitPacked.apply(w)
    if goes_to_optimizer(w):
        for p in parts:
            p._pcm_defer_ok = True
    return parts


def pack_rows(*groups):
    """[cat(g, dim=0) for g in groups] with one copy launch per dtype (every group: tensors of one shape and dtype)."""
    from .rows_linear import goes_to_optimizer

    if not groups[0][0].is_cuda or any(t.shape != g[0].shape or t.dtype != g[0].dtype for g in groups for t in g):
        return [torch.cat(list(g), dim=0) for g in groups]
    outs = list(_PackRows.apply(tuple(len(g) for g in groups), *[t for g in groups for t in g]))
    for o, g in zip(outs, groups):
        if all(goes_to_optimizer(t) for t in g):
            o._pcm_defer_ok = True  # backward hands out views of the gradient: nothing reads it here
    return outs


def shared_unbind(y, n):
    """``y.unflatten(-1, (n, E)).unbind(-2)`` for (B, S, n*E) activations whose n consumers may write their gradients straight
    into one shared buffer (GradArena): each returned tensor carries ``_pcm_grad_slot = (arena, l)``."""
    if not (y.is_cuda and y.dim() == 3 and y.requires_grad and torch.is_grad_enabled()):
        return y.unflatten(-1, (n, y.shape[-1] // n)).unbind(-2)
    arena = GradArena(n, (y.shape[0], y.shape[1], y.shape[2] // n), y.dtype, y.device)
    outs = _SharedUnbind.apply(y, n, arena)
    for l, o in enumerate(outs):
        o._pcm_grad_slot = (arena, l)
    return outs


def _activation(name):
    if name == "relu":
        return F.relu
    if name == "gelu":
        return F.gelu
    if name == "glu":
        return F.glu
    raise RuntimeError(f"activation should be relu/gelu, not {name}.")


def _add_pos(x, pos):
    if pos is not None and getattr(pos, "_pcm_sink", None) is not None:
        raise RuntimeError("a position embedding with deferred gradients (fused_ops.defer_grads) reached a consumer that "
                           "cannot push into its sink: its gradient would be lost")
    return x if pos is None else x + pos


class TransformerEncoderLayer(nn.Module):
    """transformer.py:209-287."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation = _activation(activation)
        self.normalize_before = normalize_before

    def _ffn(self, x):
        return self.linear2(self.dropout(self.activation(self.linear1(x))))

    def forward(self, src, src_key_padding_mask=None, pos=None, emit_next=False):
        if self.normalize_before:
            y = self.norm1(src)
            qk = _add_pos(y, pos)
            src = _residual(src, self.dropout1(attention(self.self_attn, qk, qk, y, src_key_padding_mask, self.training)))
            return _residual(src, self.dropout2(self._ffn(self.norm2(src))))
        src = _attn_add_norm(self.norm1, self.dropout1, src, self.self_attn, None, None, None, src_key_padding_mask, self.training,
                             qk_parts=(src, pos))
        return _ffn_norm(self, self.norm2, self.dropout2, src, emit_pos=pos if emit_next else None)


class TransformerDecoderLayer(nn.Module):
    """transformer.py:290-410."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.activation = _activation(activation)
        self.normalize_before = normalize_before

    def _ffn(self, x):
        return self.linear2(self.dropout(self.activation(self.linear1(x))))

    def forward(self, tgt, memory, memory_pos, memory_key_padding_mask=None, query_pos=None, kv=None, n_out=1, emit_next=False):
        """memory_pos = memory + pos (the cross-attention key input), shared by all layers; ``kv`` = this
        layer's already-projected memory keys / values (or None: project here)."""
        ca = self.multihead_attn
        if self.normalize_before:
            y = self.norm1(tgt)
            qk = _add_pos(y, query_pos)
            tgt = _residual(tgt, self.dropout1(attention(self.self_attn, qk, qk, y, None, self.training)))
            y = self.norm2(tgt)
            tgt = _residual(tgt, self.dropout2(
                attention(ca, _add_pos(y, query_pos), memory_pos, memory, memory_key_padding_mask, self.training, kv=kv)))
            return _residual(tgt, self.dropout3(self._ffn(self.norm3(tgt))))
        # norm1's output has two consumers (the cross-attention query and the residual): two aliases, see _attn_add_norm
        tgt, tgt_q = _attn_add_norm(self.norm1, self.dropout1, tgt, self.self_attn, None, None, None, None, self.training,
                                    qk_parts=(tgt, query_pos), n_out=2,
                                    emit_pos=query_pos if (kv is not None and query_pos is not None) else None)
        if kv is not None and query_pos is not None:  # query = tgt + query_pos is formed inside the projection's node
            tgt = _attn_add_norm(self.norm2, self.dropout2, tgt, ca, tgt_q, memory_pos, memory, memory_key_padding_mask,
                                 self.training, kv=kv, q_parts=(tgt_q, query_pos))
        else:
            tgt = _attn_add_norm(self.norm2, self.dropout2, tgt, ca, _add_pos(tgt_q, query_pos), memory_pos, memory,
                                 memory_key_padding_mask, self.training, kv=kv)
        return _ffn_norm(self, self.norm3, self.dropout3, tgt, n_out=n_out, emit_pos=query_pos if emit_next else None)


def _clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class TransformerEncoder(nn.Module):
    """transformer.py:118-159.  Batch-first: src (B, S, E), pos broadcastable to it."""

    def __init__(self, d_model=256, nhead=8, dim_feedforward=2048, dropout=0.1, activation="relu",
                 normalize_before=False, num_layers=4):
        super().__init__()
        layer = TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        self.layers = _clones(layer, num_layers)
        self.num_layers = num_layers
        self.norm = nn.LayerNorm(d_model) if normalize_before else None

    def forward(self, src, src_key_padding_mask=None, pos=None):
        out = src
        for li, layer in enumerate(self.layers):
            out = layer(out, src_key_padding_mask=src_key_padding_mask, pos=pos, emit_next=li + 1 < len(self.layers))
        return out if self.norm is None else self.norm(out)


class TransformerDecoder(nn.Module):
    """transformer.py:162-206."""

    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = _clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate
        self.batch_memory_kv = True

    def _project_memory(self, memory, pos):
        """Keys / values of the memory for ALL layers with two GEMMs (S x E)(E x L*E) instead of 2L small
        ones -- and, in backward, two input-gradient GEMMs instead of 2L plus 2L-2 accumulations of the
        (B, S, E) memory gradient.  Same per-layer weights, same sums up to fp32 re-association."""
        e = memory.shape[-1]
        wq, wk, wv, bq, bk, bv = [], [], [], [], [], []
        for layer in self.layers:
            mha = layer.multihead_attn
            if mha.in_proj_weight.is_cuda and torch.is_grad_enabled():
                w_q, w_k, w_v = split_packed(mha.in_proj_weight)  # ONE split per weight; the packed gradients of all
                b_q, b_k, b_v = split_packed(mha.in_proj_bias)    # layers are assembled by one copy launch
            else:
                w_q, w_k, w_v = torch.split(mha.in_proj_weight, [e, e, e], dim=0)  # backward: a single cat
                b_q, b_k, b_v = torch.split(mha.in_proj_bias, [e, e, e], dim=0)
            wq.append(w_q), wk.append(w_k), wv.append(w_v), bq.append(b_q), bk.append(b_k), bv.append(b_v)
        n = len(self.layers)
        wk_all, bk_all, wv_all, bv_all = pack_rows(wk, bk, wv, bv)
        from . import fused_ops

        if pos is not None and fused_ops.add_pos_linear_supported(memory, pos, wk_all, bk_all):
            # keys = W_k (memory + pos): add + cast in one launch inside the projection's node, input gradient in fp32 straight
            # out of the GEMM (the framework chain: an add, a cast, and a cast back in backward)
            keys = fused_ops.add_pos_linear(memory, pos, wk_all, bk_all)
        else:
            keys = linear_rows(_add_pos(memory, pos), wk_all, bk_all)
        k_all = shared_unbind(keys, n)
        v_all = shared_unbind(linear_rows(memory, wv_all, bv_all), n)
        return list(zip(k_all, v_all, wq, bq))

    # What a caller that reads only output [0] (ACT, act.py:270) may ask for:
    #   "keep"            all layers forward, outputs stacked: autograd then pushes exact ZEROS through layers 1.. like
    #                     the reference does (stack -> select backward);
    #   "prune_backward"  all layers forward, but only output 0 is handed on, so the engine never visits layers 1..;
    #                     their parameters keep receiving the same exact-zero gradients (zero-filled by the optimizer);
    #   "skip"            layers 1.. are not evaluated at all (their outputs are unused).
    first_only = "keep"

    def forward(self, tgt, memory, memory_key_padding_mask=None, pos=None, query_pos=None):
        out = tgt
        layers = self.layers if not (self.return_intermediate and self.first_only == "skip") else self.layers[:1]
        if self.batch_memory_kv and len(layers) == len(self.layers):
            kvs = self._project_memory(memory, pos)
            memory_pos = None  # the layers get their keys / values projected: nobody reads memory + pos
        else:
            memory_pos = _add_pos(memory, pos)
            kvs = [None] * len(layers)
        inter = []
        for li, (layer, kv) in enumerate(zip(layers, kvs)):
            if self.return_intermediate and li + 1 < len(layers):
                # the layer's output feeds the next layer AND the stack of intermediate outputs: two aliases
                out, out_i = layer(out, memory, memory_pos, memory_key_padding_mask=memory_key_padding_mask, query_pos=query_pos,
                                   kv=kv, n_out=2, emit_next=True)
                inter.append(out_i)
                continue
            out = layer(out, memory, memory_pos, memory_key_padding_mask=memory_key_padding_mask, query_pos=query_pos, kv=kv,
                        emit_next=li + 1 < len(layers))
            if self.return_intermediate:
                inter.append(out)
        if self.return_intermediate:
            if self.first_only != "keep":
                return self.norm(inter[0]).unsqueeze(0)
            # stack(norm(out_l)) == norm(stack(out_l)): LayerNorm is row-wise with shared parameters, so the L norms (and
            # their 3 L backward kernels) are ONE launch each way on the stacked tensor.  (The reference pops the last
            # entry and re-appends norm(out): same tensor.)
            return self.norm(torch.stack(inter))
        if self.norm is not None:
            out = self.norm(out)
        return out.unsqueeze(0)


class Transformer(nn.Module):
    """transformer.py:16-115.  forward() takes the same arguments in the same order; `src` and
    `pos_embed` are (B, C, H, W) like the reference, the result is (num_dec_layers, B, queries, C)."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048,
                 dropout=0.1, activation="relu", normalize_before=False, return_intermediate_dec=False):
        super().__init__()
        self.encoder = TransformerEncoder(d_model=d_model, nhead=nhead, dim_feedforward=dim_feedforward, dropout=dropout,
                                          activation=activation, normalize_before=normalize_before,
                                          num_layers=num_encoder_layers)
        dec_layer = TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        self.decoder = TransformerDecoder(dec_layer, num_decoder_layers, nn.LayerNorm(d_model),
                                          return_intermediate=return_intermediate_dec)
        for p in self.parameters():  # transformer.py:57-60
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.d_model = d_model
        self.nhead = nhead

    def forward(self, src, mask, query_embed, pos_embed, latent_input=None, proprio_input=None, additional_pos_embed=None):
        bs = src.shape[0]
        tokens = src.flatten(2).transpose(1, 2)  # (B, HW, C)
        pos = pos_embed.flatten(2).transpose(1, 2)  # (B or 1, HW, C)
        if pos.shape[0] == 1:
            pos = pos.expand(bs, -1, -1)
        add_pos = additional_pos_embed.unsqueeze(0).expand(bs, -1, -1)  # (B, 2+g, C)
        pos = torch.cat([add_pos, pos], dim=1)
        # latent_input / proprio_input arrive (1, B, C) / (k, B, C) [3-d] or (B, C) [2-d] like the reference
        if latent_input.dim() == 2:
            extra = torch.stack([latent_input, proprio_input], dim=1)
        else:
            extra = torch.cat([latent_input, proprio_input], dim=0).transpose(0, 1)
        tokens = torch.cat([extra, tokens], dim=1)
        from . import staging  # backward-stage boundaries for the overlapped gradient exchange (identity otherwise)

        tokens, pos = staging.cut("transformer.encoder", tokens, pos)
        memory = self.encoder(tokens, src_key_padding_mask=mask, pos=pos)
        memory, pos_dec = staging.cut("transformer.decoder", memory, pos)
        query_pos = query_embed.unsqueeze(0).expand(bs, -1, -1)
        dec = self.decoder
        batched_kv = dec.batch_memory_kv and not (dec.return_intermediate and dec.first_only == "skip")
        if self.training and batched_kv and not dec.layers[0].normalize_before:  # both sites of every layer can push
            from . import fused_ops

            query_pos = fused_ops.defer_grads(query_pos)  # 14 gradient sites -> one sum (no-op unless a training loop opted in)
        tgt = _zeros_like_cached(query_pos)
        return self.decoder(tgt, memory, memory_key_padding_mask=mask, pos=pos_dec, query_pos=query_pos)
