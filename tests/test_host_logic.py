

def test_weight_gradient_block_count_prefers_powers_of_two():
    """policy/rows_linear._splits: the row blocks of a split-K weight gradient always tile the rows exactly when a
    suitable divisor exists, stay within [TARGET_CHUNK, ...) rows per block, and are a power of two whenever one divides."""
    from pointcloudmatters_amd.policy.rows_linear import MAX_SPLITS, TARGET_CHUNK, _splits

    for rows in (2048, 2570, 4112, 4120, 8192, 16408, 32768, 65536, 131072, 4115, 3 * 4096):
        s = _splits(rows)
        assert 1 <= s <= MAX_SPLITS
        pow2 = [c for c in (64, 32, 16, 8, 4, 2) if rows % c == 0 and rows // c >= TARGET_CHUNK]
        if pow2:
            assert s == pow2[0], (rows, s)
            assert rows % s == 0
    assert _splits(4120) == 4 and _splits(8192) == 8 and _splits(131072) == 64


def test_gradient_sink_adds_once_and_backpropagates_into_the_source_graph():
    """policy/fused_ops.GradSink on the host: the pushed gradients are summed and sent through the source's own graph
    (here an expand of a parameter), exactly like the engine's pairwise accumulation would."""
    import torch

    from pointcloudmatters_amd.policy.fused_ops import GradSink

    emb = torch.randn(5, 4, requires_grad=True)
    src = emb.unsqueeze(0).expand(3, -1, -1)
    sink = GradSink(src)
    sink.flush()  # nothing pushed: no-op
    assert emb.grad is None
    gs = [torch.randn(3, 5, 4) for _ in range(4)]
    for g in gs:
        sink.add(g)
    sink.flush()
    torch.testing.assert_close(emb.grad, sum(gs).sum(0))
    sink.flush()  # emptied by the first flush
    torch.testing.assert_close(emb.grad, sum(gs).sum(0))
