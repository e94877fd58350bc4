

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


def _fat_bench_record():
    """A bench record as bulky as round 2's (which overflowed the driver's stdout tail): long notes everywhere."""
    note = "x" * 400
    return {
        "metric": "BC train samples/sec (obs->action), PointNet + SA tokenizer + ACT", "value": 1118.1, "unit": "samples/s", "n_gpus": 1,
        "steps": 30, "warmup": 8, "ms_per_step": 7.155, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "C2: " + note, "global_batch": 8, "parallelism": "dp1", "gradient_exchange": note, "step_mode": "graph"},
        "final_loss": 1.0,
        "step": {"launches_per_step": 700.0, "device_ms_per_step": 8.0, "share_by_family": {("family%d" % i): 0.1 for i in range(8)}},
        "roofline": {"kernel": "pcm_fps_reg_kernel", "bound": "hbm", "achieved": 0.44, "peak": 8000.0, "unit": "GB/s", "frac": 5.5e-5,
                     "traffic": 169035, "limited_by": "latency", "note": note, "selected_by": note, "clocks_per_pick_at_2400MHz": 1232.0},
        "roofline_hbm_resident": {"kernel": "pcm_adamw_flat_kernel", "bound": "hbm", "achieved": 6400.0, "peak": 8000.0, "frac": 0.8,
                                  "traffic": 723102569, "note": note},
        "extra": {("tag%d" % i): {"workload": "REF", "value": 1.0, "ms_per_step": 2.0, "error": note} for i in range(6)},
        "cpu_baseline": {"value": 4.54, "unit": "samples/s", "cores": 16, "kind": "port", "sample": note * 3},
    }


def test_bench_line_stays_under_4_kib_and_keeps_the_contract_fields():
    """bench.py prints ONE JSON line; the driver only keeps a tail of stdout (round 2's 30 KB line lost `value`).  The line is
    clipped to < 4096 bytes without ever dropping the contract fields, `roofline` or `cpu_baseline`."""
    import json

    import bench

    line = bench.compact_line(_fat_bench_record())
    assert len(line) < bench.MAX_LINE_BYTES <= 4096 and "\n" not in line
    rec = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in rec, key
    assert rec["value"] == 1118.1 and rec["config"]["workload"].startswith("C2")
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rec["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in rec["cpu_baseline"], key
    small = {"metric": "m", "value": 1.0, "roofline": {"frac": 0.5}, "cpu_baseline": {"value": 1.0}}
    assert json.loads(bench.compact_line(small)) == small  # nothing is touched when the line already fits


def test_bench_emit_writes_the_tables_to_a_side_file_and_prints_one_line(tmp_path, capsys):
    import json

    import bench

    tables = {"kernels": {("pcm_kernel_%d" % i): {"ms": 0.1, "note": "y" * 300} for i in range(60)}}
    path = str(tmp_path / "sub" / "bench_tables.json")
    bench.emit(_fat_bench_record(), tables, path)
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096
    rec = json.loads(lines[0])
    assert "kernels" not in rec and "roofline" in rec and "cpu_baseline" in rec
    with open(path) as f:
        side = json.load(f)
    assert len(side["kernels"]) == 60 and side["headline"]["value"] == 1118.1


def test_pmc_rows_are_keyed_by_the_full_template_name():
    """Round 2 attributed the <1,0> (grouping backward) variant's HBM bytes to the interpolation forward, a different
    instantiation of pcm_segment_sum_kernel.  A row is matched by its full name, or by its base name only when unique."""
    import bench

    table = {"shapes": {"REF": {
        "pcm_segment_sum_kernel<1, 0>": {"hbm_bytes_per_launch": 700}, "pcm_segment_sum_kernel<4, 1>": {"hbm_bytes_per_launch": 200},
        "pcm_sa_fwd_kernel<__hip_bfloat16, 8>": {"hbm_bytes_per_launch": 139}, "pcm_sa_index_kernel": {"hbm_bytes_per_launch": 36},
        "pcm_sa_index_lds_kernel": {"hbm_bytes_per_launch": 18}}}}
    f = lambda k: bench.pmc_traffic(k, "REF", table)  # noqa: E731
    assert f("pcm_segment_sum_kernel<4, 1>") == 200 and f("pcm_segment_sum_kernel<1,0>") == 700
    assert f("pcm_segment_sum_kernel") is None          # ambiguous: two instantiations
    assert f("pcm_sa_fwd_kernel<bf16>") == 139           # unique base name
    assert f("pcm_sa_index_kernel") == 36 and f("pcm_sa_index_lds_kernel") == 18  # a prefix of another name is not a match
    assert f("pcm_interpolation forward (pcm_segment_sum_kernel)") is None


def test_gradient_slabs_are_cut_from_the_decayed_group_only():
    """bc/trainer.plan_slabs: with a no-decay group at the tail of the flat buffer a stage that owns only biases / norm
    weights must NOT pull an earlier slab's end into the tail (round-2 advisor finding: starts = [0, 1000, 100] gave stage 0
    the slab [0, 1000), covering stage 2's gradients before they exist)."""
    from pointcloudmatters_amd.bc.trainer import plan_slabs

    # leading (decayed) group [0, 900): stage 0 -> [0, 400), stage 1 has no decayed parameter, stage 2 -> [400, 900);
    # tail [900, 1000): biases of stage 0 (900..939), stage 1 (940..969), stage 2 (970..999)
    stages = [[(0, 400), (900, 40)], [(940, 30)], [(400, 500), (970, 30)]]
    assert plan_slabs(stages, 900, 1000) == [(0, 400), (400, 400), (400, 1000)]
    # one group, three contiguous stages
    assert plan_slabs([[(0, 10), (10, 5)], [(15, 20)], [(35, 65)]], 100, 100) == [(0, 15), (15, 35), (35, 100)]
    # only the last stage holds decayed weights
    assert plan_slabs([[(50, 10)], [(0, 50), (60, 4)]], 50, 64) == [(0, 0), (0, 64)]
    # a layout that is NOT in backward order loses the overlap, never the correctness: everything goes with the last stage
    assert plan_slabs([[(500, 100)], [(0, 500)]], 600, 600) == [(0, 0), (0, 600)]


def test_attention_dropout_hash_statistics():
    """The attention-dropout mask generator (csrc/pcm_attn.hpp: Weyl sequence over key pairs + one xorshift-multiply-xorshift
    round, 16-bit thresholds) restated in numpy: keep rate, row / column means and the correlations between adjacent keys,
    adjacent pairs and adjacent rows stay at the noise level of independent bits."""
    import numpy as np

    M32 = 0xFFFFFFFF

    def mix32(h):
        h = h ^ (h >> 16)
        h = (h * 0x7FEB352D) & M32
        h = h ^ (h >> 15)
        h = (h * 0x846CA68B) & M32
        return h ^ (h >> 16)

    def mixp(x):
        x = x ^ (x >> 16)
        x = ((x & 0xFFFFFF) * 0xD35A2D) & M32
        return x ^ (x >> 12)

    rows, S = 2048, 2048
    for seed, p in ((1, 0.1), (12345678901234, 0.1), (2 ** 40 + 77, 0.5)):
        rowid = np.arange(rows, dtype=np.int64)
        k = (seed & M32) ^ (((seed >> 32) * 0x9E3779B9) & M32) ^ ((3 * 0x85EBCA6B) & M32)
        rb = mix32((k ^ rowid) & M32)
        key = np.arange(S, dtype=np.int64)
        bits = mixp((rb[:, None] + (key >> 1)[None, :] * 0x9E3779B1) & M32)
        half = np.where((key & 1)[None, :] == 1, bits >> 16, bits & 0xFFFF)
        m = (half >= int(p * 65536 + 0.5)).astype(np.float64)
        x = m - m.mean()
        v = x.var()
        assert abs(m.mean() - (1 - p)) < 1.5e-3
        for lag_keys, lag_rows in ((1, 0), (2, 0), (0, 1), (0, 2), (1, 1)):
            a = x[lag_rows:, lag_keys:]
            b = x[: rows - lag_rows, : S - lag_keys]
            assert abs((a * b).mean() / v) < 3e-3, (seed, p, lag_keys, lag_rows)
        sd = (p * (1 - p) / S) ** 0.5
        assert np.abs(m.mean(1) - (1 - p)).max() < 5.5 * sd and np.abs(m.mean(0) - (1 - p)).max() < 5.5 * (p * (1 - p) / rows) ** 0.5


def test_pointnet_accepts_both_spconv_checkpoint_layouts():
    """policy/pointnet.PointNet.load_reference_state_dict: the reference's PointNet stores its k = 1 SubMConv3d weights in
    spconv's layouts (/root/reference/src/models/components/pcd_encoder/pointnet.py:31-55): (1,1,1,Cin,Cout) (spconv 2.x
    default) or (Cout,1,1,1,Cin) (KRSC builds).  A 1x1x1 submanifold convolution is out[n, co] = sum_ci feat[n, ci] *
    W[ci -> co]; the loaded module must compute exactly that from either layout, BatchNorm buffers and the biased `final`
    head included."""
    import torch

    from pointcloudmatters_amd.policy.pointnet import PointNet

    torch.manual_seed(0)
    src = PointNet(6, num_classes=96).eval()
    with torch.no_grad():
        for blk in (src.conv1, src.conv2, src.conv3, src.conv4, src.conv5):
            blk[1].running_mean.normal_(0, 0.1), blk[1].running_var.uniform_(0.5, 1.5)
            blk[1].weight.uniform_(0.5, 1.5), blk[1].bias.normal_(0, 0.1)
    feat = torch.randn(50, 6)
    want = src({"feat": feat})

    def conv_by_definition(x, w5, layout):  # the convolution as spconv defines it, straight from the checkpoint tensor
        return torch.einsum("ni,io->no", x, w5[0, 0, 0]) if layout == "rsck" else torch.einsum("ni,oi->no", x, w5[:, 0, 0, 0])

    for layout in ("rsck", "krsc"):
        ckpt = {}
        for k, v in src.state_dict().items():
            if k.endswith(".0.weight") and k.startswith("conv"):
                v = v.t()[None, None, None].contiguous() if layout == "rsck" else v[:, None, None, None, :].contiguous()
                assert v.dim() == 5
            ckpt[k] = v.clone()
        dst = PointNet(6, num_classes=96).eval()
        res = dst.load_reference_state_dict(ckpt, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        torch.testing.assert_close(dst({"feat": feat}), want, rtol=1e-6, atol=1e-6)
        # first layer against the definition of the convolution on the raw checkpoint tensor
        y = conv_by_definition(feat, ckpt["conv1.0.weight"], layout)
        torch.testing.assert_close(dst.conv1[0](feat), y, rtol=1e-6, atol=1e-6)
        for k, v in src.state_dict().items():
            assert torch.equal(dst.state_dict()[k], v), k


def test_deferral_window_host_logic():
    """policy/deferred.py without a GPU: the window's state machine, the batch-size padding rule, and that nothing is ever
    queued for host tensors (the producer then reduces / multiplies immediately, as outside a window)."""
    import torch

    from pointcloudmatters_amd.policy import deferred

    assert not deferred.active()
    assert deferred.begin() and deferred.active()
    try:
        assert not deferred.begin()  # a nested begin does not own the window (its caller must not end it)
        part, out = torch.zeros(4, 8), torch.zeros(8)
        assert not deferred.push(part, 4, 8, out_f32=out)  # host tensors: never deferred
        assert deferred.push_wgrad(torch.zeros(6, 3), torch.zeros(6, 5), torch.float32) is None
        t = deferred.take((2, 3), torch.float32, "cpu", "site")
        assert t.shape == (2, 3) and not t.is_cuda
        assert deferred.flush() == 0
    finally:
        deferred.end()
    assert not deferred.active() and not deferred._W and not deferred._P and not deferred._C
    assert [deferred._padded(n) for n in (1, 2, 3, 4, 5, 7, 8, 13, 14, 15, 16, 17, 21)] == [1, 2, 4, 4, 5, 8, 8, 13, 16, 16, 16, 17, 21]
    # what autograd receives for a pending result: another tensor object on the same storage
    base = torch.zeros(3, 4)
    h = deferred.handout(base)
    assert h is not base and h.data_ptr() == base.data_ptr() and h.shape == base.shape
    # targets / clear: a leaf that already holds a gradient forces the immediate path
    w = torch.nn.Parameter(torch.zeros(2, 2))
    ok, leaves = deferred.targets(w, None)
    assert ok and leaves == [w]
    assert not deferred.clear(ok, leaves)  # no window
    assert deferred.begin()
    try:
        assert deferred.clear(ok, leaves)
        w.grad = torch.ones(2, 2)
        assert not deferred.clear(ok, leaves)
        assert not deferred.clear(*deferred.targets(w * 2))  # a gradient that passes through another node is never left pending
    finally:
        deferred.end()


def test_rows_linear_module_is_nn_linear_on_the_host():
    """policy/rows_linear.RowsLinear: an nn.Linear (isinstance, state-dict keys, initialisation stream) whose host forward is F.linear."""
    import torch
    import torch.nn as nn
    from pointcloudmatters_amd.policy.rows_linear import RowsLinear

    torch.manual_seed(3)
    a = RowsLinear(12, 5)
    torch.manual_seed(3)
    b = nn.Linear(12, 5)
    assert isinstance(a, nn.Linear) and list(a.state_dict()) == list(b.state_dict())
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))
    x = torch.randn(4, 12)
    assert torch.equal(a(x), b(x))


def test_every_test_file_has_a_run_order_tier():
    """tests/conftest.py orders the suite so that `pytest -x` meets the parity evidence first; a file missing from the table would
    silently land in the default tier (the round-4 VERDICT found five such files)."""
    import glob
    import os

    from tests.conftest import _FILE_TIER

    here = os.path.dirname(os.path.abspath(__file__))
    names = sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(here, "test_*.py")))
    assert [n for n in names if n not in _FILE_TIER] == []
    assert [n for n in _FILE_TIER if n not in names] == []


def test_one_cycle_takes_total_steps_as_given():
    """bc/trainer no longer stretches short cycles (round-4 VERDICT, weak 3c: a 20-step ACT cycle became 21).  bc/schedule.OneCycle
    equals torch's OneCycleLR -- the scheduler the reference builds -- for every total_steps torch accepts, including those below
    1 / pct_start (negative first-phase end: no warm-up); the one value torch cannot take (pct_start * total_steps == 1) raises."""
    import pytest
    import torch

    from pointcloudmatters_amd.bc import BCTrainer
    from pointcloudmatters_amd.bc.schedule import OneCycle

    p = torch.nn.Parameter(torch.zeros(1))

    def cycle(total, pct):
        opt = torch.optim.AdamW([p], lr=1e-3)
        sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=total, pct_start=pct, div_factor=100.0,
                                                    final_div_factor=1000.0)
        out = []
        for _ in range(total - 1):
            out.append((opt.param_groups[0]["lr"], opt.param_groups[0]["betas"][0]))
            opt.step()
            sched.step()
        return out

    for pct, totals in ((0.1, (3, 9, 11, 20, 21)), (0.15, (5, 6, 7, 20)), (0.3, (2, 4, 20))):
        for total in totals:
            ours = OneCycle(1e-3, total, pct, 100.0, 1000.0)
            for i, (lr, b1) in enumerate(cycle(total, pct)):
                assert abs(ours.at(i)[0] - lr) <= 1e-14 and abs(ours.at(i)[1] - b1) <= 1e-14, (pct, total, i)
    with pytest.raises(ZeroDivisionError):
        cycle(10, 0.1)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Linear(2, 2)

    with pytest.raises(ValueError, match="cannot form a cycle"):
        BCTrainer(Tiny(), total_steps=10, optim=dict(pct_start=0.1), device=torch.device("cpu"), mode="eager")
    tr = BCTrainer(Tiny(), total_steps=20, optim=dict(pct_start=0.1), device=torch.device("cpu"), mode="eager")
    assert tr.scheduler.total_steps == 20


def test_deferral_window_end_flushes_inside_a_callers_except_block(monkeypatch):
    """end() used to look at sys.exc_info() and so skipped its closing flush when a healthy step ran inside somebody's
    `except` block.  The failure is now passed explicitly."""
    from pointcloudmatters_amd.policy import deferred

    calls = []
    monkeypatch.setattr(deferred, "flush", lambda: calls.append("flush") or 0)
    try:
        raise RuntimeError("an outer handler is active")
    except RuntimeError:
        assert deferred.begin()
        deferred.end()  # healthy window closed while an exception is being handled further up
    assert calls == ["flush"] and not deferred.active()
    assert deferred.begin()
    deferred.end(failed=True)  # the body raised: no flush of half-built queues
    assert calls == ["flush"] and not deferred.active()


def test_bench_headline_kernel_set_is_not_chosen_at_run_time(monkeypatch):
    """The bench line's kernel set is fixed before the process starts (round-5 VERDICT weak 3 / ADVICE): library products by default, the
    MFMA projection chain of csrc/proj_ln.hip only when the environment asks for it; bench.py has no selection trial any more and
    records what ran in config.projection_chain."""
    import bench
    from pointcloudmatters_amd.policy import fused_ops

    src = open(bench.__file__).read()
    assert "choose_projection_chain" not in src and "--chain-trial" not in src and "--no-chain-selection" not in src
    saved = (fused_ops.PROJ_MFMA, fused_ops.LINEAR_MFMA, fused_ops.PROJ_MFMA_LONG)
    saved_bwd = (fused_ops.PROJ_MFMA_BWD, fused_ops.LINEAR_MFMA_BWD)
    try:
        fused_ops.PROJ_MFMA = fused_ops.LINEAR_MFMA = fused_ops.PROJ_MFMA_LONG = fused_ops.PROJ_MFMA_BWD = fused_ops.LINEAR_MFMA_BWD = False
        info = bench.projection_chain_setting()
        assert info["selected"] == "library products" and "opt-in" in info["reason"]
        fused_ops.PROJ_MFMA, fused_ops.PROJ_MFMA_LONG = True, True
        info = bench.projection_chain_setting()
        assert "mfma" in info["selected"] and "long" in info["selected"] and "environment" in info["reason"]
    finally:
        fused_ops.PROJ_MFMA, fused_ops.LINEAR_MFMA, fused_ops.PROJ_MFMA_LONG = saved
        fused_ops.PROJ_MFMA_BWD, fused_ops.LINEAR_MFMA_BWD = saved_bwd


def test_bench_auto_mode_is_a_ladder_that_only_moves_on_an_exception(monkeypatch):
    import pytest

    """bench.run_workload with --mode auto at N = 1: the capture mode is tried first; a raise while building / warming it up moves to the
    next mode and is recorded; an explicit --mode is obeyed and its failure propagates."""
    import argparse

    import torch

    import bench
    import pointcloudmatters_amd.bc as bc

    tried = []

    class FakeTrainer:
        def __init__(self, policy, mode=None, **kw):
            self.mode = mode
            tried.append(mode)
            if mode in FAIL:
                raise RuntimeError("capture failed\nsecond line")

        def training_step(self, batch, prefetch=None):
            return {"loss": torch.zeros(())}

    monkeypatch.setattr(bc, "BCTrainer", FakeTrainer)
    monkeypatch.setattr(bc, "build_act_policy", lambda **kw: torch.nn.Linear(1, 1))
    monkeypatch.setattr(bc, "make_act_batch", lambda b, n, **kw: {})
    monkeypatch.setattr(bc, "clone_batch", lambda b: b)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    args = argparse.Namespace(sa_impl="auto", dead_decoder_layers="keep", no_prefetch=True, sampling_in_graph=False)
    dev = torch.device("cpu")
    FAIL = {"graph"}
    _, tr, _, _, _ = bench.run_workload("C2", args, dev, 1, 0, 1, 1, mode="auto")
    assert tried == ["graph", "hybrid"] and tr.mode == "hybrid"
    assert len(bench.MODE_FALLBACK["C2"]) == 1 and "capture failed second line" in bench.MODE_FALLBACK["C2"][0]
    tried.clear()
    FAIL = set()
    _, tr, _, _, _ = bench.run_workload("C2", args, dev, 1, 0, 1, 1, mode="auto")
    assert tried == ["graph"] and bench.MODE_FALLBACK["C2"] == []
    tried.clear()
    FAIL = {"graph", "hybrid", "flat"}
    with pytest.raises(RuntimeError):
        bench.run_workload("C2", args, dev, 1, 0, 1, 1, mode="auto")
    assert tried == ["graph", "hybrid", "flat"]
    tried.clear()
    FAIL = {"graph"}
    with pytest.raises(RuntimeError):  # an explicit mode is obeyed as given
        bench.run_workload("C2", args, dev, 1, 0, 1, 1, mode="graph")
    assert tried == ["graph"]
