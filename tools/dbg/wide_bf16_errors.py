"""Per-tensor relative error (max |got - ref| / max |ref|) of the bf16 fused path against tests/golden/wide_ref.npz: the numbers
the bounds in tests/test_wide_fixture.py are set from.  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_wide_fixture import ACT_TAGS, _act_case, _digests, _dp_case, _run_act  # noqa: E402
from tests.util import grad_digest  # noqa: E402


def rel(name, got, ref):
    dig = grad_digest(name, got)
    if "full" in ref:
        s = float(np.abs(ref["full"]).max())
        return float(np.abs(dig["full"] - ref["full"]).max()) / (s + 1e-30), s
    s = float(ref["absmax"])
    rows, cols = ref["cols"].shape[0], ref["rows"].shape[1]
    e = max(float(np.abs(dig["rows"] - ref["rows"]).max()), float(np.abs(dig["cols"] - ref["cols"]).max()),
            float(np.abs(dig["right"] - ref["right"]).max()) / np.sqrt(cols), float(np.abs(dig["left"] - ref["left"]).max()) / np.sqrt(rows))
    return e / (s + 1e-30), s


def main():
    import pointcloudmatters_amd.pointops as po

    dev = torch.device("cuda", 0)
    for bf16 in (True, False):
        for tag, M in ACT_TAGS:
            fx, pol, batch = _act_case(tag, M, po, "fused", device=dev)
            out = _run_act(pol, batch, fused=True, bf16=bf16)
            print(f"== act {tag} bf16={bf16}")
            for k in ("a_hat", "is_pad_hat", "mu", "logvar", "loss", "action_loss", "kl_loss", "src"):
                r = fx[f"act.{tag}.out.{k}"]
                g = out[k].detach().float().cpu().numpy()
                print(f"  out.{k:12s} {np.abs(g - r).max() / (np.abs(r).max() + 1e-30):.3e}")
            grads = dict(pol.named_parameters())
            rows = []
            for name, ref in _digests(fx, f"act.{tag}.grad.").items():
                e, s = rel(name, grads[name].grad.detach().float().cpu().numpy(), ref)
                rows.append((e, name, s))
            for e, name, s in sorted(rows, reverse=True)[:25]:
                print(f"  {e:.3e}  {name}  (max|g| {s:.2e})")
            print("  median", np.median([r[0] for r in rows]))
        fx, pol, batch = _dp_case(po, "fused", device=dev)
        import contextlib
        with (torch.autocast("cuda", dtype=torch.bfloat16) if bf16 else contextlib.nullcontext()):
            out = pol(batch)
        out["loss"].backward()
        print(f"== dp bf16={bf16} loss rel", abs(float(out["loss"]) - float(fx["dp.out.loss"])) / float(fx["dp.out.loss"]))
        grads = dict(pol.named_parameters())
        rows = []
        for name, ref in _digests(fx, "dp.grad.").items():
            e, s = rel(name, grads[name].grad.detach().float().cpu().numpy(), ref)
            rows.append((e, name, s))
        for e, name, s in sorted(rows, reverse=True)[:25]:
            print(f"  {e:.3e}  {name}  (max|g| {s:.2e})")
        print("  median", np.median([r[0] for r in rows]))


main()
