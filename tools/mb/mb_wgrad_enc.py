"""Encoder weight gradients (4 layers, 4120 rows): per-layer split-K (bmm of s row blocks + closing sum, what rows_linear
does) vs ONE batched unsplit product over the layers vs a batched split product.  Graph-replay timing.
python tools/mb/mb_wgrad_enc.py"""
import torch

from mb_wgrad_batch import graph_time
from pointcloudmatters_amd.policy import rows_linear

dev = "cuda"
for rows, m, k, dt in [(4120, 512, 512, torch.bfloat16), (4120, 1024, 512, torch.bfloat16), (4120, 32, 512, torch.float32),
                       (4120, 512, 32, torch.float32), (800, 512, 512, torch.bfloat16)]:
    for n in (4, 14, 16, 21):
        if rows == 800 and n == 4 or rows != 800 and n != 4:
            continue
        gos = torch.randn(n, rows, m, device=dev).to(dt)
        xs = torch.randn(n, rows, k, device=dev).to(dt)

        def per_layer():
            for i in range(n):
                rows_linear._weight_grad(gos[i], xs[i], dt)

        def batched():
            torch.bmm(gos.transpose(1, 2), xs)

        res = {"per layer (rows_linear)": graph_time(per_layer), "one bmm": graph_time(batched)}
        if rows > 2048:
            for s in (2, 4):
                c = rows // s
                a = gos[:, : s * c].reshape(n * s, c, m)
                b = xs[:, : s * c].reshape(n * s, c, k)
                res[f"bmm {n}x{s} (+sums not timed)"] = graph_time(lambda: torch.bmm(a.transpose(1, 2), b, out_dtype=torch.float32) if dt != torch.float32 else torch.bmm(a.transpose(1, 2), b))
        print((rows, m, k, str(dt)[6:], n), {k_: "%.1f us" % v for k_, v in res.items()}, flush=True)
