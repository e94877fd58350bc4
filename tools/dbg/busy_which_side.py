"""Beside a busy device (background thread replaying a GEMM graph), which side of a product-vs-framework comparison moves: our kernel or
the framework's?  Each op is evaluated on the idle device first, then 50 times under load; results are compared bit for bit with the
idle ones."""
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.policy import rows_linear  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
go = torch.randn(4120, 1536, device=dev).to(torch.bfloat16)
x = torch.randn(4120, 512, device=dev).to(torch.bfloat16)
w = (torch.randn(1536, 512, device=dev) * 0.05).to(torch.bfloat16)
xf = torch.randn(4120, 512, device=dev)
xf_g = xf.clone().requires_grad_(True)
lnw, lnb = torch.rand(512, device=dev).requires_grad_(True), torch.rand(512, device=dev).requires_grad_(True)
big = torch.randn(32 << 20, device=dev)
LOAD = os.environ.get("LOAD", "gemm")
ops = {
    "ours: bias_grad (pcm_colsum)": lambda: rows_linear.bias_grad(go, torch.bfloat16),
    "framework: go.sum(0) bf16": lambda: go.sum(0),
    "framework: go.float().sum(0)": lambda: go.float().sum(0),
    "library: x @ w.T (bf16 GEMM)": lambda: x @ w.t(),
    "library: go.T @ x (bf16 GEMM, long K)": lambda: go.t() @ x,
    "framework: layer_norm fp32": lambda: torch.nn.functional.layer_norm(xf, (512,)),
    "framework: xf * 1.5 + 2": lambda: xf * 1.5 + 2,
    "framework: softmax": lambda: torch.softmax(xf, -1),
    "framework: bf16 cast": lambda: xf.to(torch.bfloat16),
    "framework: go[:, :64].sum(0) bf16": lambda: go[:, :64].contiguous().sum(0),
    "framework: go.sum(0) rows 512": lambda: go[:512].sum(0),
    "framework: go.sum(1) bf16": lambda: go.sum(1),
    "framework: (go*go).sum() bf16": lambda: (go * go).sum(),
    "framework: fp32 (131072 x 8).sum(0)": lambda: xf.reshape(-1, 8)[:131072].sum(0),
    "framework: bf16 (800, 7).sum(0)": lambda: go[:800, :7].contiguous().sum(0),
    "framework: bf16 (800, 1).sum(0)": lambda: go[:800, :1].contiguous().sum(0),
    "framework: bf16 (8, 512).sum(0)": lambda: go[:8, :512].contiguous().sum(0),
    "framework: bf16 (816, 512).sum(0)": lambda: go[:816, :512].contiguous().sum(0),
    "framework: bf16 (800, 64).sum(0)": lambda: go[:800, :64].contiguous().sum(0),
    "framework: bf16 .sum() all": lambda: go.sum(),
    "framework: bf16 .mean(1)": lambda: go.mean(1),
    "framework: fp32 layer_norm backward": lambda: torch.autograd.grad(torch.nn.functional.layer_norm(xf_g, (512,), lnw, lnb), (xf_g, lnw, lnb), xf)[1],
    "framework: cat bf16": lambda: torch.cat([go[:, :512], x], 0),
    "framework: bf16 add": lambda: x + x,
    "framework: bf16 mul scalar": lambda: x * 0.5,
    "framework: bf16 -> fp32 cast": lambda: go.float(),
    "framework: fp32 (5600, 512) layer_norm": lambda: torch.nn.functional.layer_norm(torch.cat([xf, xf[:1480]]), (512,), lnw, lnb),
    "framework: bf16 softmax": lambda: torch.softmax(x, -1),
    "framework: bf16 relu": lambda: torch.relu(x),
    "framework: bf16 gelu": lambda: torch.nn.functional.gelu(x),
    "framework: bf16 mse": lambda: torch.nn.functional.mse_loss(x, x * 0.5, reduction="none").mean(),
    "framework: fp32 index_select": lambda: xf[torch.arange(0, 4120, 2, device=dev)],
    "framework: bf16 max over dim": lambda: go.max(0)[0],
}
idle = {k: f() for k, f in ops.items()}
torch.cuda.synchronize()
a = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
(a @ a)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        for _ in range(100):
            if LOAD == "gemm":
                a @ a
            else:
                big.mul_(1.0000001)
torch.cuda.synchronize()
stop = threading.Event()


def loop():
    bg = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(bg):
        while not stop.is_set():
            g.replay()
            bg.synchronize()


t = threading.Thread(target=loop, daemon=True)
t.start()
for k, f in ops.items():
    outs = [f() for _ in range(50)]
    torch.cuda.synchronize()
    bad = sum(int(not torch.equal(o, idle[k])) for o in outs)
    worst = max(float((o.float() - idle[k].float()).abs().max()) for o in outs)
    print(f"{k:40s}: {bad:2d} / 50 differ from the idle result (largest |difference| {worst:.4g})", flush=True)
stop.set()
t.join()
