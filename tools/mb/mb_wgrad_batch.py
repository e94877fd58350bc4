"""Seven weight gradients of one shape (the decoder's layers): seven products vs stack + one batched product + scatter.
Timed as hipGraph replays (the step is replayed, launch gaps matter).  python tools/mb/mb_wgrad_batch.py"""
import torch

dev = "cuda"


def graph_time(f, inner=10, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(inner):
                f()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps / inner * 1e3


for rows, m, k, dt, odt in [(800, 512, 512, torch.bfloat16, torch.bfloat16), (800, 1024, 512, torch.bfloat16, torch.bfloat16),
                            (800, 3200, 512, torch.float32, torch.float32), (800, 512, 3200, torch.float32, torch.float32),
                            (816, 512, 512, torch.bfloat16, torch.bfloat16)]:
    for n in (7, 4):
        gos = [torch.randn(rows, m, device=dev).to(dt) for _ in range(n)]
        xs = [torch.randn(rows, k, device=dev).to(dt) for _ in range(n)]
        outs = [torch.empty(m, k, device=dev, dtype=odt) for _ in range(n)]

        def separate():
            for go, x, o in zip(gos, xs, outs):
                torch.mm(go.t(), x, out=o)

        def batched(pad=0):
            a = torch.stack(gos + gos[:pad])
            b = torch.stack(xs + xs[:pad])
            r = torch.bmm(a.transpose(1, 2), b)
            torch._foreach_copy_(outs, list(r[:n].unbind(0)))

        A = torch.stack(gos)
        B = torch.stack(xs)

        def bmm_only():
            torch.bmm(A.transpose(1, 2), B)

        res = {"separate": graph_time(separate), "stack+bmm+scatter": graph_time(batched), "bmm only": graph_time(bmm_only)}
        if n == 7:
            res["padded to 8"] = graph_time(lambda: batched(1))
        print((rows, m, k, str(dt)[6:], n), {k_: "%.1f us" % v for k_, v in res.items()}, flush=True)
