"""Cost of a fork/join (a second captured stream) inside a replayed hipGraph, per fork; and of a tiny dependent kernel."""
import torch
dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev); y = torch.zeros(1024, device=dev)
side = torch.cuda.Stream()
def build(n_kernels, n_forks, side_len):
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            per = max(1, n_kernels // max(n_forks, 1))
            k = 0
            for f in range(max(n_forks, 1)):
                if n_forks:
                    side.wait_stream(s)
                    with torch.cuda.stream(side):
                        for _ in range(side_len): y.add_(1.0)
                for _ in range(per):
                    x.add_(1.0); k += 1
                if n_forks:
                    s.wait_stream(side)
    return g
def timed(g, n=20):
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
base = timed(build(600, 0, 0))
print("600 dependent tiny kernels: %.1f us  (%.2f us each)" % (base, base / 600))
for forks, side_len in ((1, 10), (6, 10), (30, 10), (100, 5), (30, 0)):
    t = timed(build(600, forks, side_len))
    print("  + %3d fork/joins with %2d side kernels each: %.1f us  (%.1f us per fork beyond the base)" % (forks, side_len, t, (t - base) / forks))
