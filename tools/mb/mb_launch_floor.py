"""What does one DEPENDENT kernel node cost inside a replayed hipGraph, and do the runtime's knobs move it?

The C2 step is ~460 dependent launches in 4.37 ms; most of the chain's kernels run 4-9 us although their arithmetic needs < 2.
This probe replays graphs of N dependent tiny kernels (a one-element add; a 4120 x 512 bf16 elementwise add; an 800 x 512 x 512
bf16 GEMM) and prints us / node, once per environment variant (each in its own process: the knobs are read at HIP start-up).

    python tools/mb/mb_launch_floor.py            # all variants
"""
import json
import os
import subprocess
import sys

VARIANTS = [
    {},
    {"HIP_FORCE_DEV_KERNARG": "1"},
    {"HIP_FORCE_DEV_KERNARG": "0"},
    {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"},
    {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"},
    {"AMD_DIRECT_DISPATCH": "0"},
    {"GPU_MAX_HW_QUEUES": "1"},
    {"HSA_ENABLE_INTERRUPT": "0"},
    {"HIP_FORCE_DEV_KERNARG": "1", "DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"},
    {"AMD_OPT_FLUSH": "0"},
    {"ROC_SYSTEM_SCOPE_SIGNAL": "0"},
    {"DEBUG_HIP_GRAPH_BATCH_SIZE": "1"},
    {"DEBUG_HIP_GRAPH_BATCH_SIZE": "1024"},
    {"DEBUG_HIP_FORCE_GRAPH_QUEUES": "1"},
    {"DEBUG_HIP_DYNAMIC_QUEUES": "0"},
    {"ROC_USE_FGS_KERNARG": "0"},
    {"ROC_ACTIVE_WAIT_TIMEOUT": "1000"},
]


def child():
    import torch

    dev = torch.device("cuda")
    out = {}

    def timed(fn, n, reps=20):
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(n):
                    fn()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (reps * n)

    one = torch.zeros(1, device=dev)
    out["add_1elem"] = timed(lambda: one.add_(1.0), 400)
    big = torch.zeros(4120, 512, device=dev, dtype=torch.bfloat16)
    out["add_4120x512_bf16"] = timed(lambda: big.add_(1.0), 400)
    a = torch.randn(800, 512, device=dev, dtype=torch.bfloat16)
    w = torch.randn(512, 512, device=dev, dtype=torch.bfloat16) * 0.04
    c = torch.empty(800, 512, device=dev, dtype=torch.bfloat16)

    def chain():
        torch.mm(a, w, out=c)
        torch.mm(c, w, out=a)

    out["mm_800x512x512_bf16"] = timed(chain, 100) / 2
    a2 = torch.randn(4120, 512, device=dev, dtype=torch.bfloat16)
    c2 = torch.empty(4120, 512, device=dev, dtype=torch.bfloat16)

    def chain2():
        torch.mm(a2, w, out=c2)
        torch.mm(c2, w, out=a2)

    out["mm_4120x512x512_bf16"] = timed(chain2, 100) / 2
    # eager (no graph) dependent launches, for the hybrid / eager modes
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(50):
        one.add_(1.0)
    e0.record()
    for _ in range(2000):
        one.add_(1.0)
    e1.record()
    torch.cuda.synchronize()
    out["eager_add_1elem"] = e0.elapsed_time(e1) * 1e3 / 2000
    print(json.dumps({k: round(v, 3) for k, v in out.items()}))


def main():
    if "--child" in sys.argv:
        return child()
    for v in VARIANTS:
        env = dict(os.environ, **v)
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True, timeout=600)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED: " + r.stderr[-300:]
        print(json.dumps(v), line, flush=True)


if __name__ == "__main__":
    main()
