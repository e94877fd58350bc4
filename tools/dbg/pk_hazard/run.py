"""Runs tools/dbg/pk_hazard/pk_hazard.hip on a side stream, alone and beside a replayed graph of library GEMMs on the main stream."""
import ctypes, os, subprocess
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libpk_hazard.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", os.path.join(HERE, "pk_hazard.hip"), "-o", SO])
L = ctypes.CDLL(SO)
L.pk_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
blocks, iters = 64, 20000
x = torch.rand(blocks * 128 * 2, device=dev)
amat = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
(amat @ amat); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s2):
    with torch.cuda.graph(g, stream=s2):
        for _ in range(200):
            amat @ amat
torch.cuda.synchronize()
side = torch.cuda.Stream()
names = ["v_pk_add_f32 (plain)", "v_pk_add_f32 op_sel_hi:[1,0]", "v_pk_add_f32 op_sel_hi:[1,0] neg", "v_pk_add_f32 op_sel:[0,1] neg", "v_pk_mul_f32 op_sel_hi:[1,0]",
         "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_add_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[1,0]", "v_pk_mul_f32 op_sel:[0,1]",
         "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1]"]
for v, name in enumerate(names):
    res = {}
    for load in ("idle", "beside GEMMs"):
        outs = []
        for it in range(20):
            if load != "idle":
                g.replay()
            with torch.cuda.stream(side):
                bad = torch.zeros(blocks * 128, dtype=torch.int32, device=dev)
                assert L.pk_launch(v, blocks, iters, x.data_ptr(), bad.data_ptr(), side.cuda_stream) == 0
            outs.append(bad)
        torch.cuda.synchronize()
        tot = sum(int(o.sum()) for o in outs)
        res[load] = tot
    n = 20 * blocks * 128 * iters
    print(f"{name:38s}: wrong results idle {res['idle']:>10d}   beside GEMMs {res['beside GEMMs']:>10d}   of {n:.2e} evaluations each", flush=True)
