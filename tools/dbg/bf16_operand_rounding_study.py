import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch, torch.nn.functional as F
torch.set_num_threads(4)
from tests.golden import make_golden as mg
from tests.util import seeded_fill
ref = mg.install_reference()
from pointcloudmatters_amd.bc import make_act_batch
from pointcloudmatters_amd.policy import PointNet
import pointcloudmatters_amd.policy.pointnet as pn_mod
c = mg.WIDE
class RoundIn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x): return x.bfloat16().float()
    @staticmethod
    def backward(ctx, g): return g.bfloat16().float()   # the backward products see bf16-rounded gradients too
def lin_b16in(x, w, b=None):
    with torch.autocast("cpu", enabled=False):
        return F.linear(RoundIn.apply(x.float()), RoundIn.apply(w), b)
real_lr = pn_mod.linear_rows
def off(f):
    def g(*a, **k):
        with torch.autocast("cpu", enabled=False): return f(*a, **k)
    return g
def run(B, mode, seed=300):
    backbone = PointNet(in_channels=6, num_classes=0)
    model = mg._ref_act(ref, c, 128, backbone)
    seeded_fill(model, mg.WIDE_SEED); model.train()
    pn_mod.linear_rows = real_lr
    if mode in ("BS", "bS", "bs"):
        backbone.forward = off(backbone.forward); model.pcd_sampling = off(model.pcd_sampling)
    if mode in ("bS", "bs"):
        pn_mod.linear_rows = lin_b16in
    if mode == "bs":   # SA linear: xyz columns fp32, feature columns bf16-in / fp32-out (what the fused kernel could do)
        lin = model.linear
        def fwd(g):  # g (m, K, 3 + C)
            with torch.autocast("cpu", enabled=False):
                w = lin.weight
                return F.linear(g[..., :3].float(), w[:, :3]) + F.linear(RoundIn.apply(g[..., 3:].float()), RoundIn.apply(w[:, 3:]))
        lin.forward = fwd
    eps = torch.randn(B, c["latent_dim"], generator=torch.Generator().manual_seed(25))
    batch = make_act_batch(B, 150, seed=seed, ragged=True, num_queries=c["num_queries"])
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=mode != "fp32"):
        mg._run_ref_act(ref, model, batch, eps)
    pn_mod.linear_rows = real_lr
    return {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
for B in (2, 8):
    for seed in (300, 301):
        g32 = run(B, "fp32", seed)
        for mode in ("BS", "bS", "bs"):
            g16 = run(B, mode, seed)
            errs = {n: ((g16[n]-g32[n]).abs().max().item()/g32[n].abs().max().item()) for n in g32 if g32[n].abs().max().item() >= 1e-6}
            v = np.array(list(errs.values())); w = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
            print(f"B={B} seed={seed} mode={mode}: median {np.median(v):.4f} p90 {np.quantile(v,.9):.4f} worst {v.max():.4f}  {[(n, round(e,3)) for n,e in w]}", flush=True)
