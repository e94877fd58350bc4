"""How far apart are (a) the fused GPU composition and (b) the fp32 CPU composition from the SAME composition in fp64?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pointops_cpu
from pointcloudmatters_amd import pointops
from pointcloudmatters_amd.bc import make_act_batch
from pointcloudmatters_amd.policy.pointnet2 import PointNeXtBackbone, SAStageMSG

DEV = "cuda"
for n_points in (2048, 4096):
    torch.manual_seed(2)
    scales = ((16, 0.06, 32), (32, None, 64))
    mk = lambda po, impl: (PointNeXtBackbone(6, 32, 2, 16, 32, pointops=po, sa_impl=impl), SAStageMSG(32, n_points // 4, scales, pointops=po, sa_impl=impl))
    c_nx, c_msg = mk(pointops_cpu, "reference")
    d_nx, d_msg = mk(pointops_cpu, "reference")
    g_nx, g_msg = mk(pointops, "fused")
    for a, b in ((d_nx, c_nx), (d_msg, c_msg), (g_nx, c_nx), (g_msg, c_msg)):
        a.load_state_dict(b.state_dict())
    d_nx, d_msg = d_nx.double(), d_msg.double()
    g_nx, g_msg = g_nx.to(DEV), g_msg.to(DEV)
    res = {}
    for tag, nx, msg, dev, dt in (("f32", c_nx, c_msg, "cpu", torch.float32), ("f64", d_nx, d_msg, "cpu", torch.float64), ("gpu", g_nx, g_msg, DEV, torch.float32)):
        pcd = make_act_batch(2, n_points, seed=33, ragged=True, device=dev)["pcds"]
        if dt == torch.float64:
            pcd = {k: (v.double() if v.is_floating_point() else v) for k, v in pcd.items()}
        try:
            x = nx(pcd)
            n_p, tok, n_o = msg(pcd["coord"], x, pcd["offset"])
            tok.square().mean().backward()
        except Exception as e:
            print(tag, "failed:", type(e).__name__, e); continue
        g = {k: p.grad.detach().double().cpu() for m in (nx, msg) for k, p in m.named_parameters() if p.grad is not None}
        res[tag] = (x.detach().double().cpu(), tok.detach().double().cpu(), g)
    if "f64" in res:
        for tag in ("f32", "gpu"):
            x, t, g = res[tag]; x0, t0, g0 = res["f64"]
            worst = max(((g[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)).item() for k in g0)
            print(n_points, tag, "vs f64: x %.2e tok %.2e worst-grad %.2e" % (((x - x0).abs().max() / x0.abs().max()).item(), ((t - t0).abs().max() / t0.abs().max()).item(), worst))
    x, t, g = res["gpu"]; x0, t0, g0 = res["f32"]
    gall = torch.cat([v.flatten() for v in g0.values()]).norm().item()
    rows = sorted(((((g[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)).item(), k, g0[k].norm().item(), (g[k] - g0[k]).norm().item()) for k in g0), reverse=True)
    print(n_points, "all-param grad norm %.3e" % gall)
    for r in rows[:6]: print("   rel %.2e  %-40s |g| %.3e  |dg| %.3e  dg/|g_all| %.2e" % (r[0], r[1], r[2], r[3], r[3] / gall))
