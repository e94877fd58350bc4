#!/usr/bin/env python
"""Policy-side latency of one rollout step on one MI355X: eager vs ONE hipGraph.

  ACT : ACTPCD forward without actions (act.py:177-182) on 1 cloud of N points -> (1, 100, 7) chunk
  DP  : DiffusionUnetPcdPolicy.predict_action: encoder on To=2 clouds + 100 DDPM iterations of the 255.6 M-parameter
        U-Net (diffusion_unet_image_policy.py:106-229)

Prints one JSON line per case: {"case", "mode", "ms_per_call", "calls"}.  Synthetic observations, random weights."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, calls, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / calls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--tokens", type=int, default=2048)
    ap.add_argument("--calls", type=int, default=20)
    ap.add_argument("--cases", default="act,dp")
    a = ap.parse_args()
    from pointcloudmatters_amd.bc import build_act_policy, build_dp_policy, make_act_batch, make_dp_batch
    from pointcloudmatters_amd.policy import fused_ops
    from pointcloudmatters_amd.policy.rollout import graphed_act, graphed_dp

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    for case in a.cases.split(","):
        if case == "act":
            pol = build_act_policy(pcd_npoints=a.tokens, sa_impl="fused").to(dev).eval()
            b = make_act_batch(1, a.points, device=dev)
            obs = {"qpos": b["qpos"], "goal_cond": b["goal_cond"], "pcds": b["pcds"]}

            def eager():
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                    return pol(dict(obs, pcds=dict(obs["pcds"])))["a_hat"]

            make = graphed_act
        else:
            pol = build_dp_policy(pcd_npoints=a.tokens, sa_impl="fused").to(dev).eval()
            b = make_dp_batch(1, a.points, device=dev)
            obs = {"obs": {"pcds": b["obs"]["pcds"], "qpos": b["obs"]["qpos"][:, :2].contiguous()}}

            def eager():
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                    o = dict(obs["obs"])
                    o["pcds"] = dict(o["pcds"])
                    return pol.predict_action({"obs": o})["action"]

            make = graphed_dp
        calls = a.calls if case == "act" else max(3, a.calls // 4)
        ms = timed(eager, calls)
        print(json.dumps({"case": case, "mode": "eager", "points": a.points, "tokens": a.tokens, "ms_per_call": round(ms, 3), "calls": calls}), flush=True)
        with fused_ops.activate(fused_ops.FusedContext(dev)):
            runner = make(pol, obs)
        ms = timed(lambda: runner(obs), calls)
        print(json.dumps({"case": case, "mode": "graph", "points": a.points, "tokens": a.tokens, "ms_per_call": round(ms, 3), "calls": calls}), flush=True)
        del runner, pol
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
