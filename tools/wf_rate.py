#!/usr/bin/env python3
"""Rate of one render call (after a warm-up call): python tools/wf_rate.py SCENE PARAM W H SPP [integrator 0|1|2]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    name, param, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    integ = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    b = entry.load_binding()
    ctx = b.Context(0)
    sc = b.Scene(name, W, H, param)
    ctx.upload(sc)
    ctx.set_integrator(integ)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.render_accumulate(acc, W, H, 0, min(spp, 8), 419, 100, sc.background)
    ctx.synchronize()
    ctx.reset_counters()
    ctx.render_accumulate(acc, W, H, 8, 8 + spp, 419, 100, sc.background)
    c = ctx.counters()
    print(json.dumps({"scene": name, "param": param, "frame": [W, H], "spp": spp, "Grays_per_s": c.rays / (c.render_ms * 1e-3) / 1e9, "ms": c.render_ms,
                      "pools": os.environ.get("RT_WF_POOLS", "default"), "image_mean": float(acc.mean()) / (spp + min(spp, 8))}))


main()
