#!/usr/bin/env python3
"""What one GPU says about strong scaling of BASELINE configs[3] (1M triangles, 1920x1080, 256 spp, spp-sharded):
the rate of a render call of 256 / 128 / 64 / 32 samples, which is rank 0's share at N = 1 / 2 / 4 / 8.  The implied
efficiency at N is rate(256 / N) / rate(256): what the shards lose before the reduce is counted (the reduce moves
24.9 MB once per frame).  Writes profiles/<tag>_strong_scaling_projection.json.
  python3 tools/strong_projection.py [tag] [--tiles]    --tiles adds the image-tile split (row bands at full spp)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r03"
    tiles = "--tiles" in sys.argv
    b = entry.load_binding()
    W, H, SPP = 1920, 1080, 256
    ctx = b.Context(0)
    sc = b.Scene("tri_mesh", W, H, 1000000)
    ctx.upload(sc)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.render_accumulate(acc, W, H, 0, 8, 419, 100, sc.background)        # warm-up: allocations, stream search
    ctx.synchronize()
    rows = []
    first = 8
    for n in (1, 2, 4, 8):
        spp = SPP // n
        best = None
        for rep in range(2):
            ctx.reset_counters()
            ctx.render_accumulate(acc, W, H, first, first + spp, 419, 100, sc.background)
            c = ctx.counters()
            first += spp
            r = {"n_gpus": n, "spp_per_gpu": spp, "ms": c.render_ms, "Grays_per_s": c.rays / (c.render_ms * 1e-3) / 1e9, "wf_pools": int(c.wf_pools)}
            if best is None or r["Grays_per_s"] > best["Grays_per_s"]:
                best = r
        rows.append(best)
        print(json.dumps(best), flush=True)
    base = rows[0]["Grays_per_s"]
    for r in rows:
        r["implied_efficiency"] = r["Grays_per_s"] / base
        r["implied_node_Grays_per_s"] = r["Grays_per_s"] * r["n_gpus"]
    out = {"workload": "tri_mesh 1M triangles, %dx%d, %d spp in all, spp-sharded" % (W, H, SPP), "one_gpu_calls": rows,
           "note": "rate of ONE GPU rendering its share of the samples in one call; the reduce (24.9 MB over xGMI) is not in it"}
    if tiles:
        trows = []
        for n in (2, 4, 8):
            y1 = H // n                   # rank 0's band; the bands differ in content, so this is an indication only
            ctx.reset_counters()
            ctx.render_window(acc, W, H, (0, 0, W, y1), first, first + SPP, 419, 100, sc.background)
            c = ctx.counters()
            first += SPP
            trows.append({"n_gpus": n, "band_rows": y1, "ms": c.render_ms, "Grays_per_s": c.rays / (c.render_ms * 1e-3) / 1e9})
            print(json.dumps(trows[-1]), flush=True)
        out["image_tiles_rank0_band"] = trows
    try:
        out["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        out["commit"] = "unknown"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "%s_strong_scaling_projection.json" % tag)
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
