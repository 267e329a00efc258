"""BASELINE.json configs[1..4] at their full frame sizes and sample counts on one GPU, rendered in sample chunks
(one rt_render_accumulate per chunk, a progress line after each) and written to gpurun_out/full_configs.json.
A measurement aid: bench.py stays the contract; this records that every config completes at full size and at
what rate.  Usage: python tools/full_configs.py [name ...]"""
import importlib.util
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


B = load("rtb", os.path.join(ROOT, "cuda-raytracer_amd/binding.py"))
# name: (scene, param, W, H, spp, chunk)
CONFIGS = {
    "cornell": ("cornell", 0, 1024, 1024, 1024, 1024),
    "cornell_smoke": ("cornell_smoke", 0, 1024, 1024, 2048, 1024),
    "tri_mesh": ("tri_mesh", 1000000, 1920, 1080, 256, 64),
    "final": ("final", 100, 3840, 2160, 4096, 128),
}
names = sys.argv[1:] or list(CONFIGS)
ctx = B.Context(0)
out = {}
for name in names:
    scene, param, W, H, spp, chunk = CONFIGS[name]
    t0 = time.time()
    sc = B.Scene(scene, W, H, param)
    ctx.upload(sc)
    build_s = time.time() - t0
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.reset_counters()
    t0 = time.time()
    for s0 in range(0, spp, chunk):
        ctx.render_accumulate(acc, W, H, s0, min(spp, s0 + chunk), 419, 100, sc.background)
        ctx.synchronize()
        print("%s: %d / %d spp, %.1f s" % (name, min(spp, s0 + chunk), spp, time.time() - t0), flush=True)
    wall = time.time() - t0
    c = ctx.counters()
    rgb = ctx.resolve(acc, W, H, spp, gamma=False)
    ctx.synchronize()
    out[name] = {
        "scene": scene, "param": param, "width": W, "height": H, "spp": spp, "chunk_spp": chunk,
        "objects": int(sc.flat.contents.n_objects), "bvh_items": int(sc.flat.contents.n_bvh_items),
        "scene_build_upload_s": build_s, "wall_s": wall, "kernel_ms": c.render_ms, "launches": int(c.render_launches),
        "rays": int(c.rays), "samples": int(c.samples), "Mrays_per_s": c.rays / c.render_ms / 1e3,
        "Msamples_per_s": c.samples / c.render_ms / 1e3, "rays_per_sample": c.rays / c.samples,
        "image_mean": float(rgb.mean()), "image_finite": bool(torch.isfinite(rgb).all()),
    }
    print(json.dumps({name: out[name]}), flush=True)
    del acc, rgb
    sc.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "full_configs.json"), "w"), indent=1)
