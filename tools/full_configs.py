"""BASELINE.json configs[1..4] at their full frame sizes and sample counts on one GPU, rendered in sample chunks
(one rt_render_accumulate per chunk, a progress line after each) and written to gpurun_out/full_configs.json.
A measurement aid: bench.py stays the contract; this records that every config completes at full size and at
what rate.  Each entry also carries `cpu`: the oracle (oracle/rt_oracle.c, the scalar restatement of the reference) on
this machine's host cores on a bounded sample of the same scene: 1 thread and all the threads this job may use (the
GPU box's share is 16 although it shows more), flat-list semantics for configs 0-2 and the oracle's own BVH traversal
for configs 3-4.  configs[0] (random spheres 400x225 @ 16 spp, CPU only) is `random_spheres_cpu`: the
whole config on the oracle, and the same frame on the GPU beside it.
Usage: python tools/full_configs.py [name ...] [--cpu-only]"""
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
O = load("rto", os.path.join(ROOT, "oracle/oracle.py"))          # the CPU baseline beside the GPU numbers, never in the timed GPU path
CPU_THREADS = max(1, min(16, len(os.sched_getaffinity(0))))


def cpu_rates(scene, param, background, use_bvh, W, H, spp_all, spp_one):
    """oracle rays/s on all threads and on one, on a W x H frame of the same scene and camera"""
    sc = B.Scene(scene, W, H, param)
    orc = O.Oracle()
    t0 = time.time()
    _, st = orc.render(sc.flat, W, H, 0, spp_all, 419, 100, background, use_bvh=use_bvh, threads=CPU_THREADS)
    dt = time.time() - t0
    t0 = time.time()
    _, st1 = orc.render(sc.flat, W, H, 0, spp_one, 419, 100, background, use_bvh=use_bvh, threads=1)
    dt1 = time.time() - t0
    return {"kind": "port (oracle/rt_oracle.c)", "traversal": "host BVH" if use_bvh else "flat list (HitableList semantics)",
            "sample": "%s %dx%d, %d spp on %d threads, %d spp on 1" % (scene, W, H, spp_all, CPU_THREADS, spp_one),
            "threads": CPU_THREADS, "cores_visible": os.cpu_count(), "Mrays_per_s": st.rays / dt / 1e6, "seconds": dt,
            "Mrays_per_s_1thread": st1.rays / dt1 / 1e6, "seconds_1thread": dt1, "rays_per_sample": st.rays / st.samples}


# per config: (use_bvh, W, H, spp on all threads, spp on one thread) of the bounded CPU sample
CPU_SAMPLE = {"cornell": (False, 128, 128, 256, 16), "cornell_smoke": (False, 128, 128, 256, 16),
              "tri_mesh": (True, 192, 108, 64, 8), "final": (True, 192, 108, 32, 4)}
# name: (scene, param, W, H, spp, chunk)
CONFIGS = {
    "cornell": ("cornell", 0, 1024, 1024, 1024, 1024),
    "cornell_smoke": ("cornell_smoke", 0, 1024, 1024, 2048, 1024),
    "tri_mesh": ("tri_mesh", 1000000, 1920, 1080, 256, 64),
    "final": ("final", 100, 3840, 2160, 4096, 128),
}
cpu_only = "--cpu-only" in sys.argv
names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["random_spheres_cpu"] + list(CONFIGS)
ctx = None if cpu_only else B.Context(0)
out = {}
if "random_spheres_cpu" in names:
    # BASELINE.json configs[0]: ~500 spheres, 400 x 225 @ 16 spp, sky integrator, on the host-compiled scalar code: the
    # WHOLE config, flat list (the reference's HitableList::hit over 488 spheres), all threads and one thread
    names.remove("random_spheres_cpu")
    W, H, spp = 400, 225, 16
    sc = B.Scene("random_spheres", W, H, 0)
    entry = {"scene": "random_spheres", "width": W, "height": H, "spp": spp, "objects": int(sc.flat.contents.n_objects),
             "cpu": cpu_rates("random_spheres", 0, sc.background, False, W, H, spp, spp),
             "cpu_bvh": cpu_rates("random_spheres", 0, sc.background, True, W, H, spp, spp)}
    if not cpu_only:
        ctx.upload(sc)
        acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.render_accumulate(acc, W, H, 0, spp, 419, 100, sc.background)      # warm-up (loads the kernels)
        ctx.synchronize()
        ctx.reset_counters()
        acc.zero_()
        torch.cuda.synchronize()
        ctx.render_accumulate(acc, W, H, 0, spp, 419, 100, sc.background)
        c = ctx.counters()
        ref, _ = O.Oracle().render(sc.flat, W, H, 0, spp, 419, 100, sc.background, threads=CPU_THREADS)
        got = acc.cpu().numpy().reshape(H, W, 3)
        import numpy as np
        bad = np.any(np.abs(got - ref) > 1e-4 * (1 + np.abs(ref)), axis=2)
        entry["gpu"] = {"kernel_ms": c.render_ms, "rays": int(c.rays), "Mrays_per_s": c.rays / c.render_ms / 1e3,
                        "integrator": {1: "megakernel", 2: "wavefront"}[int(c.last_integrator)],
                        "pixels_differing_from_the_oracle_sample_for_sample": int(bad.sum()), "pixels": int(bad.size)}
    out["random_spheres_cpu"] = entry
    print(json.dumps({"random_spheres_cpu": entry}), flush=True)
for name in names:
    scene, param, W, H, spp, chunk = CONFIGS[name]
    t0 = time.time()
    sc = B.Scene(scene, W, H, param)
    cpu = cpu_rates(scene, param, sc.background, *CPU_SAMPLE[name])
    if cpu_only:
        out[name] = {"scene": scene, "param": param, "cpu": cpu}
        print(json.dumps({name: out[name]}), flush=True)
        continue
    t0 = time.time()
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
        "image_mean": float(rgb.mean()), "image_finite": bool(torch.isfinite(rgb).all()), "cpu": cpu,
    }
    print(json.dumps({name: out[name]}), flush=True)
    del acc, rgb
    sc.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "full_configs.json"), "w"), indent=1)
