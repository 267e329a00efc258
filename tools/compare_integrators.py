"""Megakernel vs wavefront rate per built-in scene (calibrates RT_WF_AUTO_ITEMS, rt_kernels.hip).
Usage: python tools/compare_integrators.py [W H spp]"""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("rtb", os.path.join(ROOT, "cuda-raytracer_amd/binding.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)
W, H, spp = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1920, 1080, 32)
ctx = B.Context(0)
for name, param in [("cornell", 0), ("cornell_smoke", 0), ("bvh_boxes", 0), ("random_spheres", 0), ("cornell_cluster", 0), ("final", 20), ("final", 100), ("tri_mesh", 20000),
                    ("tri_mesh", 200000), ("tri_mesh", 1000000)]:
    sc = B.Scene(name, W, H, param)
    ctx.upload(sc)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    out = []
    for mode in (1, 2):
        ctx.set_integrator(mode)
        ctx.render_accumulate(acc, W, H, 0, 2, 419, 100, sc.background)       # warm-up
        ctx.synchronize()
        ctx.reset_counters()
        ctx.render_accumulate(acc, W, H, 2, 2 + spp, 419, 100, sc.background)
        c = ctx.counters()
        out.append(c.rays / c.render_ms / 1e3)
    ctx.set_integrator(0)
    print("%-16s param %-8d bvh items %-8d megakernel %8.1f  wavefront %8.1f Mrays/s  (x%.2f)"
          % (name, param, sc.flat.contents.n_bvh_items, out[0], out[1], out[1] / out[0]), flush=True)
    sc.close()
