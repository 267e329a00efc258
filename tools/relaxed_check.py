#!/usr/bin/env python3
"""Judges a build of the library by BASELINE.json north_star's OWN bars instead of bit equality with the oracle:
primary-hit ids identical away from near-ties, t within 1e-5 relative, converged-image linear RMSE <= 1e-3 — and
times it.  Used to measure what bit-exactness costs: run once with the shipped library and once
with RT_AMD_LIB=tools/diag/librt_amd_relaxed.so (make -C cuda-raytracer_amd relaxed).  One JSON line on stdout."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SCENES = ["kat", "cornell", "lit_sphere", "bvh_boxes", "random_spheres", "cornell_cluster", "final", "tri_mesh", "final_custom"]


def main():
    import torch
    b = entry.load_binding()
    out = {"library": os.path.relpath(b.LIB_PATH, ROOT), "scenes": {}}
    ctx = b.Context(0)
    worst_t, id_flips, id_flips_off_tie, n_px = 0.0, 0, 0, 0
    for name in SCENES + ["tri_mesh_full", "final_full"]:
        g = np.load(os.path.join(GOLD, "geom_%s_ref.npz" % name))
        W, H = int(g["width"]), int(g["height"])
        sc = b.Scene(name.replace("_full", ""), W, H, int(g["param"]))
        ctx.upload(sc)
        prim, t = ctx.trace_primary(W, H)
        ep, et = g["primary_prim"], g["primary_t"]
        hit = ep >= 0
        both = hit & (prim >= 0)
        rel = np.abs(t[both] - et[both]) / np.abs(et[both])
        differ = prim != ep
        # a flipped id counts as a near-tie when the two t agree to 1e-5 relative (two surfaces that close)
        near = differ & both & (np.abs(t - et) <= 1e-5 * np.abs(et))
        off = differ & ~near
        out["scenes"][name] = {"pixels": int(hit.size), "id_flips": int(differ.sum()), "id_flips_off_tie": int(off.sum()),
                               "t_rel_err_max": float(rel.max()) if rel.size else 0.0, "t_bit_equal": bool(np.array_equal(t, et))}
        worst_t = max(worst_t, out["scenes"][name]["t_rel_err_max"])
        id_flips += int(differ.sum()); id_flips_off_tie += int(off.sum()); n_px += hit.size
    out["primary"] = {"pixels": n_px, "id_flips": id_flips, "id_flips_off_tie": id_flips_off_tie, "t_rel_err_max": worst_t,
                      "pass": id_flips_off_tie == 0 and worst_t <= 1e-5}

    # converged image, Cornell (different seed than the golden)
    g = np.load(os.path.join(GOLD, "image_cornell_oracle.npz"))
    W, H, spp = int(g["width"]), int(g["height"]), 4 * int(g["spp"])
    sc = b.Scene("cornell", W, H)
    ctx.upload(sc)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    tot = torch.zeros(H * W * 3, dtype=torch.float64, device="cuda")
    ctx.set_integrator(2)
    for s0 in range(0, spp, 65536):
        acc.zero_()
        torch.cuda.synchronize()
        ctx.render_accumulate(acc, W, H, s0, min(spp, s0 + 65536), 99991, 100, sc.background)
        ctx.synchronize()
        tot += acc.double()
    ctx.set_integrator(0)
    img = (tot / spp).float().cpu().numpy().reshape(H, W, 3)
    rmse = float(np.sqrt(np.mean((img - g["image"]) ** 2)))
    out["image_cornell"] = {"rmse": rmse, "spp": spp, "golden_spp": int(g["spp"]), "pass": rmse <= 1e-3}

    # rates: the headline config and the triangle mesh (fewer samples; rate is per ray)
    for name, param, W, H, spp in (("cornell", 0, 1024, 1024, 1024), ("cornell_smoke", 0, 1024, 1024, 1024), ("tri_mesh", 1000000, 1920, 1080, 64)):
        sc = b.Scene(name, W, H, param)
        ctx.upload(sc)
        acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.render_accumulate(acc, W, H, 0, spp, 419, 100, sc.background)      # warm-up
        ctx.synchronize()
        ctx.reset_counters()
        for k in range(2):
            ctx.render_accumulate(acc, W, H, (k + 1) * spp, (k + 2) * spp, 419, 100, sc.background)
        c = ctx.counters()
        out["rate_" + name] = {"Grays_per_s": c.rays / (c.render_ms * 1e-3) / 1e9, "ms_per_call": c.render_ms / c.render_launches,
                               "rays_per_sample": c.rays / c.samples}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
