#!/usr/bin/env python3
"""gen_golden.py — TEST INFRASTRUCTURE.  Writes the fixtures under tests/golden/.

Run in the container that has /root/reference (after `make -C oracle` has produced
oracle/_ref/libref_geom.so and libref_shade.so).  Every expected value in the *_ref.npz files comes out of
the REFERENCE's own code, host-compiled: hit() / value() / RotateY's constructor by oracle/ref_geom.cpp
(no stand-in of any kind: these are the reference pin), scatter() / Metal's fuzz clamp by oracle/ref_shade.cpp
(Material.cuh with curandState bound to rocRAND: a stand-in, so those vectors only cross-check the
restatement).  Each fixture records which library made which arrays (`produced_by`, `*_produced_by`) and the
tests assert it.  Inputs are seeded numpy draws.  The converged images (images_*.npz) come from the CPU restatement oracle/rt_oracle.c,
because the reference's integrator (main.cu) cannot be built here.

Each fixture stores a digest of the scene tables it was made from; tests refuse a stale fixture.
  python oracle/gen_golden.py [--images] [--full]   (--images adds the slow converged renders, --full the
                                                     primary-hit maps of the full-size BASELINE scenes)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FLT_MAX = float(np.finfo(np.float32).max)


def rays_at_box(rng, bmin, bmax, n, time_lo=0.0, time_hi=1.0):
    """rays from around the box towards points in (a slightly inflated copy of) it, un-normalised"""
    bmin, bmax = np.asarray(bmin, np.float64), np.asarray(bmax, np.float64)
    ext = np.maximum(bmax - bmin, 1e-3)
    ctr = 0.5 * (bmin + bmax)
    rad = 0.5 * np.linalg.norm(ext)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    org = ctr + d * rad * rng.uniform(0.2, 4.0, size=(n, 1))     # some origins are inside
    tgt = ctr + (rng.random((n, 3)) - 0.5) * ext * 1.3
    dirs = (tgt - org) * rng.uniform(0.3, 3.0, size=(n, 1))
    t = rng.uniform(time_lo, time_hi, size=(n, 1))
    return np.concatenate([org, dirs, t], axis=1).astype(np.float32)


def camera_rays(rng, cam, n):
    org = np.array(cam.origin[:], np.float32)
    st = rng.random((n, 2)).astype(np.float32)
    d = (np.array(cam.lower_left[:], np.float32) + st[:, :1] * np.array(cam.horizontal[:], np.float32)
         + st[:, 1:] * np.array(cam.vertical[:], np.float32) - org)
    return np.concatenate([np.tile(org, (n, 1)), d, rng.random((n, 1)).astype(np.float32)], axis=1).astype(np.float32)


def clean_front_face(flat, hits):
    """Sphere / MovingSphere / Triangle::hit never write rec.bFrontFace (Sphere.cuh:25-56, Triangle.cuh:21-46) and
    HitableList::hit hands back its uninitialised tempRec's byte (HitableList.cuh:19-28): stack garbage that changes from
    build to build.  Only rects and boxes define it (setFaceNormal, Hitable.cuh:17-21); store 0 elsewhere."""
    f = flat.contents
    defined = np.zeros(f.n_prims + 1, bool)
    for i in range(f.n_objects):
        o = f.objects[i]
        if o.prim_id >= 0 and o.type in (2, 3, 4, 5):
            defined[o.prim_id] = True
    ok = (hits["hit"] == 1) & defined[np.clip(hits["prim"], 0, f.n_prims)]
    hits["front_face"] = np.where(ok, hits["front_face"], 0)
    return hits


GEOM_SCENES = [("kat", 0), ("cornell", 0), ("lit_sphere", 0), ("bvh_boxes", 0), ("random_spheres", 0),
               ("cornell_cluster", 0), ("final", 6), ("tri_mesh", 800), ("final_custom", 6)]
W, H = 128, 72


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", nargs="*", default=None, metavar="SCENE",
                    help="also the slow converged renders (all of them, or only the scenes named)")
    ap.add_argument("--images-only", action="store_true", help="leave the vector fixtures as they are")
    ap.add_argument("--full", action="store_true", help="also the 128x72 primary-hit maps of tri_mesh at 1,000,000 triangles and "
                    "final at 100 x 100 boxes, from the reference's flat HitableList (minutes)")
    ap.add_argument("--full-only", action="store_true", help="only those")
    ap.add_argument("--special-only", action="store_true", help="only geom_special_ref.npz: rays with zero / negative-zero / huge / tiny "
                    "direction components, origins on surfaces, rays through box corners and triangle vertices")
    args = ap.parse_args()
    b = entry.load_binding()
    o = entry.load_oracle()
    global scene_digest
    scene_digest = o.scene_digest
    ref, orc = o.Reference(), o.Oracle()
    os.makedirs(GOLD, exist_ok=True)
    if args.images_only:
        make_images(args, b, orc)
        return
    if args.full_only:
        make_full(b, ref, orc)
        return
    if args.special_only:
        make_special(b, ref, orc)
        return
    rng = np.random.default_rng(20261110)

    # ---- 1. per-object and whole-world closest hits -------------------------------------------
    for name, param in GEOM_SCENES:
        sc = b.Scene(name, W, H, param)
        f = sc.flat.contents
        world = f.objects[f.world]
        out = {"digest": scene_digest(sc.flat), "param": param, "width": W, "height": H, "produced_by": ref.geom_id}
        # individual top-level objects (at most 16 of them), wrappers included
        kids = [world.child_begin + i for i in range(world.child_count)]
        picks = kids if len(kids) <= 16 else [kids[i] for i in sorted(rng.choice(len(kids), 16, replace=False))]
        obj_ix, obj_rays, obj_hits = [], [], []
        for oi in picks:
            ob = f.objects[oi]
            if ob.type == 7:      # media are not buildable from the reference here
                continue
            r = rays_at_box(rng, ob.bmin[:], ob.bmax[:], 384)
            hits = clean_front_face(sc.flat, ref.object_hit(sc.flat, oi, r, 0.001, FLT_MAX))
            obj_ix.append(oi); obj_rays.append(r); obj_hits.append(hits)
        out["obj_index"] = np.array(obj_ix, np.int32)
        out["obj_rays"] = np.stack(obj_rays)
        out["obj_hits"] = np.stack(obj_hits)
        # a bounded (tmin,tmax) window on the same rays of the first object
        out["win_rays"] = obj_rays[0]
        tm = np.median(obj_hits[0]["t"][obj_hits[0]["hit"] == 1]) if (obj_hits[0]["hit"] == 1).any() else 1.0
        out["win_range"] = np.array([0.25 * tm, 1.0 * tm], np.float32)
        out["win_hits"] = clean_front_face(sc.flat, ref.object_hit(sc.flat, picks[0], obj_rays[0], float(out["win_range"][0]), float(out["win_range"][1])))
        # whole world: camera rays + bounce-like rays leaving hit points
        cr = camera_rays(rng, f.camera, 3000)
        h0 = ref.object_hit(sc.flat, f.world, cr, 0.001, FLT_MAX)
        br = cr.copy()
        m = h0["hit"] == 1
        br[m, 0:3] = h0["p"][m]
        br[:, 3:6] = rng.normal(size=(len(br), 3)).astype(np.float32)
        wr = np.concatenate([cr, br]).astype(np.float32)
        out["world_rays"] = wr
        out["world_hits"] = clean_front_face(sc.flat, ref.object_hit(sc.flat, f.world, wr, 0.001, FLT_MAX))
        # primary-hit map (pinhole-centre rays, the same ones rt_trace_primary builds)
        po, to = orc.trace_primary(sc.flat, W, H)           # restatement, to recover the rays' results
        cam = f.camera
        jj, ii = np.mgrid[0:H, 0:W]
        u = ((ii + np.float32(0.5)).astype(np.float32) / np.float32(W)).astype(np.float32)
        v = ((jj + np.float32(0.5)).astype(np.float32) / np.float32(H)).astype(np.float32)
        ll, hz, vt, og = (np.array(x[:], np.float32) for x in (cam.lower_left, cam.horizontal, cam.vertical, cam.origin))
        d = ((ll + u[..., None] * hz).astype(np.float32) + v[..., None] * vt).astype(np.float32) - og
        pr = np.concatenate([np.broadcast_to(og, d.shape), d.astype(np.float32), np.full(d.shape[:2] + (1,), cam.time0, np.float32)], axis=2)
        ph = ref.object_hit(sc.flat, f.world, pr.reshape(-1, 7), 0.001, FLT_MAX)
        out["primary_prim"] = ph["prim"].reshape(H, W)
        out["primary_t"] = np.where(ph["hit"] == 1, ph["t"], 0).astype(np.float32).reshape(H, W)
        agree = np.array_equal(out["primary_prim"], po) and np.array_equal(out["primary_t"], to)
        print("%-16s objects %4d  world hits %5d  primary map restatement==reference: %s" % (name, len(obj_ix), int(m.sum()), agree))
        np.savez_compressed(os.path.join(GOLD, "geom_%s_ref.npz" % name), **out)

    # ---- 2. materials and textures (scene "kat" carries one of each, "random_spheres" the glass) ----
    sc = b.Scene("kat", W, H)
    f = sc.flat.contents
    out = {"digest": scene_digest(sc.flat), "mat_produced_by": ref.shade_id, "tex_produced_by": ref.geom_id}
    n, XI = 512, 96
    mats, rays_in, recs, ps, res = [], [], [], [], []
    for mi in range(f.n_materials):
        d = rng.normal(size=(n, 3)) * rng.uniform(0.2, 5.0, size=(n, 1))
        nrm = rng.normal(size=(n, 3))
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        # half the normals face the ray (the usual case), half do not (inside glass)
        flip = (np.einsum("ij,ij->i", d, nrm) > 0) & (rng.random(n) < 0.5)
        nrm[flip] *= -1
        p = rng.uniform(-6, 6, size=(n, 3))
        ray = np.concatenate([p - d, d, rng.random((n, 1))], axis=1).astype(np.float32)
        rec = np.concatenate([np.ones((n, 1)), rng.random((n, 2)), p, nrm], axis=1).astype(np.float32)
        pxs = np.stack([rng.integers(0, 1 << 20, n), rng.integers(0, 4096, n)], axis=1).astype(np.uint32)
        mats.append(mi); rays_in.append(ray); recs.append(rec); ps.append(pxs)
        r_out = ref.scatter(sc.flat, mi, ray, rec, 419, pxs)
        res.append(r_out)
        # The uniforms the reference's scatter() drew are the first `draws` values of the rocRAND Philox
        # stream (seed 419, subsequence pixel | sample<<32).  Tests rebuild that stream (misc_ref.npz pins
        # it) and feed it to the restatement as a script of XI values.
        assert r_out["draws"].max() < XI - 2, "rejection loop ran past the script length"
    out.update(mat_index=np.array(mats, np.int32), mat_rays=np.stack(rays_in), mat_recs=np.stack(recs),
               mat_pixel_sample=np.stack(ps), mat_out=np.stack(res), seed=419, xi_len=XI)
    texs, uvps, rgbs = [], [], []
    for ti in range(f.n_textures):
        uvp = np.concatenate([rng.uniform(-0.1, 1.1, size=(n, 2)), rng.uniform(-6, 6, size=(n, 3))], axis=1).astype(np.float32)
        uvp[:8, 0] = [0, 1, 0.5, 1, 0, 0.999999, 1e-7, 0.25]      # image clamp edges
        uvp[:8, 1] = [0, 1, 1, 0, 0.5, 1e-7, 0.999999, 0.75]
        texs.append(ti); uvps.append(uvp); rgbs.append(ref.texture_value(sc.flat, ti, uvp))
    out.update(tex_index=np.array(texs, np.int32), tex_uvp=np.stack(uvps), tex_rgb=np.stack(rgbs))
    np.savez_compressed(os.path.join(GOLD, "shade_kat_ref.npz"), **out)
    print("materials %d, textures %d" % (len(mats), len(texs)))

    # ---- 3. RNG stream, RotateY constructor, Metal fuzz clamp -----------------------------------
    keys = [(419, 0, 0), (419, 12345, 7), (0xDEADBEEFCAFE, 1048575, 4095), (1, 1, 1)]
    uni = np.stack([ref.uniforms(s, p, q, 64) for s, p, q in keys])
    angles = np.array([-18, 15, 0, 120, 33, -50, 90, 180, 359.5, -0.25], np.float32)
    sc_ = np.array([ref.rotate_sincos(float(a)) for a in angles], np.float32)
    fz = np.array([0.0, 0.3, 0.999, 1.0, 1.5, 10.0], np.float32)
    np.savez_compressed(os.path.join(GOLD, "misc_ref.npz"), rng_keys=np.array(keys, np.uint64), rng_uniforms=uni,
                        rng_produced_by="rocRAND Philox4x32-10 of the image, through " + ref.shade_id,
                        sincos_produced_by=ref.geom_id, fuzz_produced_by=ref.shade_id,
                        angles=angles, sincos=sc_, fuzz_in=fz, fuzz_out=np.array([ref.metal_fuzz(float(x)) for x in fz], np.float32))
    print("misc ok")

    if args.full:
        make_full(b, ref, orc)
    make_special(b, ref, orc)
    make_images(args, b, orc)


def primary_rays(cam, w, h):
    """the pinhole-centre ray of every pixel, built in fp32 exactly as rt_trace_primary / rto_trace_primary build it"""
    jj, ii = np.mgrid[0:h, 0:w]
    u = ((ii + np.float32(0.5)).astype(np.float32) / np.float32(w)).astype(np.float32)
    v = ((jj + np.float32(0.5)).astype(np.float32) / np.float32(h)).astype(np.float32)
    ll, hz, vt, og = (np.array(x[:], np.float32) for x in (cam.lower_left, cam.horizontal, cam.vertical, cam.origin))
    d = ((ll + u[..., None] * hz).astype(np.float32) + v[..., None] * vt).astype(np.float32) - og
    return np.concatenate([np.broadcast_to(og, d.shape), d.astype(np.float32), np.full(d.shape[:2] + (1,), cam.time0, np.float32)],
                          axis=2).reshape(-1, 7)


SPECIAL_SCENES = [("kat", 0), ("cornell", 0), ("bvh_boxes", 0), ("tri_mesh", 800), ("cornell_cluster", 0)]


SPECIAL_KINDS = ("corner", "axis_on_surface", "axis_offset", "in_plane", "scaled", "far")


def special_rays(flat, rng):
    """Rays the random draws of section 1 never produce: where a division's divisor is +-0, a product is 0 * inf, a hit lies
    exactly on an edge, a corner or the tMin / tMax bound, or the operands sit at the ends of the float range.  Returns
    the rays and their kind (index into SPECIAL_KINDS):
      corner           from the camera exactly at box corners / centres (edges, vertices, shared faces)
      axis_on_surface  along the axes and the face diagonals (zero and negative-zero components) from those same points:
                       the origin lies ON bounding planes, so 0 / 0 and 0 * inf occur
      axis_offset      the same directions from origins moved off those planes by a random amount: divisors of +-0
                       without 0 / 0
      in_plane         along z from the camera's side, inside face planes
      scaled           `corner` with the direction scaled by 1e18, 1e-18, 1e-30 (same line, t scales)
      far              `corner` with the origin moved 1e5 direction lengths back (coordinates of 1e7 and more: what is left of
                       float precision there decides, and culling by boxes decides differently from the primitives)"""
    f = flat.contents
    cam = f.camera
    og = np.array(cam.origin[:], np.float32)
    corners = []
    for i in range(f.n_objects):
        o = f.objects[i]
        if o.prim_id < 0:
            continue
        lo, hi = np.array(o.bmin[:], np.float32), np.array(o.bmax[:], np.float32)
        if not (np.isfinite(lo).all() and np.isfinite(hi).all()):
            continue
        for m in range(8):
            corners.append(np.where([(m >> k) & 1 for k in range(3)], hi, lo))
        corners.append((0.5 * (lo.astype(np.float64) + hi)).astype(np.float32))
    corners = np.unique(np.array(corners, np.float32), axis=0)
    if len(corners) > 160:
        corners = corners[np.sort(rng.choice(len(corners), 160, replace=False))]
    out, kinds = [], []

    def add(rays, kind):
        out.append(np.asarray(rays, np.float32))
        kinds.append(np.full(len(rays), SPECIAL_KINDS.index(kind), np.int32))

    a = np.concatenate([np.tile(og, (len(corners), 1)), corners - og], axis=1)
    add(a, "corner")
    dirs = []
    for ax in range(3):
        for sgn in (1.0, -1.0):
            e = np.zeros(3, np.float32); e[ax] = sgn
            dirs.append(e)
            e2 = e.copy(); e2[(ax + 1) % 3] = -0.0      # a negative zero beside it
            dirs.append(e2)
    for ax in range(3):
        for s1 in (1.0, -1.0):
            for s2 in (1.0, -1.0):
                e = np.zeros(3, np.float32); e[ax] = s1; e[(ax + 1) % 3] = s2
                dirs.append(e)
    dirs = np.array(dirs, np.float32)
    some = corners[:: max(1, len(corners) // 24)]
    add(np.concatenate([np.repeat(some, len(dirs), axis=0), np.tile(dirs, (len(some), 1))], axis=1), "axis_on_surface")
    ext = float(np.max(corners.max(axis=0) - corners.min(axis=0)))
    moved = (np.repeat(some, len(dirs), axis=0) + rng.uniform(-0.37, 0.37, size=(len(some) * len(dirs), 3)) * 0.05 * ext).astype(np.float32)
    add(np.concatenate([moved, np.tile(dirs, (len(some), 1))], axis=1), "axis_offset")
    far = some.copy(); far[:, 2] = og[2]
    zdir = np.tile(np.array([0.0, 0.0, 1.0], np.float32) * np.sign(-og[2] if og[2] != 0 else 1.0), (len(far), 1))
    add(np.concatenate([far, zdir], axis=1), "in_plane")
    a = a[: min(64, len(a))]
    for scale in (1e18, 1e-18, 1e-30):
        b = a.copy(); b[:, 3:6] = (b[:, 3:6].astype(np.float64) * scale).astype(np.float32)
        add(b, "scaled")
    b = a.copy(); b[:, 0:3] = (a[:, 0:3].astype(np.float64) - 1e5 * a[:, 3:6]).astype(np.float32)
    add(b, "far")
    rays = np.concatenate(out).astype(np.float32)
    return np.concatenate([rays, np.full((len(rays), 1), cam.time0, np.float32)], axis=1), np.concatenate(kinds)


def make_special(b, ref, orc):
    # ---- 6. special rays (see special_rays) through the reference's flat list, whole world, two (tMin, tMax) windows ----
    rng = np.random.default_rng(20261111)
    out = {"produced_by": ref.geom_id, "scenes": np.array([n for n, _ in SPECIAL_SCENES]), "kinds": np.array(SPECIAL_KINDS)}
    for name, param in SPECIAL_SCENES:
        sc = b.Scene(name, W, H, param)
        f = sc.flat.contents
        rays, kind = special_rays(sc.flat, rng)
        hits = clean_front_face(sc.flat, ref.object_hit(sc.flat, f.world, rays, 0.001, FLT_MAX))
        got = orc.trace_rays(sc.flat, rays, 0.001, FLT_MAX)
        h = hits["hit"] == 1
        same = np.array_equal(got["hit"], hits["hit"]) and np.array_equal(got["t"][h], hits["t"][h], equal_nan=True)
        # Rects report a hit with t = NaN when the ray lies in their plane (0 / 0 passes every `<` and `>` test of
        # Rectangle.cuh:27-33): a quirk of the reference that the flat-list restatement reproduces NaN for NaN
        print("%-16s special rays %5d  hits %5d  of them with t = NaN %d  restatement==reference (hit, t): %s"
              % (name, len(rays), int(h.sum()), int(np.isnan(hits["t"][h]).sum()), same))
        out["digest_" + name] = scene_digest(sc.flat)
        out["param_" + name] = param
        out["rays_" + name] = rays
        out["kind_" + name] = kind
        out["hits_" + name] = hits
    np.savez_compressed(os.path.join(GOLD, "geom_special_ref.npz"), **out)


def make_full(b, ref, orc):
    # ---- 5. the BASELINE target scenes at their real size: the reference's flat HitableList over every primitive ----
    import time
    for name, param in (("tri_mesh", 1000000), ("final", 100)):
        sc = b.Scene(name, W, H, param)
        f = sc.flat.contents
        t0 = time.time()
        ph = ref.object_hit(sc.flat, f.world, primary_rays(f.camera, W, H), 0.001, FLT_MAX, threads=os.cpu_count() or 1)
        prim = ph["prim"].reshape(H, W)
        t = np.where(ph["hit"] == 1, ph["t"], 0).astype(np.float32).reshape(H, W)
        po, to = orc.trace_primary(sc.flat, W, H, use_bvh=True)          # the restatement's BVH traversal
        differ = prim != po
        print("%-10s param %7d: %d objects, reference flat list %.0f s; restatement (BVH) ids differ at %d pixels, t equal there: %s, t bit-equal: %s"
              % (name, param, f.n_objects, time.time() - t0, int(differ.sum()), np.array_equal(t[differ], to[differ]), np.array_equal(t, to)))
        np.savez_compressed(os.path.join(GOLD, "geom_%s_full_ref.npz" % name), digest=scene_digest(sc.flat), param=param, width=W, height=H,
                            produced_by=ref.geom_id, primary_prim=prim, primary_t=t)


# converged goldens: 64 x 64, samples per pixel chosen from the measured per-pixel variance of each scene (mean over
# pixels and channels, sigma^2: cornell 0.021, cornell_smoke 0.023, random_spheres 0.038, lit_sphere 0.196, final(8) 0.147)
# so that the golden's own noise sigma / sqrt(spp) stays at or below 6e-4; the GPU side of the test renders 4 x as many
# samples (3e-4), which leaves the sum of both well inside north_star's RMSE 1e-3.
IMAGE_SCENES = [("cornell", 131072, False), ("cornell_smoke", 131072, False), ("random_spheres", 131072, True),
                ("lit_sphere", 524288, True), ("final", 524288, True)]
IMAGE_CHUNK = 8192      # samples rendered into one fp32 buffer; chunks are added in float64 (a 500k-term fp32 sum drops small terms)


def make_images(args, b, orc):
    # ---- 4. converged images from the restatement (slow) ------------------------------------------
    if args.images is not None:
        import time
        IW = IH = 64
        threads = int(os.environ.get("RT_GOLDEN_THREADS", os.cpu_count() or 1))
        for name, spp, bvh in IMAGE_SCENES:
            if args.images and name not in args.images:
                continue
            sc = b.Scene(name, IW, IH, 8 if name == "final" else 0)      # final: the mini scene, 8 x 8 boxes
            total = np.zeros((IH, IW, 3), np.float64)
            rays, t0 = 0, time.time()
            for s0 in range(0, spp, IMAGE_CHUNK):
                acc, st = orc.render(sc.flat, IW, IH, s0, min(spp, s0 + IMAGE_CHUNK), 1000003, 100, sc.background, use_bvh=bvh, threads=threads)
                total += acc
                rays += st.rays
            img = (total / spp).astype(np.float32)
            np.savez_compressed(os.path.join(GOLD, "image_%s_oracle.npz" % name), digest=scene_digest(sc.flat), image=img,
                                spp=spp, seed=1000003, width=IW, height=IH, rays=rays)
            print("image %-16s %d spp, %d rays, mean %.5f, %.0f s" % (name, spp, rays, img.mean(), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
