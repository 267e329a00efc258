"""A mechanical check of the two restatements the reference cannot pin.

`ConstantMedium::hit` (Volumes.cuh:26-49) and `Camera` (Camera.cuh:12-46) include <curand_kernel.h>, which this image lacks,
so `oracle/_ref` holds no build of them and no fixture of the reference covers them: rt_oracle.c restates them, checked by
reading.  This file restates the same reference lines a SECOND time, independently, in numpy float32, from the pieces that
ARE pinned -- the boundary's own hit() (rto_object_hit: bit-exact against libref_geom.so), the Philox block (Random123
known answers) -- and requires the C restatement to agree bit for bit.  Two readings of the same lines that agree catch a
typo in either; they pin nothing (PARITY UNPINNED stays, and the test says so).  CPU only."""
import ctypes as C
import ctypes.util

import numpy as np
import pytest

from conftest import FLT_MAX

f32 = np.float32
RNG_CAMERA, RNG_DISK, RNG_MEDIUM = 0, 1, 4
RT_OBJ_MEDIUM = 7
SEED = 0x1234567800000419

_libm = C.CDLL(ctypes.util.find_library("m"))
_libm.logf.restype = C.c_float
_libm.logf.argtypes = [C.c_float]
_libm.tanf.restype = C.c_float
_libm.tanf.argtypes = [C.c_float]


def uniforms(oracle, purpose, c0, bounce, pixel, sample, n):
    """words 0..n-1 of the block (c0, purpose<<24 | bounce, pixel, sample) as (0,1] uniforms (csrc/philox.h)"""
    w = oracle.philox_block([c0, (purpose << 24) | bounce, pixel, sample], [SEED & 0xffffffff, SEED >> 32])
    return [f32(2.3283064e-10) + f32(w[i]) * f32(2.3283064e-10) for i in range(n)]


def media_of(flat):
    f = flat.contents
    return [i for i in range(f.n_objects) if f.objects[i].type == RT_OBJ_MEDIUM]


def medium_hit_numpy(oracle, flat, mi, ray, tmin, tmax, psb):
    """Volumes.cuh:26-49, line by line; returns None or (t, p)"""
    o = flat.contents.objects[mi]
    assert o.op_count == 0        # the scenes wrap the boundary, not the medium
    q = ray[None, :]
    r1 = oracle.object_hit(flat, o.child_begin, q, -FLT_MAX, FLT_MAX)[0]                       # :29
    if not r1["hit"]:
        return None
    r2 = oracle.object_hit(flat, o.child_begin, q, float(f32(r1["t"]) + f32(.0001)), FLT_MAX)[0]    # :31
    if not r2["hit"]:
        return None
    t1, t2 = f32(r1["t"]), f32(r2["t"])
    if t1 < f32(tmin):
        t1 = f32(tmin)                                                                         # :33
    if t2 > f32(tmax):
        t2 = f32(tmax)                                                                         # :34
    if t1 >= t2:
        return None                                                                            # :35
    if t1 < 0:
        t1 = f32(0)                                                                            # :36
    d = ray[3:6].astype(f32)
    ray_length = np.sqrt(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))          # :38, Vector3.cuh:47
    inside = f32(f32(t2 - t1) * ray_length)                                                    # :39
    xi = uniforms(oracle, RNG_MEDIUM, o.medium_slot, int(psb[2]), int(psb[0]), int(psb[1]), 1)[0]
    hit_distance = f32(f32(o.f[0]) * f32(_libm.logf(float(xi))))                               # :40, f[0] = -1/density
    if hit_distance > inside:
        return None                                                                            # :42
    t = f32(t1 + f32(hit_distance / ray_length))                                               # :43
    p = ray[0:3].astype(f32) + t * d                                                           # :44
    return t, p.astype(f32)


@pytest.mark.parametrize("name,param", [("cornell_smoke", 0), ("final", 6), ("kat", 0)])
def test_constant_medium_hit_restated_twice(binding, oracle, name, param):
    sc = binding.Scene(name, 64, 64, param)
    flat = sc.flat
    media = media_of(flat)
    if not media:
        pytest.skip("no medium in " + name)
    rng = np.random.default_rng(7)
    f = flat.contents
    cam = np.array(f.camera.origin[:], dtype=f32)
    n = 600
    n_hit = 0
    for mi in media:
        o = f.objects[mi]
        b = f.objects[o.child_begin]
        lo, hi = np.array(b.bmin[:], dtype=f32), np.array(b.bmax[:], dtype=f32)
        ext = np.minimum(hi - lo, f32(2000))
        mid = np.where(np.isfinite(lo + hi), (lo + hi) / 2, 0).astype(f32)
        # rays from the camera, from inside the boundary and from beyond it, aimed at points around the boundary; ranges
        # that start inside, end inside, and the render loop's own (.001, FLT_MAX)
        org = np.where(rng.random((n, 1)) < 0.4, cam[None, :], mid[None, :] + (rng.random((n, 3)) - 0.5).astype(f32) * ext * f32(1.6))
        tgt = mid[None, :] + (rng.random((n, 3)) - 0.5).astype(f32) * ext * f32(1.1)
        d = (tgt - org).astype(f32)
        d *= rng.choice(np.array([1.0, 1.0, 0.01, 37.0], dtype=f32), (n, 1))      # direction length enters :38-:43
        rays = np.concatenate([org.astype(f32), d, rng.random((n, 1)).astype(f32)], axis=1).astype(f32)
        psb = np.stack([rng.integers(0, 1 << 20, n), rng.integers(0, 4096, n), rng.integers(0, 100, n)], axis=1).astype(np.uint32)
        for tmin, tmax in ((.001, FLT_MAX), (.001, 0.6), (0.3, 0.9), (-1.0, FLT_MAX)):
            got = oracle.object_hit_addressed(flat, mi, rays, tmin, tmax, SEED, psb)
            for i in range(n):
                want = medium_hit_numpy(oracle, flat, mi, rays[i], tmin, tmax, psb[i])
                assert bool(got[i]["hit"]) == (want is not None), (name, mi, i, tmin, tmax)
                if want is None:
                    continue
                n_hit += 1
                t, p = want
                assert got[i]["t"].tobytes() == t.tobytes(), (name, mi, i)
                assert got[i]["p"].tobytes() == p.tobytes()
                assert tuple(got[i]["n"]) == (1.0, 0.0, 0.0) and got[i]["front_face"] == 1     # :45-:46
                assert got[i]["material"] == o.material
    assert n_hit > 100       # the comparison is not vacuous


def cross(a, b):       # Vector3.cuh:110-115
    return np.array([f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(-(f32(a[0] * b[2]) - f32(a[2] * b[0]))), f32(a[0] * b[1]) - f32(a[1] * b[0])], dtype=f32)


def unit(v):           # Vector3.cuh:166: v / length (true division)
    ln = np.sqrt(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))
    return (v / ln).astype(f32)


def camera_ctor_numpy(lookfrom, lookat, vup, vfov, aspect, aperture, focus, t0, t1):
    """Camera.cuh:23-39"""
    lookfrom, lookat, vup = (np.array(x, dtype=f32) for x in (lookfrom, lookat, vup))
    lens_radius = f32(f32(aperture) / f32(2))
    theta = f32(f32(f32(vfov) * f32(np.pi)) / f32(180))
    half_h = f32(_libm.tanf(float(f32(theta / f32(2)))))
    half_w = f32(f32(aspect) * half_h)
    origin = lookfrom
    w = unit((lookfrom - lookat).astype(f32))
    u = unit(cross(vup, w))
    v = cross(w, u)
    fd = f32(focus)
    llc = (((origin - (half_w * u).astype(f32) * fd).astype(f32) - (half_h * v).astype(f32) * fd).astype(f32) - w * fd).astype(f32)
    hor = ((f32(f32(2) * half_w) * u).astype(f32) * fd).astype(f32)
    ver = ((f32(f32(2) * half_h) * v).astype(f32) * fd).astype(f32)
    return dict(lower_left=llc, horizontal=hor, vertical=ver, origin=origin, u=u, v=v, w=w, lens_radius=lens_radius, time0=f32(t0), time1=f32(t1))


# look() parameters of the built-in scenes (csrc/scenes.cpp; the Cornell ones are main.cu:281-293's)
LOOKS = {
    "cornell": ((278, 278, -800), (278, 278, 0), 40.0, 0.0, 10.0),
    "random_spheres": ((13, 2, 3), (0, 0, 0), 20.0, 0.0, 10.0),
    "lit_sphere": ((13, 2, 3), (0, 0, 0), 30.0, 0.1, 10.0),
    "tri_mesh": ((0, 260, -760), (0, 20, 0), 40.0, 0.0, 10.0),
}


@pytest.mark.parametrize("name", sorted(LOOKS))
def test_camera_constructor_restated_twice(binding, name):
    """the product's derive_camera (csrc/lower.cpp) against a numpy float32 reading of Camera.cuh:23-39.  tan() is libm's
    tanf on both sides here; on another libm the members that depend on it may move by an ulp, hence 2 ulp, not bits."""
    W, H = 200, 100
    sc = binding.Scene(name, W, H, 100 if name == "tri_mesh" else 0)
    cam = sc.flat.contents.camera
    frm, at, vfov, aperture, focus = LOOKS[name]
    want = camera_ctor_numpy(frm, at, (0, 1, 0), vfov, f32(W) / f32(H), aperture, focus, 0.0, 1.0)
    for key, val in want.items():
        got = np.array(getattr(cam, key)[:] if hasattr(getattr(cam, key), "__len__") else [getattr(cam, key)], dtype=f32)
        val = np.atleast_1d(val).astype(f32)
        assert np.all(np.abs(got - val) <= 2 * np.spacing(np.maximum(np.abs(val), f32(1e-30)))), (key, got, val)


def generate_ray_numpy(oracle, cam, s, t, pixel, sample):
    """Camera.cuh:12-18 (randomInUnitDisk) and :40-46 (generateRay), with the addressed draws of csrc/philox.h: the jitter
    block's word 2 is the shutter draw, disk candidate k is block k of RNG_DISK"""
    k = 0
    while True:
        x = uniforms(oracle, RNG_DISK, k, 0, pixel, sample, 2)
        p = np.array([f32(f32(2) * x[0]) - f32(1), f32(f32(2) * x[1]) - f32(1), f32(0) - f32(0)], dtype=f32)
        k += 1
        if not (f32(f32(p[0] * p[0]) + f32(p[1] * p[1])) + f32(p[2] * p[2]) > f32(1)):
            break
    rd = (f32(cam.lens_radius) * p).astype(f32)
    u, v = np.array(cam.u[:], dtype=f32), np.array(cam.v[:], dtype=f32)
    offset = ((u * rd[0]).astype(f32) + (v * rd[1]).astype(f32)).astype(f32)
    xi = uniforms(oracle, RNG_CAMERA, 0, 0, pixel, sample, 3)[2]
    time = f32(f32(cam.time0) + f32(xi * f32(f32(cam.time1) - f32(cam.time0))))
    origin = np.array(cam.origin[:], dtype=f32)
    llc, hor, ver = (np.array(getattr(cam, n_)[:], dtype=f32) for n_ in ("lower_left", "horizontal", "vertical"))
    o = (origin + offset).astype(f32)
    d = ((((llc + (f32(s) * hor).astype(f32)).astype(f32) + (f32(t) * ver).astype(f32)).astype(f32) - origin).astype(f32) - offset).astype(f32)
    return o, d, time


@pytest.mark.parametrize("name", ["cornell", "lit_sphere", "final"])
def test_generate_ray_restated_twice(binding, oracle, name):
    sc = binding.Scene(name, 64, 48, 6 if name == "final" else 0)
    cam = sc.flat.contents.camera
    rng = np.random.default_rng(11)
    n = 400
    st = rng.random((n, 2)).astype(f32)
    ps = np.stack([rng.integers(0, 64 * 48, n), rng.integers(0, 1 << 16, n)], axis=1).astype(np.uint32)
    got = oracle.camera_rays(sc.flat, st, SEED, ps)
    for i in range(n):
        o, d, time = generate_ray_numpy(oracle, cam, st[i, 0], st[i, 1], int(ps[i, 0]), int(ps[i, 1]))
        assert got[i, 0:3].tobytes() == o.tobytes(), (name, i)
        assert got[i, 3:6].tobytes() == d.tobytes(), (name, i)
        assert got[i, 6].tobytes() == time.tobytes()
