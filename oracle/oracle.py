"""ctypes loader for the CPU oracle (liboracle.so) and, where it was built, the host build of
the reference's own headers (_ref/libref_geom.so: geometry + textures, no stand-in; _ref/libref_shade.so:
Material.cuh with its RNG names bound to rocRAND, a labelled stand-in).  TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import hashlib
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_PATH = os.path.join(_HERE, "liboracle.so")
REF_GEOM_PATH = os.path.join(_HERE, "_ref", "libref_geom.so")
REF_SHADE_PATH = os.path.join(_HERE, "_ref", "libref_shade.so")

HIT_DTYPE = np.dtype([("hit", "<i4"), ("prim", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4"),
                      ("p", "<f4", (3,)), ("n", "<f4", (3,)), ("front_face", "<i4"), ("material", "<i4")])
SCATTER_DTYPE = np.dtype([("ok", "<i4"), ("attenuation", "<f4", (3,)), ("origin", "<f4", (3,)),
                          ("direction", "<f4", (3,)), ("time", "<f4"), ("emitted", "<f4", (3,)), ("draws", "<u4")])


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64), ("prim_tests", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("philox_blocks", C.c_uint64), ("sphere_iters", C.c_uint64),
                ("rect_tests", C.c_uint64), ("sphere_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("rot_ops", C.c_uint64), ("trans_ops", C.c_uint64)]


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class _Common:
    """entry points with the same shape in both libraries (prefix rto_ / ref_)"""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def object_hit(self, flat, obj, rays, tmin, tmax):
        rays = _f32(rays)
        out = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        fn = self._fn("object_hit")
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        fn.restype = None
        fn(C.cast(flat, C.c_void_p), obj, rays.shape[0], _ptr(rays), tmin, tmax, _ptr(out))
        return out

    def texture_value(self, flat, tex, uvp):
        uvp = _f32(uvp)
        out = np.zeros((uvp.shape[0], 3), dtype=np.float32)
        fn = self._fn("texture_value")
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        fn.restype = None
        fn(C.cast(flat, C.c_void_p), tex, uvp.shape[0], _ptr(uvp), _ptr(out))
        return out

    def uniforms(self, seed, pixel, sample, n):
        out = np.zeros(n, dtype=np.float32)
        fn = self._fn("uniforms")
        fn.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        fn.restype = None
        fn(seed, pixel, sample, n, _ptr(out))
        return out


class Oracle(_Common):
    def __init__(self, path=ORACLE_PATH):
        if not os.path.exists(path):
            raise RuntimeError("%s missing: run `make -C oracle`" % path)
        super().__init__(C.CDLL(path), "rto_")

    def object_hit_addressed(self, flat, obj, rays, tmin, tmax, seed, pixel_sample_bounce):
        """one object's hit() with the addressed draws of (seed, pixel, sample, bounce): media take their draw"""
        rays = _f32(rays)
        psb = np.ascontiguousarray(pixel_sample_bounce, dtype=np.uint32)
        out = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        fn = self.lib.rto_object_hit_addressed
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p]
        fn.restype = None
        fn(C.cast(flat, C.c_void_p), obj, rays.shape[0], _ptr(rays), tmin, tmax, seed, _ptr(psb), _ptr(out))
        return out

    def scatter(self, flat, mat, ray_in, rec_in, xi):
        """xi[n, stride]: scripted uniforms, consumed per call in the order the reference draws them"""
        ray_in, rec_in, xi = _f32(ray_in), _f32(rec_in), _f32(xi)
        out = np.zeros(ray_in.shape[0], dtype=SCATTER_DTYPE)
        fn = self.lib.rto_scatter
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        fn.restype = None
        fn(C.cast(flat, C.c_void_p), mat, ray_in.shape[0], _ptr(ray_in), _ptr(rec_in), _ptr(xi), xi.shape[1], _ptr(out))
        return out

    def scatter_addressed(self, flat, materials, ray_in, rec_in, seed, pixel_sample_bounce):
        """scatter()/emitted() with the render loop's own addressed Philox draws for (seed, pixel, sample, bounce)"""
        ray_in, rec_in = _f32(ray_in), _f32(rec_in)
        mats = np.ascontiguousarray(materials, dtype=np.int32)
        psb = np.ascontiguousarray(pixel_sample_bounce, dtype=np.uint32)
        out = np.zeros(ray_in.shape[0], dtype=SCATTER_DTYPE)
        fn = self.lib.rto_scatter_addressed
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        fn.restype = None
        fn(C.cast(flat, C.c_void_p), ray_in.shape[0], _ptr(ray_in), _ptr(rec_in), _ptr(mats), seed, _ptr(psb), _ptr(out))
        return out

    def trace_rays(self, flat, rays, tmin, tmax, use_bvh=False, stats=None):
        rays = _f32(rays)
        out = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        fn = self.lib.rto_trace_rays
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        fn.restype = None
        fn(C.cast(flat, C.c_void_p), rays.shape[0], _ptr(rays), tmin, tmax, int(use_bvh), _ptr(out),
           C.cast(C.byref(stats), C.c_void_p) if stats is not None else None)
        return out

    def trace_primary(self, flat, W, H, s_off=0.5, t_off=0.5, use_bvh=False):
        prim = np.zeros((H, W), dtype=np.int32)
        t = np.zeros((H, W), dtype=np.float32)
        fn = self.lib.rto_trace_primary
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        fn.restype = None
        fn(C.cast(flat, C.c_void_p), W, H, s_off, t_off, int(use_bvh), _ptr(prim), _ptr(t))
        return prim, t

    def camera_rays(self, flat, st, seed, pixel_sample):
        st = _f32(st)
        ps = np.ascontiguousarray(pixel_sample, dtype=np.uint32)
        out = np.zeros((st.shape[0], 7), dtype=np.float32)
        fn = self.lib.rto_camera_rays
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        fn.restype = None
        fn(C.cast(flat, C.c_void_p), st.shape[0], _ptr(st), seed, _ptr(ps), _ptr(out))
        return out

    def philox_block(self, ctr, key):
        c = np.ascontiguousarray(ctr, dtype=np.uint32)
        k = np.ascontiguousarray(key, dtype=np.uint32)
        o = np.zeros(4, dtype=np.uint32)
        fn = self.lib.rto_philox_block
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        fn.restype = None
        fn(_ptr(c), _ptr(k), _ptr(o))
        return o

    def render(self, flat, W, H, s0, s1, seed, max_depth=100, background=0, use_bvh=False, threads=1, accum=None):
        if accum is None:
            accum = np.zeros((H, W, 3), dtype=np.float32)
        st = Stats()
        fn = self.lib.rto_render
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                       C.c_void_p, C.c_void_p]
        fn.restype = None
        fn(C.cast(flat, C.c_void_p), W, H, s0, s1, seed, max_depth, background, int(use_bvh), threads, _ptr(accum),
           C.cast(C.byref(st), C.c_void_p))
        return accum, st

    def resolve(self, accum, spp, gamma=True):
        accum = _f32(accum)
        out = np.zeros_like(accum)
        fn = self.lib.rto_resolve
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        fn.restype = None
        fn(_ptr(accum), accum.size // 3, spp, int(gamma), _ptr(out))
        return out


class Reference(_Common):
    """The reference's own headers, host-compiled (only where /root/reference was present at build).
    object_hit / texture_value / rotate_sincos come from libref_geom.so, which contains no stand-in;
    scatter / uniforms / metal_fuzz come from libref_shade.so, whose curandState is rocRAND's (a stand-in)."""

    def __init__(self, geom_path=REF_GEOM_PATH, shade_path=REF_SHADE_PATH):
        for path in (geom_path, shade_path):
            if not os.path.exists(path):
                raise RuntimeError("%s missing (needs /root/reference): run `make -C oracle`" % path)
        super().__init__(C.CDLL(geom_path), "ref_")
        self.shade = C.CDLL(shade_path)
        for lib in (self.lib, self.shade):
            lib.ref_library_id.restype = C.c_char_p
        self.geom_id = self.lib.ref_library_id().decode()
        self.shade_id = self.shade.ref_library_id().decode()

    def object_hit(self, flat, obj, rays, tmin, tmax, threads=1):
        rays = _f32(rays)
        out = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        fn = self.lib.ref_object_hit_mt
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_int]
        fn.restype = None
        fn(C.cast(flat, C.c_void_p), obj, rays.shape[0], _ptr(rays), tmin, tmax, _ptr(out), threads)
        return out

    def uniforms(self, seed, pixel, sample, n):
        out = np.zeros(n, dtype=np.float32)
        fn = self.shade.ref_uniforms
        fn.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        fn.restype = None
        fn(seed, pixel, sample, n, _ptr(out))
        return out

    def scatter(self, flat, mat, ray_in, rec_in, seed, pixel_sample):
        ray_in, rec_in = _f32(ray_in), _f32(rec_in)
        ps = np.ascontiguousarray(pixel_sample, dtype=np.uint32)
        out = np.zeros(ray_in.shape[0], dtype=SCATTER_DTYPE)
        fn = self.shade.ref_scatter
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        fn.restype = None
        fn(C.cast(flat, C.c_void_p), mat, ray_in.shape[0], _ptr(ray_in), _ptr(rec_in), seed, _ptr(ps), _ptr(out))
        return out

    def rotate_sincos(self, degrees):
        s, c = C.c_float(), C.c_float()
        fn = self.lib.ref_rotate_sincos
        fn.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        fn.restype = None
        fn(degrees, C.byref(s), C.byref(c))
        return s.value, c.value

    def metal_fuzz(self, f):
        fn = self.shade.ref_metal_fuzz
        fn.argtypes = [C.c_float]
        fn.restype = C.c_float
        return fn(f)


def scene_digest(flat):
    """sha1 over the raw bytes of every table of an rt_flat_scene"""
    f = flat.contents
    h = hashlib.sha1()
    for ptr, n, size in [(f.objects, f.n_objects, 104), (f.ops, f.n_ops, 16), (f.materials, f.n_materials, 16),
                         (f.textures, f.n_textures, 40), (f.image_bytes, f.n_image_bytes, 1)]:
        if n:
            h.update(C.string_at(C.cast(ptr, C.c_void_p), int(n) * size))
    h.update(bytes(f.camera))
    return h.hexdigest()


def reference_available():
    return os.path.exists(REF_GEOM_PATH) and os.path.exists(REF_SHADE_PATH)
