"""ctypes binding of librt_amd.so (include/rt_api.h) for tests and bench.py.

This is plumbing: device memory is torch tensors, every call goes through the C ABI.
The library is required; there is no fallback path of any kind.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")   # RT_AMD_LIB: diagnostic builds (tools/diag)

RT_BG_BLACK, RT_BG_SKY = 0, 1
RT_INTEGRATOR_AUTO, RT_INTEGRATOR_MEGAKERNEL, RT_INTEGRATOR_WAVEFRONT = 0, 1, 2


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class RtObject(C.Structure):
    _fields_ = [("type", C.c_int32), ("flags", C.c_uint32), ("material", C.c_int32), ("prim_id", C.c_int32),
                ("op_begin", C.c_int32), ("op_count", C.c_int32), ("child_begin", C.c_int32),
                ("child_count", C.c_int32), ("bvh_root", C.c_int32), ("medium_slot", C.c_int32),
                ("f", C.c_float * 10), ("bmin", C.c_float * 3), ("bmax", C.c_float * 3)]


class RtXformOp(C.Structure):
    _fields_ = [("type", C.c_int32), ("a", C.c_float), ("b", C.c_float), ("c", C.c_float)]


class RtMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("texture", C.c_int32), ("param", C.c_float), ("pad", C.c_int32)]


class RtTexture(C.Structure):
    _fields_ = [("type", C.c_int32), ("colour", C.c_float * 3), ("even", C.c_int32), ("odd", C.c_int32),
                ("width", C.c_int32), ("height", C.c_int32), ("image_offset", C.c_uint64)]


class RtCamera(C.Structure):
    _fields_ = [("lower_left", C.c_float * 3), ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("origin", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lens_radius", C.c_float), ("time0", C.c_float), ("time1", C.c_float)]


class RtBvhNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("left_first", C.c_uint32), ("bmax", C.c_float * 3), ("count", C.c_uint32)]


class RtFlatScene(C.Structure):
    _fields_ = [("objects", C.POINTER(RtObject)), ("n_objects", C.c_int32),
                ("ops", C.POINTER(RtXformOp)), ("n_ops", C.c_int32),
                ("materials", C.POINTER(RtMaterial)), ("n_materials", C.c_int32),
                ("textures", C.POINTER(RtTexture)), ("n_textures", C.c_int32),
                ("image_bytes", C.POINTER(C.c_uint8)), ("n_image_bytes", C.c_uint64),
                ("bvh_nodes", C.POINTER(RtBvhNode)), ("n_bvh_nodes", C.c_int32),
                ("bvh_items", C.POINTER(C.c_int32)), ("n_bvh_items", C.c_int32),
                ("world", C.c_int32), ("n_prims", C.c_int32), ("n_media", C.c_int32),
                ("camera", RtCamera)]


class RtHitRecord(C.Structure):
    _fields_ = [("hit", C.c_int32), ("prim", C.c_int32), ("t", C.c_float), ("u", C.c_float), ("v", C.c_float),
                ("p", C.c_float * 3), ("n", C.c_float * 3), ("front_face", C.c_int32), ("material", C.c_int32)]


class RtCounters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64), ("render_ms", C.c_double),
                ("last_render_ms", C.c_double), ("render_launches", C.c_uint32), ("last_integrator", C.c_uint32),
                ("wf_pools", C.c_uint32), ("wf_streams_found", C.c_uint32),
                ("wf_kernel", C.c_uint32), ("reserved", C.c_uint32)]


# numpy dtype with the layout of rt_scatter_record / rto_scatter_out (60 bytes)
SCATTER_DTYPE = [("ok", "<i4"), ("attenuation", "<f4", (3,)), ("origin", "<f4", (3,)), ("direction", "<f4", (3,)),
                 ("time", "<f4"), ("emitted", "<f4", (3,)), ("draws", "<u4")]
# numpy dtype with the layout of rt_hit_record / rto_hit (52 bytes)
HIT_DTYPE = [("hit", "<i4"), ("prim", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4"),
             ("p", "<f4", (3,)), ("n", "<f4", (3,)), ("front_face", "<i4"), ("material", "<i4")]

EXPORTS = [
    "rt_last_error", "rt_scene_create_builtin", "rt_scene_from_flat", "rt_scene_flat",
    "rt_scene_default_background", "rt_scene_destroy", "rt_ctx_create", "rt_ctx_destroy",
    "rt_scene_upload", "rt_trace_rays", "rt_trace_rays_wavefront", "rt_trace_primary", "rt_shade_rays", "rt_texture_values", "rt_render_accumulate", "rt_render_accumulate_window", "rt_render_accumulate_noalloc", "rt_resolve",
    "rt_ctx_set_integrator", "rt_reduce_accum", "rt_reduce_scatter_rows", "rt_synchronize", "rt_get_counters", "rt_reset_counters", "rt_quantise_rgb8", "rt_write_ppm", "rt_write_png", "rt_read_ppm", "rt_free_pixels", "rt_flat_features", "rt_flat_program", "rt_flat_program_split", "rt_flat_bvh_stats", "rt_flat_bvh4_single_refs", "rt_flat_bvh8_check",
]


class RtError(RuntimeError):
    pass


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise RtError("%s is missing: build it with `make -C cuda-raytracer_amd` (or __graft_entry__.build())" % path)
    lib = C.CDLL(path)
    vp, i32, u32, u64, f32 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_scene_create_builtin.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(vp)]
    lib.rt_scene_from_flat.argtypes = [C.POINTER(RtFlatScene), C.POINTER(vp)]
    lib.rt_scene_flat.argtypes = [vp]
    lib.rt_scene_flat.restype = C.POINTER(RtFlatScene)
    lib.rt_scene_default_background.argtypes = [vp]
    lib.rt_scene_destroy.argtypes = [vp]
    lib.rt_scene_destroy.restype = None
    lib.rt_ctx_create.argtypes = [i32, vp, C.POINTER(vp)]
    lib.rt_ctx_destroy.argtypes = [vp]
    lib.rt_ctx_destroy.restype = None
    lib.rt_scene_upload.argtypes = [vp, C.POINTER(RtFlatScene)]
    lib.rt_trace_rays.argtypes = [vp, i32, vp, f32, f32, vp]
    lib.rt_trace_rays_wavefront.argtypes = [vp, i32, vp, vp]
    lib.rt_trace_primary.argtypes = [vp, i32, i32, f32, f32, vp, vp]
    lib.rt_shade_rays.argtypes = [vp, i32, vp, vp, vp, vp, u64, vp]
    lib.rt_texture_values.argtypes = [vp, i32, i32, vp, vp]
    lib.rt_render_accumulate.argtypes = [vp, i32, i32, u32, u32, u64, i32, i32, vp]
    lib.rt_render_accumulate_window.argtypes = [vp, i32, i32, i32, i32, i32, i32, u32, u32, u64, i32, i32, vp]
    lib.rt_render_accumulate_noalloc.argtypes = [vp, i32, i32, u32, u32, u64, i32, i32, vp]
    lib.rt_resolve.argtypes = [vp, vp, i32, i32, u32, i32, vp]
    lib.rt_ctx_set_integrator.argtypes = [vp, i32]
    lib.rt_reduce_accum.argtypes = [C.POINTER(vp), C.POINTER(vp), i32, u64, i32]
    lib.rt_reduce_scatter_rows.argtypes = [C.POINTER(vp), C.POINTER(vp), i32, i32, i32, C.POINTER(i32)]
    lib.rt_synchronize.argtypes = [vp]
    lib.rt_get_counters.argtypes = [vp, C.POINTER(RtCounters)]
    lib.rt_reset_counters.argtypes = [vp]
    lib.rt_quantise_rgb8.argtypes = [vp, i32, i32, vp]
    lib.rt_write_ppm.argtypes = [C.c_char_p, vp, i32, i32]
    lib.rt_write_png.argtypes = [C.c_char_p, vp, i32, i32]
    lib.rt_read_ppm.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]
    lib.rt_free_pixels.argtypes = [vp]
    lib.rt_free_pixels.restype = None
    lib.rt_flat_features.argtypes = [C.POINTER(RtFlatScene)]
    lib.rt_flat_features.restype = u32
    lib.rt_flat_program.argtypes = [C.POINTER(RtFlatScene), vp, i32]
    lib.rt_flat_program_split.argtypes = [C.POINTER(RtFlatScene), vp, i32, vp, i32, C.POINTER(C.c_int)]
    lib.rt_flat_bvh_stats.argtypes = [C.POINTER(RtFlatScene)] + [C.POINTER(C.c_int)] * 4
    lib.rt_flat_bvh4_single_refs.argtypes = [C.POINTER(RtFlatScene)]
    lib.rt_flat_bvh8_check.argtypes = [C.POINTER(RtFlatScene), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]
    return lib


def load_reduce_test_library():
    """TEST build of csrc/rt_reduce.cpp (-DRT_TEST_HOOKS): the exchange logic with rt_debug_reduce_with_fake; tests only"""
    path = os.path.join(_HERE, "librt_reduce_test.so")
    if not os.path.exists(path):
        raise RtError("%s is missing: build it with `make -C cuda-raytracer_amd`" % path)
    t = C.CDLL(path)
    t.rt_last_error.restype = C.c_char_p
    return t


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load_library()
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RtError("%s failed (%d): %s" % (what, rc, lib().rt_last_error().decode()))


class Scene:
    """A lowered scene (host tables)."""

    def __init__(self, name, width, height, param=0):
        h = C.c_void_p()
        _check(lib().rt_scene_create_builtin(name.encode(), width, height, param, C.byref(h)), "rt_scene_create_builtin")
        self.handle = h
        self.name, self.width, self.height = name, width, height

    @property
    def flat(self):
        return lib().rt_scene_flat(self.handle)

    @property
    def background(self):
        return lib().rt_scene_default_background(self.handle)

    def close(self):
        if self.handle:
            lib().rt_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One device + one stream.  Tensors passed in must live on that device."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        _check(lib().rt_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)), "rt_ctx_create")
        self.handle = h
        self.device = device

    def upload(self, scene):
        _check(lib().rt_scene_upload(self.handle, scene.flat), "rt_scene_upload")

    def trace_rays(self, rays, tmin, tmax):
        import numpy as np
        import torch
        assert rays.is_cuda and rays.dtype == torch.float32 and rays.is_contiguous() and rays.shape[-1] == 7
        n = rays.shape[0]
        out = torch.zeros(n * 52, dtype=torch.uint8, device=rays.device)
        torch.cuda.synchronize(rays.device)
        _check(lib().rt_trace_rays(self.handle, n, rays.data_ptr(), tmin, tmax, out.data_ptr()), "rt_trace_rays")
        self.synchronize()
        return np.frombuffer(out.cpu().numpy().tobytes(), dtype=np.dtype(HIT_DTYPE))

    def trace_rays_wavefront(self, rays):
        """closest hits through the wavefront form's intersect kernel; tmin = .001, tmax = FLT_MAX"""
        import numpy as np
        import torch
        assert rays.is_cuda and rays.dtype == torch.float32 and rays.is_contiguous() and rays.shape[-1] == 7
        n = rays.shape[0]
        out = torch.zeros(n * 52, dtype=torch.uint8, device=rays.device)
        torch.cuda.synchronize(rays.device)
        _check(lib().rt_trace_rays_wavefront(self.handle, n, rays.data_ptr(), out.data_ptr()), "rt_trace_rays_wavefront")
        self.synchronize()
        return np.frombuffer(out.cpu().numpy().tobytes(), dtype=np.dtype(HIT_DTYPE))

    def shade_rays(self, rays, recs, materials, psb, seed):
        """Material::scatter + emitted for (ray, record, material, RNG address) tuples; numpy in, numpy out"""
        import numpy as np
        import torch
        dev = torch.device("cuda", self.device)
        n = len(rays)
        t = [torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev) for a, dt in
             ((rays, np.float32), (recs, np.float32), (materials, np.int32), (np.asarray(psb, np.uint32).view(np.int32), np.int32))]
        out = torch.zeros(max(1, n) * 60, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        _check(lib().rt_shade_rays(self.handle, n, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), seed, out.data_ptr()),
               "rt_shade_rays")
        self.synchronize()
        return np.frombuffer(out.cpu().numpy().tobytes()[:n * 60], dtype=np.dtype(SCATTER_DTYPE))

    def texture_values(self, texture, uvp):
        import numpy as np
        import torch
        dev = torch.device("cuda", self.device)
        q = torch.from_numpy(np.ascontiguousarray(uvp, np.float32)).to(dev)
        rgb = torch.zeros((len(uvp), 3), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        _check(lib().rt_texture_values(self.handle, texture, len(uvp), q.data_ptr(), rgb.data_ptr()), "rt_texture_values")
        self.synchronize()
        return rgb.cpu().numpy()

    def trace_primary(self, width, height, s_off=0.5, t_off=0.5):
        import torch
        dev = torch.device("cuda", self.device)
        prim = torch.full((height, width), -2, dtype=torch.int32, device=dev)
        t = torch.zeros((height, width), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        _check(lib().rt_trace_primary(self.handle, width, height, s_off, t_off, prim.data_ptr(), t.data_ptr()), "rt_trace_primary")
        self.synchronize()
        return prim.cpu().numpy(), t.cpu().numpy()

    def render_accumulate(self, accum, width, height, s0, s1, seed, max_depth=100, background=RT_BG_BLACK):
        import torch
        assert accum.is_cuda and accum.dtype == torch.float32 and accum.is_contiguous() and accum.numel() == width * height * 3
        _check(lib().rt_render_accumulate(self.handle, width, height, s0, s1, seed, max_depth, background, accum.data_ptr()),
               "rt_render_accumulate")

    def render_accumulate_noalloc(self, accum, width, height, s0, s1, seed, max_depth=100, background=RT_BG_BLACK):
        """megakernel form with the library's built-in scratch: no allocation inside the call"""
        import torch
        assert accum.is_cuda and accum.dtype == torch.float32 and accum.is_contiguous() and accum.numel() == width * height * 3
        _check(lib().rt_render_accumulate_noalloc(self.handle, width, height, s0, s1, seed, max_depth, background, accum.data_ptr()),
               "rt_render_accumulate_noalloc")

    def render_window(self, accum, width, height, window, s0, s1, seed, max_depth=100, background=RT_BG_BLACK):
        """window = (x0, y0, x1, y1) inside the width x height frame; accum is the full frame's buffer"""
        import torch
        assert accum.is_cuda and accum.dtype == torch.float32 and accum.is_contiguous() and accum.numel() == width * height * 3
        x0, y0, x1, y1 = window
        _check(lib().rt_render_accumulate_window(self.handle, width, height, x0, y0, x1, y1, s0, s1, seed, max_depth, background,
                                                 accum.data_ptr()), "rt_render_accumulate_window")

    def resolve(self, accum, width, height, spp, gamma=True):
        import torch
        rgb = torch.empty_like(accum)
        _check(lib().rt_resolve(self.handle, accum.data_ptr(), width, height, spp, 1 if gamma else 0, rgb.data_ptr()), "rt_resolve")
        return rgb

    def set_integrator(self, mode):
        _check(lib().rt_ctx_set_integrator(self.handle, mode), "rt_ctx_set_integrator")

    def synchronize(self):
        _check(lib().rt_synchronize(self.handle), "rt_synchronize")

    def counters(self):
        c = RtCounters()
        _check(lib().rt_get_counters(self.handle, C.byref(c)), "rt_get_counters")
        return c

    def reset_counters(self):
        _check(lib().rt_reset_counters(self.handle), "rt_reset_counters")

    def close(self):
        if self.handle:
            lib().rt_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
