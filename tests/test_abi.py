"""The C-ABI library loads, exports what include/rt_api.h declares, and its host-only entry points
behave (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rt_api.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(binding):
    lib = binding.load_library()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "librt_amd.so does not export %s" % n
    assert sorted(binding.EXPORTS) == names


def test_struct_layouts(binding):
    assert C.sizeof(binding.RtObject) == 104
    assert C.sizeof(binding.RtXformOp) == 16
    assert C.sizeof(binding.RtMaterial) == 16
    assert C.sizeof(binding.RtTexture) == 40
    assert C.sizeof(binding.RtCamera) == 96          # same members as the reference Camera
    assert C.sizeof(binding.RtBvhNode) == 32
    assert C.sizeof(binding.RtHitRecord) == 52
    assert C.sizeof(binding.RtCounters) == 56
    assert np.dtype(binding.HIT_DTYPE).itemsize == 52


def test_unknown_scene_is_an_error_not_an_exit(binding):
    with pytest.raises(binding.RtError) as e:
        binding.Scene("no_such_scene", 8, 8)
    assert "unknown built-in scene" in str(e.value)
    with pytest.raises(binding.RtError):
        binding.Scene("cornell", 0, 8)


def test_quantise_flips_rows_and_clamps(binding):
    lib = binding.lib()
    W, H = 3, 2
    rgb = np.zeros((H, W, 3), np.float32)
    rgb[0, 0] = [0.0, 0.5, 1.0]      # bottom-left
    rgb[1, 2] = [2.0, -1.0, np.nan]  # top-right: above 1 must clamp (the reference wraps, main.cu:688-694)
    out = np.zeros((H, W, 3), np.uint8)
    assert lib.rt_quantise_rgb8(rgb.ctypes.data, W, H, out.ctypes.data) == 0
    assert out[1, 0].tolist() == [0, int(255.99 * 0.5), 255]      # bottom row comes out last
    assert out[0, 2].tolist() == [255, 0, 0]
    assert lib.rt_quantise_rgb8(None, W, H, out.ctypes.data) < 0


def test_write_ppm(binding, tmp_path):
    lib = binding.lib()
    img = (np.arange(4 * 2 * 3) % 256).astype(np.uint8)
    p = str(tmp_path / "x.ppm").encode()
    assert lib.rt_write_ppm(p, img.ctypes.data, 4, 2) == 0
    data = open(p, "rb").read()
    assert data.startswith(b"P6\n4 2\n255\n") and data.endswith(img.tobytes())
    assert lib.rt_write_ppm(b"/nonexistent_dir/x.ppm", img.ctypes.data, 4, 2) < 0
    assert b"cannot open" in lib.rt_last_error()


def test_from_flat_round_trip_and_validation(binding, oracle_mod):
    sc = binding.Scene("kat", 32, 24)
    h = C.c_void_p()
    assert binding.lib().rt_scene_from_flat(sc.flat, C.byref(h)) == 0
    copy = binding.lib().rt_scene_flat(h)
    assert oracle_mod.scene_digest(copy) == oracle_mod.scene_digest(sc.flat)
    binding.lib().rt_scene_destroy(h)
    # corrupt one index at a time: each must be refused on the host
    f = sc.flat.contents
    first = [f.objects[i] for i in range(f.n_objects) if f.objects[i].type < 7][0]   # some surface
    for field, bad in [("material", 10 ** 6), ("op_begin", 10 ** 6), ("type", 99)]:
        keep = getattr(first, field)
        setattr(first, field, bad)
        h2 = C.c_void_p()
        rc = binding.lib().rt_scene_from_flat(sc.flat, C.byref(h2))
        setattr(first, field, keep)
        assert rc < 0, "corrupted %s accepted" % field
        assert b"invalid scene" in binding.lib().rt_last_error()
    keep = f.bvh_items[0]
    f.bvh_items[0] = 10 ** 6
    h3 = C.c_void_p()
    assert binding.lib().rt_scene_from_flat(sc.flat, C.byref(h3)) < 0
    f.bvh_items[0] = keep
    # a wrapper chain on the world object itself: the scan would run in one frame and the shading in another
    world = f.objects[f.world]
    keep = (world.op_begin, world.op_count)
    world.op_begin, world.op_count = 0, 1
    h4 = C.c_void_p()
    assert binding.lib().rt_scene_from_flat(sc.flat, C.byref(h4)) < 0
    assert b"world object carries a transform chain" in binding.lib().rt_last_error()
    world.op_begin, world.op_count = keep
    # more BVH items than a 27-bit leaf reference can address (the count alone is checked before any item is read)
    keep = f.n_bvh_items
    f.n_bvh_items = 1 << 27
    h5 = C.c_void_p()
    assert binding.lib().rt_scene_from_flat(sc.flat, C.byref(h5)) < 0
    assert b"2^27" in binding.lib().rt_last_error()
    f.n_bvh_items = keep
    assert binding.lib().rt_scene_from_flat(sc.flat, C.byref(h5)) == 0
    binding.lib().rt_scene_destroy(h5)


def test_stale_object_bounds_keep_leaves_whole(binding):
    """rt_pack_bvh4 opens a two-item leaf into one child per item behind the ITEM's rt_object::bmin/bmax.  A caller's
    flat table may carry right node boxes and unset object bounds: such a leaf must stay whole (behind the node's box),
    or the wavefront form would silently lose hits the megakernel's binary tree finds"""
    lib = binding.lib()
    sc = binding.Scene("tri_mesh", 32, 24, 2000)
    f = sc.flat.contents
    opened = lib.rt_flat_bvh4_single_refs(sc.flat)
    assert opened > 100
    items = [f.bvh_items[i] for i in range(f.n_bvh_items)]
    keep = [(tuple(f.objects[i].bmin), tuple(f.objects[i].bmax)) for i in items]
    for case in ("zero", "nan", "outside"):
        for i in items:
            o = f.objects[i]
            for a in range(3):
                if case == "zero":
                    o.bmin[a] = o.bmax[a] = 0.0
                elif case == "nan":
                    o.bmin[a] = float("nan")
                else:
                    o.bmin[a] -= 50.0
        left = lib.rt_flat_bvh4_single_refs(sc.flat)
        n2, d2, n4, d4 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        assert lib.rt_flat_bvh_stats(sc.flat, C.byref(n2), C.byref(d2), C.byref(n4), C.byref(d4)) == 0     # still covers every item once
        for i, (lo, hi) in zip(items, keep):
            for a in range(3):
                f.objects[i].bmin[a], f.objects[i].bmax[a] = lo[a], hi[a]
        assert left < opened // 2, (case, left, opened)     # only leaves the builder itself made of one item remain
    assert lib.rt_flat_bvh4_single_refs(sc.flat) == opened


def test_test_hooks_are_not_in_the_product_library(binding):
    lib = binding.load_library()
    assert not hasattr(lib, "rt_debug_reduce_with_fake")
    assert hasattr(binding.load_reduce_test_library(), "rt_debug_reduce_with_fake")


def test_builtin_scenes_are_built_once_per_process_and_thread_safe(binding, oracle_mod):
    """rt_render builds the scene of every GPU in a thread of its own; the shared procedural texture map must come out
    the same in all of them (it used to be filled lazily without synchronisation)"""
    import threading
    digests, errors = [], []

    def build():
        try:
            sc = binding.Scene("final", 64, 48, 6)
            digests.append(oracle_mod.scene_digest(sc.flat))
        except Exception as e:       # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=build) for _ in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and len(set(digests)) == 1 and len(digests) == 8


def test_headers_are_plain_c(tmp_path):
    """the boundary is a C ABI: include/rt_api.h and rt_scene.h compile as C99 and as C++ with nothing else on
    the include path, and the structs have the sizes the bindings assume"""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "rt_api.h"\n'
                   'typedef char a1[sizeof(rt_object) == 104 ? 1 : -1];\n'
                   'typedef char a2[sizeof(rt_bvh_node) == 32 ? 1 : -1];\n'
                   'typedef char a3[sizeof(rt_hit_record) == 52 ? 1 : -1];\n'
                   'typedef char a4[sizeof(rt_camera) == 96 ? 1 : -1];\n'
                   'typedef char a5[sizeof(rt_xform_op) == 16 && sizeof(rt_material) == 16 && sizeof(rt_texture) == 40 ? 1 : -1];\n'
                   'int main(void) { const char* (*f)(void) = rt_last_error; return f != rt_last_error; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)])
