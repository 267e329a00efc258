"""What rt_scene_upload derives from the flat tables, checked on the host through the introspection entry
points of include/rt_api.h: the scan program the closest-hit scan executes in place of walking
HitableList::list (HitableList.cuh:17-32), the feature bits that pick the kernel variant, and the device BVHs."""
import ctypes as C

import numpy as np
import pytest

# rt_host.h
P_END, P_XFORM, P_RECT, P_SPHERE, P_GENERIC, P_BOX = 0, 1, 2, 3, 4, 5
F = dict(BVH=1 << 0, GROUPLIST=1 << 1, MEDIA=1 << 2, SPHERE=1 << 3, MSPHERE=1 << 4, TRI=1 << 5, PLANAR=1 << 6,
         LONGCHAIN=1 << 7, LENS=1 << 8, GLASS=1 << 9, METAL=1 << 10, ISO=1 << 11, CHECKER=1 << 12, IMAGE=1 << 13,
         XFORM=1 << 14, NOISE=1 << 15)


def program(binding, sc):
    n = binding.lib().rt_flat_program(sc.flat, None, 0)
    assert n >= 2
    buf = (C.c_uint32 * (8 * n))()
    assert binding.lib().rt_flat_program(sc.flat, buf, n) == n
    w = np.frombuffer(buf, dtype=np.uint32).reshape(n, 8).copy()
    return w, w.view(np.float32)


def test_cornell_program(binding):
    sc = binding.Scene("cornell", 64, 64)
    w, f = program(binding, sc)
    kinds = list(w[:, 0] & 0xff)
    # five walls + light in list order, then each box under its own Translate(RotateY()) frame, two terminators
    assert kinds == [P_RECT] * 6 + [P_XFORM, P_BOX, P_XFORM, P_BOX, P_END, P_END]
    fl = sc.flat.contents
    world = fl.objects[fl.world]
    assert list(w[:6, 1]) == [world.child_begin + i for i in range(6)]
    # axis of the fixed coordinate: YZRect x, YZRect x, XZRect y (light), XZRect y, XZRect y, XYRect z
    assert list((w[:6, 0] >> 8) & 0xff) == [0, 0, 1, 1, 1, 2]
    light = f[2]
    assert tuple(light[2:7]) == (554.0, 113.0, 443.0, 127.0, 432.0)          # k, a0, a1, b0, b1 (main.cu:288)
    x = w[6]
    assert (x[0] >> 8) & 0xff == 2 and (x[0] >> 16) & 0xf == 0 and (x[0] >> 20) & 0xf == 1   # Translate, then RotateY
    assert tuple(f[6][2:5]) == (130.0, 0.0, 65.0)
    assert np.isclose(f[6][5], np.sin(np.radians(-18.0)), atol=1e-7) and np.isclose(f[6][6], np.cos(np.radians(-18.0)), atol=1e-7)
    assert tuple(f[7][2:8]) == (0.0, 0.0, 0.0, 165.0, 165.0, 165.0)


def test_generic_records_and_frames(binding):
    """media, groups and moving spheres go through their rows; the frame is re-established after them"""
    sc = binding.Scene("cornell_smoke", 64, 64)
    w, _ = program(binding, sc)
    assert list(w[:, 0] & 0xff) == [P_RECT] * 6 + [P_GENERIC, P_GENERIC, P_END, P_END]
    sc = binding.Scene("final", 64, 64, 6)
    w, _ = program(binding, sc)
    kinds = list(w[:, 0] & 0xff)
    # BvhNode group, light rect (frame reset first), moving sphere, three spheres, two media, sphere, instanced group
    assert kinds == [P_GENERIC, P_XFORM, P_RECT, P_GENERIC, P_XFORM, P_SPHERE, P_SPHERE, P_SPHERE, P_GENERIC, P_GENERIC,
                     P_XFORM, P_SPHERE, P_GENERIC, P_END, P_END]
    assert all(((w[i, 0] >> 8) & 0xff) == 0 for i, k in enumerate(kinds) if k == P_XFORM)   # all world-frame resets
    # a world that has a BVH of its own is traversed directly: no scan records
    sc = binding.Scene("random_spheres", 64, 64)
    w, _ = program(binding, sc)
    assert list(w[:, 0] & 0xff) == [P_END, P_END]


@pytest.mark.parametrize("name,param,want,dont", [
    ("cornell", 0, ["PLANAR", "XFORM"], ["BVH", "MEDIA", "SPHERE", "TRI", "GLASS", "METAL", "ISO", "LENS", "IMAGE", "CHECKER"]),
    ("cornell_smoke", 0, ["PLANAR", "XFORM", "MEDIA", "ISO"], ["BVH", "SPHERE", "GLASS"]),
    ("tri_mesh", 2000, ["BVH", "TRI", "PLANAR"], ["XFORM", "MEDIA", "SPHERE", "GLASS", "METAL", "LENS"]),
    ("random_spheres", 0, ["BVH", "SPHERE", "MSPHERE", "GLASS", "METAL", "CHECKER", "IMAGE"], ["MEDIA", "TRI", "LENS"]),
    ("lit_sphere", 0, ["SPHERE", "XFORM", "LENS", "IMAGE", "CHECKER"], ["BVH", "PLANAR"]),
    ("final", 6, ["BVH", "MEDIA", "SPHERE", "MSPHERE", "PLANAR", "XFORM", "GLASS", "METAL", "ISO", "IMAGE"], ["TRI", "LENS", "NOISE"]),
    ("perlin", 0, ["NOISE", "SPHERE", "PLANAR", "XFORM", "METAL", "CHECKER"], ["BVH", "MEDIA", "IMAGE", "GLASS"]),
])
def test_feature_bits(binding, name, param, want, dont):
    sc = binding.Scene(name, 64, 64, param)
    bits = binding.lib().rt_flat_features(sc.flat)
    for k in want:
        assert bits & F[k], k
    for k in dont:
        assert not bits & F[k], k


@pytest.mark.parametrize("name,param", [("tri_mesh", 30000), ("final", 12), ("cornell_cluster", 0), ("random_spheres", 0), ("kat", 0)])
def test_device_bvhs(binding, name, param):
    """both device trees are built from the ABI tree; the call itself checks that the 4-wide trees' leaf
    references cover the leaf table exactly once"""
    sc = binding.Scene(name, 64, 64, param)
    n2, d2, n4, d4 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = binding.lib().rt_flat_bvh_stats(sc.flat, C.byref(n2), C.byref(d2), C.byref(n4), C.byref(d4))
    assert rc == 0, binding.lib().rt_last_error()
    fl = sc.flat.contents
    interior = sum(1 for i in range(fl.n_bvh_nodes) if fl.bvh_nodes[i].count == 0)
    assert n2.value == interior                      # one two-box node per interior ABI node
    assert 0 < n4.value <= n2.value                  # collapsing only removes nodes
    assert n4.value >= (n2.value + 2) // 3           # a 4-wide node absorbs at most two further interior nodes
    assert 0 < d4.value <= d2.value and 3 * d4.value + 1 <= 64


@pytest.mark.parametrize("name,param", [("kat", 0), ("random_spheres", 0), ("final", 12), ("tri_mesh", 5000), ("cornell_cluster", 0), ("bvh_boxes", 0)])
def test_eight_wide_trees(binding, name, param):
    """rt_flat_bvh8_check builds the eight-wide quantised trees (rt_pack_bvh8) and walks them: every BVH item sits in
    exactly one leaf row, inside its slot's decoded box and inside the decoded slot box of every ancestor"""
    sc = binding.Scene(name, 64, 64, param)
    d8, rows, slack = C.c_int(), C.c_int(), C.c_float()
    rc = binding.lib().rt_flat_bvh8_check(sc.flat, C.byref(d8), C.byref(rows), C.byref(slack))
    assert rc > 0, binding.lib().rt_last_error()          # the number of eight-wide nodes
    assert rows.value == sc.flat.contents.n_bvh_items
    n2, d2, n4, d4 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert binding.lib().rt_flat_bvh_stats(sc.flat, C.byref(n2), C.byref(d2), C.byref(n4), C.byref(d4)) == 0
    assert 0 < d8.value <= d4.value + 1      # eight children a node: about two thirds of the four-wide tree's depth
    assert rc <= n4.value
    assert slack.value >= 0.0


def test_scene_without_bvh(binding):
    sc = binding.Scene("cornell", 32, 32)
    n2, d2, n4, d4 = C.c_int(7), C.c_int(7), C.c_int(7), C.c_int(7)
    assert binding.lib().rt_flat_bvh_stats(sc.flat, C.byref(n2), C.byref(d2), C.byref(n4), C.byref(d4)) == 0
    assert (n2.value, d2.value, n4.value, d4.value) == (0, 0, 0, 0)


def split_program(binding, sc):
    lib = binding.lib()
    ng = C.c_int()
    n = lib.rt_flat_program_split(sc.flat, None, 0, None, 0, C.byref(ng))
    assert n >= 2
    buf = (C.c_uint32 * (8 * n))()
    groups = (C.c_int32 * max(1, ng.value))()
    assert lib.rt_flat_program_split(sc.flat, buf, n, groups, ng.value, C.byref(ng)) == n
    return np.frombuffer(buf, dtype=np.uint32).reshape(n, 8).copy(), list(groups)[:ng.value]


@pytest.mark.parametrize("name,param", [("final", 12), ("cornell_cluster", 0), ("tri_mesh", 3000), ("random_spheres", 0), ("cornell", 0), ("kat", 0)])
def test_program_split_for_the_wavefront_form(binding, name, param):
    """the wavefront integrator runs the plain records when it hands a ray to a lane and then walks the BVH groups:
    together the two parts are the whole list, order kept within each"""
    sc = binding.Scene(name, 64, 64, param)
    full, _ = program(binding, sc)
    plain, groups = split_program(binding, sc)
    fl = sc.flat.contents
    world = fl.objects[fl.world]
    is_bvh_group = lambda oi: fl.objects[oi].type == 8 and fl.objects[oi].bvh_root >= 0
    if world.bvh_root >= 0:
        assert groups == [fl.world] and [int(k) & 0xff for k in plain[:, 0]] == [P_END, P_END]
        return
    want_groups = [int(r[1]) for r in full if (int(r[0]) & 0xff) == P_GENERIC and is_bvh_group(int(r[1]))]
    assert groups == want_groups
    kept = [r for r in full if not ((int(r[0]) & 0xff) == P_GENERIC and is_bvh_group(int(r[1])))]
    assert np.array_equal(plain, np.array(kept, dtype=np.uint32))
    assert not any((int(r[0]) & 0xff) == P_GENERIC and is_bvh_group(int(r[1])) for r in plain)
    if name in ("final", "cornell_cluster", "tri_mesh", "kat"):      # kat: its surfaces were gathered into one BVH group
        assert len(groups) >= 1
    if name == "cornell":
        assert groups == []
