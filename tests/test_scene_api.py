"""Scene-description API and lowering (host only)."""
import os
import re
import subprocess
import textwrap

import numpy as np
import pytest

from conftest import FLT_MAX, ROOT

API = os.path.join(ROOT, "cuda-raytracer_amd", "api")
PKG = os.path.join(ROOT, "cuda-raytracer_amd")
REF_MAIN = "/root/reference/raytracer/main.cu"


def test_cornell_lowering(binding):
    sc = binding.Scene("cornell", 64, 64)
    f = sc.flat.contents
    assert (f.n_objects, f.n_prims, f.n_media, f.n_bvh_nodes) == (9, 8, 0, 0)
    w = f.objects[f.world]
    assert (w.type, w.child_count) == (8, 8)
    kids = [f.objects[w.child_begin + i] for i in range(8)]
    assert [k.type for k in kids] == [4, 4, 3, 3, 3, 2, 5, 5]          # list order of main.cu:286-295
    assert [k.prim_id for k in kids] == list(range(8))
    assert kids[5].flags & 1                                            # FlipNormals lowers to a flag
    assert [k.op_count for k in kids] == [0, 0, 0, 0, 0, 0, 2, 2]
    ops = [f.ops[kids[6].op_begin + i] for i in range(2)]
    assert [o.type for o in ops] == [0, 1]                              # Translate outermost, then RotateY
    assert (ops[0].a, ops[0].b, ops[0].c) == (130.0, 0.0, 65.0)
    # world bounds of the rotated+translated box contain its translated base corner
    assert kids[6].bmin[1] == 0.0 and kids[6].bmax[1] == 165.0
    assert sc.background == binding.RT_BG_BLACK


def test_smoke_and_final_lowering(binding):
    sc = binding.Scene("cornell_smoke", 64, 64)
    f = sc.flat.contents
    assert f.n_media == 2
    w = f.objects[f.world]
    med = [f.objects[w.child_begin + i] for i in range(w.child_count) if f.objects[w.child_begin + i].type == 7]
    assert [m.medium_slot for m in med] == [0, 1]
    for m in med:
        b = f.objects[m.child_begin]
        assert b.type == 5 and b.op_count == 2 and b.prim_id == -1
        assert f.materials[m.material].type == 4                        # Isotropic
        assert m.f[0] == np.float32(-1) / np.float32(.01)
    sc = binding.Scene("final", 64, 64, 20)
    f = sc.flat.contents
    w = f.objects[f.world]
    kinds = [f.objects[w.child_begin + i].type for i in range(w.child_count)]
    assert kinds == [8, 3, 1, 0, 0, 0, 7, 7, 0, 8]                      # main.cu:382-399 list order
    groups = [f.objects[w.child_begin + i] for i in (0, 9)]
    assert groups[0].child_count == 400 and groups[1].child_count == 1000
    assert all(g.bvh_root >= 0 for g in groups) and groups[1].op_count == 2
    assert binding.Scene("random_spheres", 64, 64).background == binding.RT_BG_SKY


def test_mixed_world_gets_a_surface_bvh(binding):
    """kat's world holds a medium: its surfaces move into one BVH group, the medium stays in the list"""
    sc = binding.Scene("kat", 64, 64)
    f = sc.flat.contents
    w = f.objects[f.world]
    kids = [f.objects[w.child_begin + i] for i in range(w.child_count)]
    assert [k.type for k in kids] == [8, 7]
    assert kids[0].bvh_root >= 0 and kids[0].child_count == 12


@pytest.mark.parametrize("name,param", [("kat", 0), ("random_spheres", 0), ("final", 12), ("tri_mesh", 5000), ("cornell_cluster", 0)])
def test_bvh_is_well_formed(binding, name, param):
    sc = binding.Scene(name, 64, 64, param)
    f = sc.flat.contents
    assert f.n_bvh_nodes > 0
    for gi in range(f.n_objects):
        g = f.objects[gi]
        if g.type != 8 or g.bvh_root < 0:
            continue
        seen = []
        stack = [(g.bvh_root, 1)]
        deepest = 0
        while stack:
            ni, d = stack.pop()
            n = f.bvh_nodes[ni]
            deepest = max(deepest, d)
            if n.count:
                for k in range(n.count):
                    oi = f.bvh_items[n.left_first + k]
                    o = f.objects[oi]
                    seen.append(oi)
                    for a in range(3):      # leaf boxes contain their objects' bounds
                        assert n.bmin[a] <= o.bmin[a] and n.bmax[a] >= o.bmax[a]
            else:
                for c in (n.left_first, n.left_first + 1):
                    ch = f.bvh_nodes[c]
                    for a in range(3):      # children inside parents
                        assert n.bmin[a] <= ch.bmin[a] and n.bmax[a] >= ch.bmax[a]
                    stack.append((c, d + 1))
        assert sorted(seen) == list(range(g.child_begin, g.child_begin + g.child_count))
        assert deepest <= 48


def test_triangle_bounds_cover_what_hit_reports(binding, oracle):
    """Triangle::hit reports the mirrored-plane image of the triangle (Triangle.cuh:24,26); the bounds
    used by the BVH must contain every point it can report"""
    sc = binding.Scene("tri_mesh", 64, 64, 400)
    f = sc.flat.contents
    rng = np.random.default_rng(3)
    tris = [i for i in range(f.n_objects) if f.objects[i].type == 6][:50]
    for oi in tris:
        o = f.objects[oi]
        lo, hi = np.array(o.bmin[:]), np.array(o.bmax[:])
        org = 0.5 * (lo + hi) + rng.normal(size=(64, 3)) * 50
        tgt = lo + rng.random((64, 3)) * (hi - lo)
        rays = np.concatenate([org, tgt - org, np.zeros((64, 1))], axis=1).astype(np.float32)
        h = oracle.object_hit(sc.flat, oi, rays, 0.001, FLT_MAX)
        p = h["p"][h["hit"] == 1]
        assert len(p) > 0
        assert np.all(p >= lo - 1e-3) and np.all(p <= hi + 1e-3)


SCENE_SRC = r'''
#include <iostream>
#include <float.h>
#include "Vector3.cuh"
#include "Ray.cuh"
#include "HitableList.cuh"
#include "Sphere.cuh"
#include "Triangle.cuh"
#include "Camera.cuh"
#include "Material.cuh"
#include "Surface.cuh"
#include "Rectangle.cuh"
#include "Box.cuh"
#include "Volumes.cuh"
#include "BvhNode.cuh"
#include "Transform.cuh"
#include "../csrc/lower.h"
#define RND (curand_uniform(&localRandState))
%s
int main()
{
    rt::Arena arena;
    curandState rs; curand_init(419, 0, 0, &rs);
    Hitable** dList = new Hitable*[4096];
    Hitable* world = nullptr; Camera* cam = nullptr;
    unsigned char tex[3 * 4 * 2] = {0};
    ImageTexture* img = new ImageTexture(tex, 4, 2);
    %s
    rt::FlatScene flat; std::string err;
    if (!rt::lower(world, cam, rt::LowerOptions(), flat, err)) { std::cout << "ERR " << err << "\n"; return 1; }
    std::cout << flat.view.n_objects << " " << flat.view.n_prims << " " << flat.view.n_media << " "
              << flat.objects[flat.view.world].child_count << "\n";
    return 0;
}
'''


def compile_and_run(tmp_path, body, call):
    src = tmp_path / "user_scene.cpp"
    src.write_text(SCENE_SRC % (body, call))
    exe = tmp_path / "user_scene"
    cmd = ["g++", "-std=c++17", "-O1", "-w", "-I", API, str(src), "-o", str(exe),
           "-L", PKG, "-lrt_amd", "-Wl,-rpath," + PKG, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"]
    subprocess.check_call(cmd)
    out = subprocess.check_output([str(exe)]).decode().split()
    return [int(x) if x.lstrip("-").isdigit() else x for x in out]


def test_user_scene_source_in_reference_style_compiles(tmp_path):
    """a scene function written the way main.cu's sceneN are (same signature, __device__, RND macro,
    `new` everywhere) builds against the api/ headers and lowers"""
    body = textwrap.dedent('''
        __device__ inline void myScene(Hitable** dList, Hitable** dWorld, Camera** dCamera, int width, int height,
                                       ImageTexture** texture, curandState* randState)
        {
            curandState localRandState = *randState;
            Material* white = new Lambert(new ConstantTexture(Vector3(.7f, .7f, .7f)));
            int i = 0;
            dList[i++] = new Sphere(Vector3(0, -100.5f, -1), 100, new Lambert(new CheckerTexture(
                new ConstantTexture(Vector3(.2f, .3f, .1f)), new ConstantTexture(Vector3(.9f, .9f, .9f)))));
            for (int k = 0; k < 5; k++)
                dList[i++] = new MovingSphere(Vector3(k + RND, .2f, RND), Vector3(k, .4f, 0) + Vector3(0, .5f * RND, 0), 0.f, 1.f, .2f, white);
            dList[i++] = new Sphere(Vector3(0, 1, 0), 1.f, new Lambert(*texture));
            dList[i++] = new Sphere(Vector3(4, 1, 0), 1.0, new Metal(new ConstantTexture(Vector3(0.7, 0.6, 0.5)), .1));
            dList[i++] = new Sphere(Vector3(-4, 1, 0), 1.0, new Dielectric(1.5f));
            dList[i++] = new FlipNormals(new XYRect(0, 5, 0, 5, 5, white));
            Hitable* b1 = new Translate(new RotateY(new Box(Vector3(0, 0, 0), Vector3(1, 1, 1), white), -18), Vector3(1, 0, 1));
            dList[i++] = new ConstantMedium(b1, .01, new ConstantTexture(Vector3(1.f, 1.f, 1.f)), randState);
            dList[i++] = new Triangle(Vector3(0, 0, 0), Vector3(1, 0, 0), Vector3(0, 1, 0), white);
            Hitable** many = new Hitable*[64];
            for (int j = 0; j < 64; j++) many[j] = new Sphere(Vector3(RND, RND, RND), .05f, white);
            dList[i++] = new Translate(new RotateY(new BvhNode(many, 64, 0.f, 1.f, randState), 15), Vector3(-1, 2, 3));
            *randState = localRandState;
            *dWorld = new HitableList(dList, i);
            *dCamera = new Camera(Vector3(13, 2, 3), Vector3(0, 0, 0), Vector3(0, 1, 0), 20.f, float(width) / float(height), 0.1f, 10.f, 0.f, 1.f);
        }
    ''')
    out = compile_and_run(tmp_path, body, "myScene(dList, &world, &cam, 400, 225, &img, &rs);")
    n_objects, n_prims, n_media, n_world = out
    assert n_world == 3 and n_media == 1          # 11 surfaces -> one BVH group, + the medium, + the sphere group
    assert n_prims == 12 + 64                     # leaves: 12 in the list (medium counts) + 64 in the group


def test_camera_has_the_reference_public_members(tmp_path):
    """Camera.cuh:47-49 declares lowerLeftCorner, horizontal, vertical, origin, u, v, w, lensRadius, time0, time1 as public
    data: a user program that reads them compiles, and sees bit for bit what the lowered camera block holds"""
    body = textwrap.dedent('''
        #include <cstring>
        __device__ inline void camScene(Hitable** dList, Hitable** dWorld, Camera** dCamera, int width, int height,
                                        ImageTexture** texture, curandState* randState)
        {
            dList[0] = new Sphere(Vector3(0, 0, 0), 1.f, new Lambert(new ConstantTexture(Vector3(.5f, .5f, .5f))));
            *dWorld = new HitableList(dList, 1);
            *dCamera = new Camera(Vector3(13, 2, 3), Vector3(0, .3f, -1), Vector3(0, 1, 0), 27.5f, float(width) / float(height), 0.1f, 9.7f, 0.25f, 1.5f);
        }
        static int same(const Vector3& a, const float* b) { return std::memcmp(a.e, b, 12) == 0; }
    ''')
    call = textwrap.dedent('''
        camScene(dList, &world, &cam, 400, 225, &img, &rs);
        {
            rt::FlatScene fs; std::string e;
            if (!rt::lower(world, cam, rt::LowerOptions(), fs, e)) return 4;
            const rt_camera& c = fs.view.camera;
            Camera* k = cam;
            int ok = same((*&k)->origin, c.origin) && same(k->lowerLeftCorner, c.lower_left) && same(k->horizontal, c.horizontal)
                  && same(k->vertical, c.vertical) && same(k->u, c.u) && same(k->v, c.v) && same(k->w, c.w)
                  && k->lensRadius == c.lens_radius && k->time0 == c.time0 && k->time1 == c.time1
                  && k->lensRadius == 0.05f && k->time0 == 0.25f && k->time1 == 1.5f && k->origin.x() == 13.f;
            if (!ok) { std::cout << "MEMBERS DIFFER\\n"; return 5; }
        }
    ''')
    out = compile_and_run(tmp_path, body, call)
    assert out[-1] == 1 and "MEMBERS" not in out


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference sources are only present in the build container")
def test_reference_scene_functions_compile_unchanged(tmp_path):
    """scene1..scene7 are taken from the reference's main.cu AT TEST TIME (nothing is stored in this
    repository) and must compile unchanged against the api/ headers and lower."""
    text = open(REF_MAIN).read()
    start = text.index("//Scene 1: Loads Of Spheres!")
    end = text.index("//Select active scene here")
    body = text[start:end]
    assert len(re.findall(r"__device__ inline void scene\d", body)) == 7
    expect = {1: 488, 2: 3, 3: 8, 5: 10, 6: 8, 7: 11}
    calls = "\n    ".join(
        "{ Hitable* w = nullptr; Camera* c = nullptr; scene%d(dList, &w, &c, 400, 225, &img, &rs); "
        "rt::FlatScene fs; std::string e; if (!rt::lower(w, c, rt::LowerOptions(), fs, e) || "
        "fs.objects[fs.view.world].child_count != %d) { std::cout << \"BAD %d \" << e << \"\\n\"; return 2; } }" % (k, n, k)
        for k, n in expect.items())
    calls += "\n    { Hitable* w = nullptr; Camera* c = nullptr; scene4(dList, &w, &c, 400, 225, &rs); if (!w || !c) return 3; }"
    calls += "\n    scene3(dList, &world, &cam, 400, 225, &img, &rs);"
    out = compile_and_run(tmp_path, body, calls)
    assert out[-4:] == [9, 8, 0, 8], out
