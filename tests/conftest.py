import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FLT_MAX = float(np.finfo(np.float32).max)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def binding():
    return entry.load_binding()


@pytest.fixture(scope="session")
def oracle_mod():
    return entry.load_oracle()


@pytest.fixture(scope="session")
def oracle(oracle_mod):
    return oracle_mod.Oracle()


@pytest.fixture(scope="session")
def gpu_ctx(binding):
    import torch
    assert torch.cuda.is_available(), "GPU test without a GPU"
    ctx = binding.Context(0)
    yield ctx
    ctx.close()


def load_golden(name):
    return np.load(os.path.join(GOLD, name))


def made_by_reference_geometry(who):
    """fixtures that pin parity must come from oracle/_ref/libref_geom.so, the host build of the reference's
    geometry + texture headers that contains no stand-in (oracle/ref_geom.cpp)"""
    who = str(who)
    return who.startswith("libref_geom.so") and "no stand-in" in who


def golden_scene(binding, oracle_mod, gold, name):
    """the scene a fixture was made from; refuses a stale fixture and one not made by the reference's geometry build"""
    assert made_by_reference_geometry(gold["produced_by"]), gold["produced_by"]
    sc = binding.Scene(name, int(gold["width"]), int(gold["height"]), int(gold["param"]))
    assert oracle_mod.scene_digest(sc.flat) == str(gold["digest"]), (
        "tests/golden fixture for %s is stale: re-run oracle/gen_golden.py where /root/reference exists" % name)
    return sc


SPECIAL_SCENES = ["kat", "cornell", "bvh_boxes", "tri_mesh", "cornell_cluster"]      # tests/golden/geom_special_ref.npz


def special_scene(binding, oracle_mod, gold, name):
    """the scene, rays and reference hits of one entry of geom_special_ref.npz (oracle/gen_golden.py, special_rays)"""
    assert made_by_reference_geometry(gold["produced_by"]), gold["produced_by"]
    sc = binding.Scene(name, 128, 72, int(gold["param_" + name]))
    assert oracle_mod.scene_digest(sc.flat) == str(gold["digest_" + name]), "geom_special_ref.npz is stale for " + name
    return sc, gold["rays_" + name], gold["hits_" + name]


def special_order_free(gold, name):
    """The special rays whose closest hit does not depend on the order in which the primitives are tested, i.e. the ones a
    BVH must answer like the reference's flat list.  Left out: (i) rays that can meet 0 / 0 — a zero direction component
    with the origin on a bounding plane (`axis_on_surface`, `in_plane`, and corner rays that happen to share a
    coordinate with the camera): a rect in whose plane the ray lies reports a hit with t = NaN (Rectangle.cuh:27-33), and
    a NaN `closest so far` then accepts or rejects whatever is tested next; (ii) `far`: origins at 1e7 and beyond, where
    box culling and the primitives' own arithmetic round differently."""
    kinds = [str(k) for k in gold["kinds"]]
    kind = gold["kind_" + name]
    d = gold["rays_" + name][:, 3:6]
    generic = (d != 0).all(axis=1) & np.isin(kind, [kinds.index("corner"), kinds.index("scaled")])
    return generic | (kind == kinds.index("axis_offset"))


GEOM_SCENES = ["kat", "cornell", "lit_sphere", "bvh_boxes", "random_spheres", "cornell_cluster", "final", "tri_mesh", "final_custom"]
RECT_TYPES = (2, 3, 4, 5)       # front_face is only defined by the reference for rects and boxes
MOVING_SPHERE = 1               # u,v are left stale by the reference (Sphere.cuh:91-120)


def prim_types(flat):
    f = flat.contents
    t = {}
    for i in range(f.n_objects):
        o = f.objects[i]
        if o.prim_id >= 0:
            t[o.prim_id] = (o.type, o.op_count)
    return t


def exclude_ties(rays, hits, flat, oracle, rel=1e-5):
    """mask of rays whose nearest two candidates are closer than rel (ids there depend on visiting order)"""
    h = hits["hit"] == 1
    t = hits["t"]
    lo = oracle.trace_rays(flat, rays, 0.001, FLT_MAX)        # same as hits (list order)
    # second query just past the winner: another surface within rel*t means a near-tie
    nxt = rays.copy()
    mask = np.zeros(len(rays), bool)
    again = oracle.trace_rays(flat, nxt, 0.001, FLT_MAX, use_bvh=True)
    mask |= (again["prim"] != lo["prim"])
    del t, h
    return mask
