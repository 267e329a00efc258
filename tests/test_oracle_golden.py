"""The CPU oracle (oracle/rt_oracle.c) against the golden vectors recorded from the host build of the
reference's own headers (oracle/gen_golden.py).  Bit-exact: same compiler-independent IEEE float
operations in the same order, no FMA."""
import numpy as np
import pytest

from conftest import (FLT_MAX, GEOM_SCENES, MOVING_SPHERE, RECT_TYPES, SPECIAL_SCENES, golden_scene, load_golden,
                      made_by_reference_geometry, prim_types, special_order_free, special_scene)


def check_hits(got, exp, types, what):
    assert np.array_equal(got["hit"], exp["hit"]), what + ": hit flags"
    h = exp["hit"] == 1
    for k in ("prim", "t", "p", "n"):
        assert np.array_equal(got[k][h], exp[k][h]), what + ": " + k
    ty = np.array([types[int(p)][0] for p in exp["prim"][h]], dtype=np.int32)
    uv_ok = ty != MOVING_SPHERE        # reference leaves u,v stale there
    uv_ok &= ty != 6                   # Triangle leaves them unset too
    assert np.array_equal(got["u"][h][uv_ok], exp["u"][h][uv_ok]), what + ": u"
    assert np.array_equal(got["v"][h][uv_ok], exp["v"][h][uv_ok]), what + ": v"
    ff = np.isin(ty, RECT_TYPES)       # bFrontFace is only written by rects / boxes / RotateY
    assert np.array_equal(got["front_face"][h][ff] != 0, exp["front_face"][h][ff] != 0), what + ": front_face"


@pytest.mark.parametrize("name", GEOM_SCENES)
def test_geometry_matches_reference(binding, oracle_mod, oracle, name):
    g = load_golden("geom_%s_ref.npz" % name)
    sc = golden_scene(binding, oracle_mod, g, name)
    types = prim_types(sc.flat)
    f = sc.flat.contents
    for k, oi in enumerate(g["obj_index"]):
        got = oracle.object_hit(sc.flat, int(oi), g["obj_rays"][k], 0.001, FLT_MAX)
        if f.objects[int(oi)].type == 8:       # a group: rotated groups re-face normals, check all fields
            pass
        check_hits(got, g["obj_hits"][k], types, "%s object %d" % (name, oi))
    lo, hi = (float(x) for x in g["win_range"])
    got = oracle.object_hit(sc.flat, int(g["obj_index"][0]), g["win_rays"], lo, hi)
    check_hits(got, g["win_hits"], types, name + " bounded range")
    got = oracle.trace_rays(sc.flat, g["world_rays"], 0.001, FLT_MAX)
    check_hits(got, g["world_hits"], types, name + " world")
    prim, t = oracle.trace_primary(sc.flat, int(g["width"]), int(g["height"]))
    assert np.array_equal(prim, g["primary_prim"])
    assert np.array_equal(t, g["primary_t"])


@pytest.mark.parametrize("name", ["tri_mesh", "final"])
def test_full_size_primary_map_matches_reference(binding, oracle_mod, oracle, name):
    """BASELINE.json's target scenes at their real size (1,000,000 triangles; final scene with 100 x 100 boxes): the
    restatement's BVH traversal against the primary-hit map the reference's own flat HitableList over every primitive
    gave (Triangle.cuh:21-46 / Box.cuh through libref_geom.so, ~10^10 primitive tests, oracle/gen_golden.py --full)"""
    g = load_golden("geom_%s_full_ref.npz" % name)
    sc = golden_scene(binding, oracle_mod, g, name)
    assert sc.flat.contents.n_bvh_items >= (999000 if name == "tri_mesh" else 11000)
    prim, t = oracle.trace_primary(sc.flat, int(g["width"]), int(g["height"]), use_bvh=True)
    differ = prim != g["primary_prim"]
    assert np.array_equal(t[differ], g["primary_t"][differ]), "ids may only differ at exact ties"
    assert differ.mean() < 0.01
    assert np.array_equal(t, g["primary_t"])
    assert (g["primary_prim"] >= 0).mean() > 0.4      # media are not part of a bare closest-hit query


@pytest.mark.parametrize("name", SPECIAL_SCENES)
def test_special_rays_match_reference(binding, oracle_mod, oracle, name):
    """Rays the random fixtures never draw (oracle/gen_golden.py special_rays): direction components of +0 / -0 (divisors
    of zero, 0 * inf), origins on surfaces, rays through box corners and triangle vertices, directions of 1e18 and
    1e-30.  The reference's flat list reports a rect hit with t = NaN for a ray that lies in the rect's plane (0 / 0
    passes every `<` / `>` rejection of Rectangle.cuh:27-33); the restatement has to agree NaN for NaN."""
    g = load_golden("geom_special_ref.npz")
    sc, rays, exp = special_scene(binding, oracle_mod, g, name)
    got = oracle.trace_rays(sc.flat, rays, 0.001, FLT_MAX)
    assert np.array_equal(got["hit"], exp["hit"])
    h = exp["hit"] == 1
    assert np.array_equal(got["prim"][h], exp["prim"][h])
    for k in ("t", "p", "n"):
        assert np.array_equal(got[k][h], exp[k][h], equal_nan=True), k
    if name in ("kat", "cornell", "bvh_boxes"):
        assert np.isnan(exp["t"][h]).any(), "the fixture is meant to hold in-plane rays"
    # through the host-built BVH: the same on every ray whose answer does not depend on the order of the tests
    bvh = oracle.trace_rays(sc.flat, rays, 0.001, FLT_MAX, use_bvh=True)
    m = special_order_free(g, name)
    assert m.sum() > 600
    assert np.array_equal(bvh["hit"][m], exp["hit"][m])
    hm = h & m
    assert np.array_equal(bvh["t"][hm], exp["t"][hm])
    differ = bvh["prim"][hm] != exp["prim"][hm]      # shared corners and edges: another id only at the identical t (checked above)
    assert differ.mean() < 0.5


@pytest.mark.parametrize("name", GEOM_SCENES)
def test_bvh_traversal_equals_flat_list(binding, oracle_mod, oracle, name):
    """closest hit through the host-built BVH == HitableList scan (exact ties excluded)"""
    g = load_golden("geom_%s_ref.npz" % name)
    sc = golden_scene(binding, oracle_mod, g, name)
    flat = oracle.trace_rays(sc.flat, g["world_rays"], 0.001, FLT_MAX)
    bvh = oracle.trace_rays(sc.flat, g["world_rays"], 0.001, FLT_MAX, use_bvh=True)
    assert np.array_equal(flat["hit"], bvh["hit"])
    assert np.array_equal(flat["t"], bvh["t"])
    diff = flat["prim"] != bvh["prim"]
    # a different id at an identical t is a tie between touching surfaces (e.g. neighbouring ground boxes)
    assert np.array_equal(flat["t"][diff], bvh["t"][diff])
    assert diff.mean() < 0.05


def test_materials_match_reference(binding, oracle_mod, oracle):
    """scatter()/emitted() of the reference's Material.cuh.  These vectors come from libref_shade.so, where curandState /
    curand_uniform are bound to rocRAND (the CUDA toolkit is absent): a stand-in, so they cross-check the restatement's
    arithmetic but are not counted as a reference pin."""
    g = load_golden("shade_kat_ref.npz")
    assert str(g["mat_produced_by"]).startswith("libref_shade.so") and "stand-in" in str(g["mat_produced_by"])
    sc = binding.Scene("kat", 128, 72)
    assert oracle_mod.scene_digest(sc.flat) == str(g["digest"])
    XI = int(g["xi_len"])
    kinds = set()
    for k, mi in enumerate(g["mat_index"]):
        ps = g["mat_pixel_sample"][k]
        xi = np.stack([oracle.uniforms(int(g["seed"]), int(a), int(b), XI) for a, b in ps])
        got = oracle.scatter(sc.flat, int(mi), g["mat_rays"][k], g["mat_recs"][k], xi)
        exp = g["mat_out"][k]
        kinds.add(sc.flat.contents.materials[int(mi)].type)
        assert np.array_equal(got["ok"], exp["ok"])
        assert np.array_equal(got["draws"], exp["draws"])
        assert np.array_equal(got["emitted"], exp["emitted"])
        ok = exp["ok"] == 1
        mtype = sc.flat.contents.materials[int(mi)].type
        for field in ("attenuation", "origin", "direction", "time"):
            sel = ok if mtype != 1 else np.ones_like(ok)      # Metal fills its outputs even when it absorbs
            if mtype == 3:
                continue                                       # DiffuseLight::scatter writes nothing
            assert np.array_equal(got[field][sel], exp[field][sel]), "material %d %s" % (mi, field)
    assert kinds == {0, 1, 2, 3, 4}, "fixture must cover all five materials"


def test_glass_takes_all_three_branches(binding, oracle):
    """reflect, refract and total internal reflection all occur in the fixture"""
    g = load_golden("shade_kat_ref.npz")
    sc = binding.Scene("kat", 128, 72)
    f = sc.flat.contents
    k = [i for i, mi in enumerate(g["mat_index"]) if f.materials[int(mi)].type == 2][0]
    d_in = g["mat_rays"][k][:, 3:6]
    n = g["mat_recs"][k][:, 6:9]
    d_out = g["mat_out"][k]["direction"]
    refl = d_in - 2 * np.einsum("ij,ij->i", d_in, n)[:, None] * n
    is_refl = np.all(np.abs(d_out - refl) <= 1e-5 * (1 + np.abs(refl)), axis=1)
    inside = np.einsum("ij,ij->i", d_in, n) > 0
    assert is_refl.any() and (~is_refl).any() and inside.any() and (~inside).any()


def test_textures_match_reference(binding, oracle_mod, oracle):
    g = load_golden("shade_kat_ref.npz")
    assert made_by_reference_geometry(g["tex_produced_by"])
    sc = binding.Scene("kat", 128, 72)
    kinds = set()
    for k, ti in enumerate(g["tex_index"]):
        got = oracle.texture_value(sc.flat, int(ti), g["tex_uvp"][k])
        assert np.array_equal(got, g["tex_rgb"][k]), "texture %d" % ti
        kinds.add(sc.flat.contents.textures[int(ti)].type)
    assert kinds == {0, 1, 2}


def test_rng_stream_is_rocrand_philox(oracle):
    g = load_golden("misc_ref.npz")
    for (seed, pixel, sample), exp in zip(g["rng_keys"], g["rng_uniforms"]):
        got = oracle.uniforms(int(seed), int(pixel), int(sample), len(exp))
        assert np.array_equal(got, exp)
        assert got.min() > 0.0 and got.max() <= 1.0
    # Random123 known-answer vectors for philox4x32-10
    assert oracle.philox_block([0, 0, 0, 0], [0, 0]).tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox_block([0xffffffff] * 4, [0xffffffff] * 2).tolist() == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox_block([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]).tolist() == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_rotate_y_and_fuzz_as_reference_constructors(binding):
    """the lowering derives sin/cos and the Metal fuzz clamp exactly like RotateY / Metal's constructors"""
    g = load_golden("misc_ref.npz")
    assert made_by_reference_geometry(g["sincos_produced_by"]) and str(g["fuzz_produced_by"]).startswith("libref_shade.so")
    table = {float(a): (float(s), float(c)) for a, (s, c) in zip(g["angles"], g["sincos"])}
    for name, angles in [("cornell", [-18.0, 15.0]), ("kat", [33.0, -50.0]), ("lit_sphere", [0.0, 120.0])]:
        sc = binding.Scene(name, 32, 32)
        f = sc.flat.contents
        rot = [(f.ops[i].a, f.ops[i].b) for i in range(f.n_ops) if f.ops[i].type == 1]
        assert rot == [table[a] for a in angles], name
    sc = binding.Scene("kat", 32, 32)
    f = sc.flat.contents
    fuzz = sorted(f.materials[i].param for i in range(f.n_materials) if f.materials[i].type == 1)
    exp = dict(zip(g["fuzz_in"].tolist(), g["fuzz_out"].tolist()))
    assert fuzz == sorted([exp[np.float32(0.3).item()], exp[10.0]])
