"""NoiseTexture (RT_TEX_NOISE): the Perlin marble the task names but the reference does not have.

PARITY UNPINNED: Texture.cuh has no noise class and no reference fixture covers one, so nothing here is
checked against reference output.  What is checked: the lowered tables, the oracle's restatement against an
independent numpy restatement of "Ray Tracing: The Next Week" ch. 5, the properties gradient noise must have,
and (test_gpu_parity.py) the HIP path against the oracle on the "perlin" scene.
"""
import ctypes as C
import textwrap

import numpy as np
import pytest

from test_scene_api import compile_and_run

RT_TEX_CHECKER, RT_TEX_NOISE = 1, 3
TABLE_BYTES = 256 * 3 * 4 + 3 * 256
F_NOISE = 1 << 15


def noise_textures(sc):
    f = sc.flat.contents
    return [(i, f.textures[i]) for i in range(f.n_textures) if f.textures[i].type == RT_TEX_NOISE]


def table_of(sc, tex):
    f = sc.flat.contents
    raw = np.ctypeslib.as_array(f.image_bytes, shape=(f.n_image_bytes,))
    blob = raw[tex.image_offset:tex.image_offset + TABLE_BYTES].copy()
    grad = blob[:3072].view(np.float32).reshape(256, 3)
    perm = blob[3072:].reshape(3, 256)
    return blob, grad, perm


def np_noise(grad, perm, p):
    """perlin::noise + perlin_interp in float32, the book's operation order"""
    f32 = np.float32
    p = p.astype(np.float32)
    fl = np.floor(p)
    u, v, w = (p - fl).T
    i, j, k = fl.astype(np.int64).T
    uu, vv, ww = (x * x * (f32(3) - f32(2) * x) for x in (u, v, w))
    acc = np.zeros(len(p), np.float32)
    for di in range(2):
        for dj in range(2):
            for dk in range(2):
                h = perm[0][(i + di) & 255] ^ perm[1][(j + dj) & 255] ^ perm[2][(k + dk) & 255]
                c = grad[h]
                wx, wy, wz = u - f32(di), v - f32(dj), w - f32(dk)
                d = c[:, 0] * wx + c[:, 1] * wy + c[:, 2] * wz
                a = f32(di) * uu + f32(1 - di) * (f32(1) - uu)
                b = f32(dj) * vv + f32(1 - dj) * (f32(1) - vv)
                e = f32(dk) * ww + f32(1 - dk) * (f32(1) - ww)
                acc = acc + a * b * e * d
    return acc


def np_turb(grad, perm, p, depth):
    acc = np.zeros(len(p), np.float32)
    weight = np.float32(1)
    p = p.astype(np.float32)
    for _ in range(depth):
        acc = acc + weight * np_noise(grad, perm, p)
        weight = weight * np.float32(0.5)
        p = np.float32(2) * p
    return np.abs(acc)


def oracle_turb(oracle_mod, oracle, blob, p, depth):
    fn = oracle.lib.rto_noise_turb
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    fn.restype = C.c_float
    out = np.zeros(len(p), np.float32)
    for n, q in enumerate(np.ascontiguousarray(p, np.float32)):
        out[n] = fn(blob.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), depth)
    return out


def test_lowered_tables(binding):
    sc = binding.Scene("perlin", 64, 48)
    noises = noise_textures(sc)
    assert len(noises) == 2
    assert binding.lib().rt_flat_features(sc.flat) & F_NOISE
    blobs = []
    for _, t in noises:
        assert t.image_offset % 4 == 0 and t.width == 7
        blob, grad, perm = table_of(sc, t)
        assert np.allclose(np.linalg.norm(grad, axis=1), 1.0, atol=1e-6)
        for axis in range(3):
            assert sorted(perm[axis].tolist()) == list(range(256))
        assert not np.array_equal(perm[0], np.arange(256))
        blobs.append(blob)
    assert not np.array_equal(blobs[0], blobs[1])      # each NoiseTexture draws its own tables from the scene's stream
    assert sorted(t.colour[0] for _, t in noises) == [4.0, 11.0]
    # the same scene built twice is the same scene
    sc2 = binding.Scene("perlin", 64, 48)
    assert np.array_equal(table_of(sc2, noise_textures(sc2)[0][1])[0], blobs[0])
    # reference scenes carry no noise bit
    assert not binding.lib().rt_flat_features(binding.Scene("cornell", 64, 48).flat) & F_NOISE


def test_oracle_equals_numpy_restatement(binding, oracle_mod, oracle):
    sc = binding.Scene("perlin", 64, 48)
    rng = np.random.default_rng(5)
    for ti, t in noise_textures(sc):
        blob, grad, perm = table_of(sc, t)
        p = np.concatenate([rng.uniform(-6, 6, (300, 3)), rng.uniform(-1000, 1000, (200, 3)), rng.normal(0, .01, (50, 3))]).astype(np.float32)
        for depth in (1, 7):
            got = oracle_turb(oracle_mod, oracle, blob, p, depth)
            assert np.array_equal(got, np_turb(grad, perm, p, depth)), depth
        # value(): grey, within [0,1], the marble formula over the turbulence (sinf of two libms: 1 ulp of slack)
        uvp = np.concatenate([np.zeros((len(p), 2), np.float32), p], axis=1)
        rgb = oracle.texture_value(sc.flat, ti, uvp)
        assert np.array_equal(rgb[:, 0], rgb[:, 1]) and np.array_equal(rgb[:, 0], rgb[:, 2])
        assert rgb.min() >= 0.0 and rgb.max() <= 1.0 and rgb.std() > 0.1
        arg = np.float32(t.colour[0]) * p[:, 2] + np.float32(10) * np_turb(grad, perm, p, 7)
        exp = np.float32(0.5) * (np.float32(1) + np.sin(arg.astype(np.float32)))
        assert np.allclose(rgb[:, 0], exp, atol=2e-7 * np.maximum(1, np.abs(arg)).max())


def test_gradient_noise_properties(binding, oracle_mod, oracle):
    sc = binding.Scene("perlin", 64, 48)
    _, t = noise_textures(sc)[0]
    blob, grad, perm = table_of(sc, t)
    rng = np.random.default_rng(6)
    lattice = rng.integers(-300, 300, (200, 3)).astype(np.float32)
    assert np.array_equal(oracle_turb(oracle_mod, oracle, blob, lattice, 7), np.zeros(200, np.float32))   # zero at lattice points, every octave
    p = rng.uniform(-50, 50, (2000, 3)).astype(np.float32)
    one = oracle_turb(oracle_mod, oracle, blob, p, 1)
    assert one.max() <= 1.0 and one.mean() > 0.05          # |noise| stays below 1; not degenerate
    # continuous across cell faces
    q = p.copy()
    q[:, 0] = np.round(q[:, 0])
    lo, hi = q.copy(), q.copy()
    lo[:, 0] -= 1e-3
    hi[:, 0] += 1e-3
    assert np.abs(np_noise(grad, perm, lo) - np_noise(grad, perm, hi)).max() < 1e-2
    # period 256 along each axis (the permutation index is masked)
    shifted = p + np.float32([256, 0, -512])
    assert np.allclose(np_noise(grad, perm, p), np_noise(grad, perm, shifted), atol=2e-4)
    # far outside the int range the cell index is defined (0), not undefined behaviour
    far = np.float32([[3e10, -4e12, 1e20], [-3e30, 0.5, 2.5]])
    assert np.isfinite(oracle_turb(oracle_mod, oracle, blob, far, 7)).all()


def test_bad_noise_tables_are_rejected(binding):
    sc = binding.Scene("perlin", 64, 48)
    f = sc.flat.contents
    ti, t = noise_textures(sc)[0]
    texs = (type(t) * f.n_textures)(*[f.textures[i] for i in range(f.n_textures)])
    bad = type(f)()
    C.pointer(bad)[0] = f
    bad.textures = C.cast(texs, type(f.textures))
    lib = binding.lib()
    assert lib.rt_flat_features(C.byref(bad)) & F_NOISE
    for field, value in (("image_offset", t.image_offset + 2), ("image_offset", f.n_image_bytes - 100), ("width", 0), ("width", 99)):
        keep = getattr(texs[ti], field)
        setattr(texs[ti], field, value)
        assert lib.rt_flat_features(C.byref(bad)) == 0, (field, value)       # 0 = the scene does not validate
        setattr(texs[ti], field, keep)
    assert lib.rt_flat_features(C.byref(bad)) & F_NOISE


def test_noise_texture_in_a_user_scene(tmp_path):
    """NoiseTexture is built like the reference's other stateful nodes (ConstantMedium, BvhNode): scale, then the scene's curandState"""
    body = textwrap.dedent('''
        __device__ inline void marbleScene(Hitable** dList, Hitable** dWorld, Camera** dCamera, int width, int height,
                                           ImageTexture** texture, curandState* randState)
        {
            Texture* pertext = new NoiseTexture(4, randState);
            int i = 0;
            dList[i++] = new Sphere(Vector3(0, -1000, 0), 1000, new Lambert(pertext));
            dList[i++] = new Sphere(Vector3(0, 2, 0), 2, new Lambert(pertext));
            dList[i++] = new Sphere(Vector3(0, 7, 0), 2, new DiffuseLight(new ConstantTexture(Vector3(4, 4, 4))));
            dList[i++] = new XYRect(3, 5, 1, 3, -2, new DiffuseLight(new NoiseTexture(0.1f)));
            *dWorld = new HitableList(dList, i);
            *dCamera = new Camera(Vector3(26, 3, 6), Vector3(0, 2, 0), Vector3(0, 1, 0), 20.f, float(width) / float(height), 0.f, 10.f, 0.f, 1.f);
        }
    ''')
    n_objects, n_prims, n_media, n_world = compile_and_run(tmp_path, body, "marbleScene(dList, &world, &cam, 400, 225, &img, &rs);")
    assert (n_prims, n_media, n_world) == (4, 0, 4)
