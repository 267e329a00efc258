"""Quick GPU-vs-oracle comparison (development aid; the real parity tests live in tests/)."""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m
B = load('rtb', os.path.join(ROOT, 'cuda-raytracer_amd/binding.py'))
O = load('rto', os.path.join(ROOT, 'oracle/oracle.py'))
import torch
orc = O.Oracle()
ctx = B.Context(0)
ctx.set_integrator(int(os.environ.get('RT_INTEGRATOR', '0')))
rng = np.random.default_rng(1)
names = sys.argv[1:] or ["cornell", "kat", "random_spheres", "lit_sphere", "bvh_boxes", "cornell_cluster", "final", "cornell_smoke"]
for name in names:
    W, H = 96, 64
    sc = B.Scene(name, W, H)
    f = sc.flat.contents
    ctx.upload(sc)
    cam = f.camera
    n = 50000
    org = np.array(cam.origin[:], dtype=np.float32)
    st = rng.random((n, 2)).astype(np.float32)
    d = (np.array(cam.lower_left[:], dtype=np.float32) + st[:, :1] * np.array(cam.horizontal[:], dtype=np.float32)
         + st[:, 1:] * np.array(cam.vertical[:], dtype=np.float32) - org)
    rays = np.concatenate([np.tile(org, (n, 1)), d, rng.random((n, 1)).astype(np.float32)], axis=1).astype(np.float32)
    # second half: secondary-like rays from hit points in random directions
    a0 = orc.trace_rays(sc.flat, rays, 0.001, 3.4e38)
    h = a0['hit'] == 1
    rays2 = rays.copy()
    rays2[h, 0:3] = a0['p'][h]
    rays2[:, 3:6] = rng.normal(size=(n, 3)).astype(np.float32)
    allr = np.concatenate([rays, rays2])
    a = orc.trace_rays(sc.flat, allr, 0.001, 3.4e38)
    g = ctx.trace_rays(torch.from_numpy(allr).cuda(), 0.001, 3.4e38)
    hit = a['hit'] == 1
    msg = [name, "hits %d/%d" % (hit.sum(), len(allr))]
    for k in ["hit", "prim", "t", "u", "v", "p", "n", "material"]:
        x, y = (a[k], g[k]) if k == "hit" else (a[k][hit], g[k][hit & (g['hit'] == 1)] if (g['hit'] == a['hit']).all() else g[k][hit])
        nd = int(np.count_nonzero(np.any(np.atleast_2d(x.reshape(len(x), -1) != y.reshape(len(y), -1)), axis=1)))
        msg.append("%s:%d" % (k, nd))
    print(" ".join(msg), flush=True)
    # primary map
    pg, tg = ctx.trace_primary(W, H)
    po, to = orc.trace_primary(sc.flat, W, H)
    print("   primary: prim mismatches %d, t mismatches %d" % ((pg != po).sum(), (tg != to).sum()), flush=True)
    # small render
    spp = 8
    bg = sc.background
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device='cuda')
    torch.cuda.synchronize()
    ctx.reset_counters()
    t0 = time.time()
    ctx.render_accumulate(acc, W, H, 0, spp, 419, 100, bg)
    c = ctx.counters()
    t1 = time.time()
    ao, st_ = orc.render(sc.flat, W, H, 0, spp, 419, 100, bg, threads=16)
    ag = acc.cpu().numpy().reshape(H, W, 3)
    diff = np.abs(ag - ao)
    bad = np.count_nonzero(np.any(diff > 1e-4 * (1 + np.abs(ao)), axis=2))
    print("   render: gpu rays %d oracle rays %d | pixels differing %d/%d, max abs diff %.3g, mean gpu %.5f oracle %.5f | kernel %.2f ms"
          % (c.rays, st_.rays, bad, W * H, diff.max(), ag.mean(), ao.mean(), c.last_render_ms), flush=True)
