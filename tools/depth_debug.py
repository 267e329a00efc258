"""development aid: GPU vs oracle render at small depth limits, to localise a divergence"""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m
B = load('rtb', os.path.join(ROOT, 'cuda-raytracer_amd/binding.py'))
O = load('rto', os.path.join(ROOT, 'oracle/oracle.py'))
import torch
orc = O.Oracle(); ctx = B.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W, H = 96, 64
sc = B.Scene(name, W, H); ctx.upload(sc)
for depth in (1, 2, 3, 4, 6, 100):
    for spp in (1, 4):
        acc = torch.zeros(H * W * 3, dtype=torch.float32, device='cuda'); torch.cuda.synchronize()
        ctx.reset_counters()
        ctx.render_accumulate(acc, W, H, 0, spp, 419, depth, sc.background)
        c = ctx.counters()
        ao, st = orc.render(sc.flat, W, H, 0, spp, 419, depth, sc.background, threads=8)
        ag = acc.cpu().numpy().reshape(H, W, 3)
        bad = np.any(np.abs(ag - ao) > 1e-4 * (1 + np.abs(ao)), axis=2)
        ys, xs = np.nonzero(bad)
        ex = "" if not len(ys) else " e.g. (%d,%d) gpu %s oracle %s" % (xs[0], ys[0], ag[ys[0], xs[0]], ao[ys[0], xs[0]])
        print("depth %3d spp %d: rays gpu %d oracle %d, bad pixels %d%s" % (depth, spp, c.rays, st.rays, bad.sum(), ex), flush=True)
