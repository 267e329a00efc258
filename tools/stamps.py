"""Section shares of the render loop from the diagnostic (-DRT_STAMPS) build: RT_AMD_LIB=tools/diag/librt_amd_stamps.so"""
import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RT_AMD_LIB", os.path.join(ROOT, "tools/diag/librt_amd_stamps.so"))
spec = importlib.util.spec_from_file_location("rtb", os.path.join(ROOT, "cuda-raytracer_amd/binding.py")); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
import torch
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W, H, spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024, int(sys.argv[3]) if len(sys.argv) > 3 else 1024, int(sys.argv[4]) if len(sys.argv) > 4 else 64
param = int(sys.argv[5]) if len(sys.argv) > 5 else 0
sc = B.Scene(name, W, H, param)
ctx = B.Context(0)
ctx.upload(sc)
ctx.set_integrator(int(os.environ.get('RT_INTEGRATOR', '0')))
acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
B.lib().rt_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
ctx.render_accumulate(acc, W, H, 0, 2, 419, 100, sc.background)
ctx.synchronize()
base = (C.c_ulonglong * 12)()
B.lib().rt_debug_stamps(ctx.handle, base)
ctx.reset_counters()
ctx.render_accumulate(acc, W, H, 2, 2 + spp, 419, 100, sc.background)
c = ctx.counters()
out = (C.c_ulonglong * 12)()
B.lib().rt_debug_stamps(ctx.handle, out)
for q in range(6, 10):
    out[q] -= base[q]      # the device-global loop counters are not reset with the context's
if os.environ.get("RT_INTEGRATOR") == "2" or (sc.flat.contents.n_bvh_nodes > 0 and os.environ.get("RT_INTEGRATOR") != "1"):
    # the intersect kernel's eight counters live in the context's counter block, which reset_counters() has just zeroed: they
    # are the measured call's alone (subtracting the warm-up's reading, as this script did until round 3, made every
    # per-ray figure too small by warm-up samples / samples)
    d = [float(out[q]) for q in (0, 1, 2, 3, 4, 5, 10, 11)]
    print("%s %dx%d@%d wavefront: %.1f ms, %.1f Mrays/s, rays %d" % (name, W, H, spp, c.last_render_ms, c.rays / c.last_render_ms / 1e3, c.rays))
    print("  intersect: %.1f trips per ray-lane (trips*64/rays), busy lanes/trip %.1f, node lanes %.1f, prim lanes %.1f"
          % (d[0] * 64 / c.rays, d[1] / d[0], d[2] / d[0], d[3] / d[0]))
    if os.environ.get("RT_STAMPS_REFILL"):      # library built with `make stamps STAMPS_EXTRA=-DRT_STAMPS_REFILL`: words 4, 5 = cycles inside refills, refills
        print("  refills per trip %.3f, %.0f cycles each, %.1f %% of the kernel's cycles; trips with a primitive step %.3f (%.1f lanes each); cycles/trip %.0f"
              % (d[5] / d[0], d[4] / max(1.0, d[5]), 100.0 * d[4] / max(1.0, d[7]), d[6] / d[0], d[3] / max(1.0, d[6]), d[7] / d[0]))
    else:
        print("  trips with a node step %.3f (%.1f lanes each), with a primitive step %.3f (%.1f lanes each); stalled lanes/trip %.1f; cycles/trip %.0f"
              % (d[5] / d[0], d[2] / max(1.0, d[5]), d[6] / d[0], d[3] / max(1.0, d[6]), d[4] / d[0], d[7] / d[0]))
    print("  rows fetched per ray: %.2f four-wide nodes (128-B rows) + %.2f leaf primitives (64-B rows)" % (d[2] / c.rays, d[3] / c.rays))
    if hasattr(B.lib(), "rt_debug_sections") and os.environ.get("RT_STAMPS_RF3"):      # variant built with -DRT_STAMPS -DRT_STAMPS_REFILL -DRT_STAMPS_RF3
        sec = (C.c_ulonglong * 16)()
        B.lib().rt_debug_sections.argtypes = [C.c_void_p, C.c_void_p]
        B.lib().rt_debug_sections(ctx.handle, sec)
        n_rf = max(1.0, d[5])
        print("  a refill of the one-kernel form: bounce %.0f cycles (%.1f lanes), claims %.0f, plain records + group entry %.0f (%.1f rays started)"
              % (sec[0] / n_rf, sec[3] / n_rf, sec[1] / n_rf, sec[2] / n_rf, sec[4] / n_rf))
        kinds = ["lambert", "metal", "glass", "light", "isotropic"]
        print("  material kinds in a batch: " + ", ".join("%s in %.0f %% of the batches (%.1f lanes when present)" % (k, 100.0 * sec[10 + i] / n_rf, sec[5 + i] / max(1.0, sec[10 + i])) for i, k in enumerate(kinds)))
    if hasattr(B.lib(), "rt_debug_sections") and os.environ.get("RT_WF_KERNEL", "1") == "2":
        sec = (C.c_ulonglong * 16)()
        B.lib().rt_debug_sections.argtypes = [C.c_void_p, C.c_void_p]
        B.lib().rt_debug_sections(ctx.handle, sec)
        names = ["refill", "trip top + loads", "pipeline", "node step", "queueing + stack", "prim step", "tail", "spins"]
        tot = float(sum(sec[:8])) or 1.0
        print("  kernel 2 cycles by section: " + ", ".join("%s %.1f %%" % (n, 100.0 * sec[i] / tot) for i, n in enumerate(names)))
        print("  per trip: " + ", ".join("%s %.0f" % (n, sec[i] / d[0]) for i, n in enumerate(names)) + "; spins per trip %.3f, claims %d" % (sec[8] / d[0], sec[9]))
    sys.exit(0)
names = ["regen", "closest_hit", "rebuild", "ball", "tail", "scatter+rest"]
tot = sum(out[:6]) or 1
print("%s %dx%d@%d: %.1f ms, %.1f Mrays/s, rays %d" % (name, W, H, spp, c.last_render_ms, c.rays / c.last_render_ms / 1e3, c.rays))
for n, v in zip(names, out[:6]):
    print("  %-14s %5.1f %%   %8.0f cycles per wave-step (rays/64)" % (n, 100.0 * v / tot, v / (c.rays / 64.0)))
if out[6]:
    r = float(c.rays)
    print("  BVH: %.1f node visits and %.1f leaf tests per ray" % (out[6] / r, out[7] / r))
    print("       lane efficiency of the node loop %.2f, of the leaf loop %.2f" % (out[6] / max(1, out[8]), out[7] / max(1, out[9])))
