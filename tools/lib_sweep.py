"""Rate of the wavefront workloads under several builds of the library (tools/diag/librt_<name>.so, made with other
tuning constants): python tools/lib_sweep.py name [name ...]; "base" = the shipped library."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for wl, extra in (("final", ["--width", "1920", "--height", "1080", "--spp", "64"]), ("tri_mesh", ["--spp", "32"])):
    for v in sys.argv[1:]:
        env = dict(os.environ)
        if v != "base":
            env["RT_AMD_LIB"] = os.path.join(ROOT, "tools/diag/librt_%s.so" % v)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--workload", wl, "--steps", "2", "--warmup", "1"] + extra,
                             env=env, capture_output=True, text=True, timeout=200)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            print(wl, v, "%.1f Mrays/s %.1f ms" % (j["value"], j["ms_per_step"]), flush=True)
        except Exception:
            print(wl, v, "failed", out.stderr[-300:], flush=True)
