import os, sys, subprocess, json
for wl, extra in (("final", ["--width", "1920", "--height", "1080", "--spp", "64"]), ("tri_mesh", ["--spp", "32"])):
    for v in sys.argv[1:]:
        env = dict(os.environ, RT_WF_REFILL=v)
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--workload", wl, "--steps", "2", "--warmup", "1"] + extra,
                             env=env, capture_output=True, text=True, timeout=200)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1]); print(wl, v, "%.1f Mrays/s %.1f ms" % (j["value"], j["ms_per_step"]), flush=True)
        except Exception as e:
            print(wl, v, "failed", out.stderr[-300:], flush=True)
