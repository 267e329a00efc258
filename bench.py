#!/usr/bin/env python3
"""bench.py — Mrays/s of the path-tracing hot path on MI355X.

A step = one full render of the workload: every rank renders its own range of sample indices for every
pixel of the frame into an fp32 accumulation buffer resident in HBM (C ABI rt_render_accumulate),
the buffers are summed onto rank 0 (one RCCL reduce; nothing to do at N=1) and rank 0 resolves
(mean + sqrt gamma).  Two ways to grow with N (--scaling; the default follows the BASELINE.json config):
  weak    every GPU renders `spp` samples, N GPUs = N x spp  (cornell, cornell_smoke: "... on one MI355X")
  strong  the workload's `spp` is the total, rank r renders sample_range(r, N, spp)
          (tri_mesh: "spp-sharded at 1/2/4/8 GPUs"; final: "4096 spp across 8xMI355X")

Default workload = BASELINE.json configs[1]: Cornell box, 1024x1024, 1024 spp, depth 100.
A ray = one closest-hit query = one iteration of colour()'s loop (reference main.cu:53-56).

  python bench.py [--gpus N --steps K --warmup W] [--workload cornell|cornell_smoke|tri_mesh|final]
                  [--width --height --spp] [--scaling weak|strong] [--no-cpu-baseline]
One process per GPU.  Under torch.distributed.run (WORLD_SIZE set) this file is a rank; started plainly with --gpus N > 1
it starts that launcher itself as a CHILD process, before anything here touches the GPU, and passes on the child's JSON
line and exit code (the reference's own two-GPU mode is two host threads of one process, main.cu:652-699).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

# algorithmic flop weights per unit of work, SURVEY.md 8(d) / DESIGN.md "Roofline"
F_RECT, F_SPHERE, F_TRI, F_NODE, F_ROT, F_TRANS = 15.0, 30.0, 105.0, 50.0, 12.0, 3.0
F_SHADE, F_SPHERE_ITER, F_PHILOX = 60.0, 20.0, 100.0
PEAK_VALU_TFLOPS = 157.3   # MI355X fp32 vector peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0      # HBM3E spec peak
N_CUS = 256
CLOCK_GHZ = 2.4
TA_LANES_PER_CLOCK = 1.0       # divergent 16-byte lane loads a CU's address unit takes per clock, measured (profiles/r02_gather_sizes.jsonl)
L2_MISS_REQUESTS_PEAK = 58.0   # G random 64-B / 128-B row requests per second that leave the L2s, measured (profiles/r02_gather_rates.jsonl)

WORKLOADS = {
    # name: (scene, param, width, height, spp)
    "cornell": ("cornell", 0, 1024, 1024, 1024),
    "cornell_smoke": ("cornell_smoke", 0, 1024, 1024, 2048),
    "tri_mesh": ("tri_mesh", 1000000, 1920, 1080, 256),
    "final": ("final", 100, 3840, 2160, 4096),
}
# how each BASELINE.json config grows with the number of GPUs (module docstring)
DEFAULT_SCALING = {"cornell": "weak", "cornell_smoke": "weak", "tri_mesh": "strong", "final": "strong"}


def step_sample_range(scaling, k, rank, world, spp):
    """[begin, end) of global sample indices rank `rank` of `world` renders in step k; every step renders fresh indices.
    weak: every rank `spp` samples; strong: `spp` samples in all, cut by shard.sample_range."""
    if scaling == "weak":
        s0 = (k * world + rank) * spp
        return s0, s0 + spp
    shard = entry._load("rt_shard", os.path.join(entry.PKG, "shard.py"))
    return shard.sample_range(rank, world, spp, first=k * spp)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run torch.distributed.run as a child (no exec: this process has not
    touched the GPU and never will), pass its one JSON line to stdout, everything else to stderr, and return its code"""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in p.stdout:
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return p.wait()


def sources_sha16():
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "cuda-raytracer_amd", "csrc", "*"))):
        if f.endswith((".h", ".hip", ".cpp")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def flops_per_ray(st):
    """algorithmic fp32 ops per ray from the instrumented CPU traversal of the same workload"""
    rays = max(1, st.rays)
    f = (F_RECT * st.rect_tests + F_SPHERE * st.sphere_tests + F_TRI * st.tri_tests + F_NODE * st.nodes_visited
         + F_ROT * st.rot_ops + F_TRANS * st.trans_ops + F_SPHERE_ITER * st.sphere_iters + F_PHILOX * st.philox_blocks) / rays
    return f + F_SHADE


def bytes_per_ray(st, width, height, spp, scene_bytes, wavefront=False):
    """algorithmic bytes per ray: 32 B per BVH node box tested and 64 B per BVH-leaf primitive row
    (counted by the instrumented CPU traversal of the same BVH), plus the accumulation buffer read+write
    (24 B/pixel per launch) and one read of the scene tables amortised over the launch's rays; the wavefront
    form adds its path-state traffic: ray 32 B written + read, hit 16 B written + read, attenuation / emission
    32 B read + written = 160 B per ray."""
    rays = max(1, st.rays)
    per_ray = 32.0 * st.nodes_visited / rays
    if st.nodes_visited:
        per_ray += 64.0 * st.prim_tests / rays
    if wavefront:
        per_ray += 160.0
    samples = max(1, st.samples)
    rays_per_launch = (rays / samples) * width * height * spp
    return per_ray + (24.0 * width * height + scene_bytes) / rays_per_launch


def cpu_baseline(binding, oracle_mod, scene_name, param, background, use_bvh, budget_s=12.0):
    """the oracle (a scalar port of the reference, see oracle/rt_oracle.c) on the host cores, on a bounded
    sample of the same workload: the same scene and camera at 128x128, 4 spp per pass, for ~budget_s."""
    W = H = 128
    scene = binding.Scene(scene_name, W, H, param)
    orc = oracle_mod.Oracle()
    # all the host cores this job may use: the scheduler's share where it sets one, and never more than the 16 a one-GPU
    # box grants (its os.cpu_count() shows the whole machine); the line states what was used and what was visible
    threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    import numpy as np
    acc = np.zeros((H, W, 3), dtype=np.float32)
    tot = None
    t0 = time.perf_counter()
    s = 0
    while True:
        _, st = orc.render(scene.flat, W, H, s, s + 4, 419, 100, background, use_bvh=use_bvh, threads=threads, accum=acc)
        s += 4
        if tot is None:
            tot = st
        else:
            for name, _ in st._fields_:
                setattr(tot, name, getattr(tot, name) + getattr(st, name))
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    # one-core rate on a short slice
    t1 = time.perf_counter()
    _, st1 = orc.render(scene.flat, W, H, 0, 1, 419, 100, background, use_bvh=use_bvh, threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": tot.rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": "%s at %dx%d, %d spp, depth 100, %d threads, %.1f s" % (scene_name, W, H, s, threads, dt),
        "value_1core": st1.rays / dt1 / 1e6, "cores_visible": os.cpu_count(), "cores_cap": 16,
    }, tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--param", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--integrator", default="auto", choices=["auto", "megakernel", "wavefront"])
    ap.add_argument("--no-extras", action="store_true", help="skip the other configs' lines of the default run (the last one takes a minute)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"], help="auto: the workload's own (DEFAULT_SCALING)")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL; must be set before the first HIP call
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible")
    # RT_BENCH_BACKEND=gloo is a rehearsal of the N > 1 flow on a box with fewer GPUs than ranks (ranks share
    # the cards, collectives go through host copies); its rate means nothing and the line says so
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # RT_BENCH_FORCE_DIST=1: at N = 1 too, create the process group and run every collective of the N > 1 flow (reduce onto
    # rank 0, the two all-reduces, the barriers) -- with one rank they move nothing, but the RCCL calls, their ordering
    # against the render stream and the teardown execute on the hardware at hand (tests/test_z_bench_gpu.py)
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if "MASTER_ADDR" not in os.environ:       # forced at N = 1 without a launcher
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def coll(fn, t, **kw):
        if backend == "gloo":
            c = t.cpu()
            fn(c, **kw)
            t.copy_(c)
        else:
            fn(t, **kw)

    b = entry.load_binding()

    def run(workload, W, H, spp, param, steps, warmup, want_cpu):
        """time `steps` renders of one workload; returns the result dict (rank 0) or None"""
        scene_name, dparam, dW, dH, dspp = WORKLOADS[workload]
        W, H, spp = W or dW, H or dH, spp or dspp
        scaling = DEFAULT_SCALING[workload] if args.scaling == "auto" else args.scaling
        if scaling == "strong" and spp < world:
            sys.exit("bench.py: %d samples cannot be cut over %d GPUs" % (spp, world))
        spp_total = spp * world if scaling == "weak" else spp     # samples per pixel of the image rank 0 resolves
        param = dparam if param < 0 else param
        scene = b.Scene(scene_name, W, H, param)
        flat = scene.flat.contents
        background = scene.background
        has_bvh = flat.n_bvh_nodes > 0
        scene_bytes = (flat.n_objects * 128 + flat.n_ops * 16 + flat.n_materials * 16 + flat.n_textures * 40
                       + flat.n_image_bytes + flat.n_bvh_nodes * 32 + flat.n_bvh_items * 64)
        # one stream for everything: torch's fills, the library's kernels and the collective's hand-over.  (A null
        # stream handle would make the library create a stream of its own, unordered against torch's.)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        assert stream.cuda_stream != 0
        ctx = b.Context(dev_index, stream.cuda_stream)
        ctx.upload(scene)
        ctx.set_integrator({"auto": 0, "megakernel": 1, "wavefront": 2}[args.integrator])
        accum = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
        rgb = None

        def step(k):
            nonlocal rgb
            accum.zero_()
            s0, s1 = step_sample_range(scaling, k, rank, world, spp)
            ctx.render_accumulate(accum, W, H, s0, s1, 419, 100, background)
            if use_dist:
                coll(dist.reduce, accum, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                rgb = ctx.resolve(accum, W, H, spp_total, gamma=True)

        def fence():
            torch.cuda.synchronize(dev)
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize(dev)

        for k in range(warmup):
            step(k)
        fence()
        ctx.reset_counters()
        t0 = time.perf_counter()
        for k in range(steps):
            step(warmup + k)
        fence()
        dt = time.perf_counter() - t0
        c = ctx.counters()
        wavefront = c.last_integrator == 2            # the form the library ran (rt_counters.last_integrator)

        tot = torch.tensor([float(c.rays), float(c.samples)], dtype=torch.float64, device=dev)
        tmax = torch.tensor([dt, c.render_ms / max(1, c.render_launches)], dtype=torch.float64, device=dev)
        if use_dist:
            coll(dist.all_reduce, tot, op=dist.ReduceOp.SUM)
            coll(dist.all_reduce, tmax, op=dist.ReduceOp.MAX)
        rays, samples = tot.tolist()
        dt, kernel_ms = tmax.tolist()
        ctx.close()
        if rank != 0:
            return None
        finite = bool(torch.isfinite(rgb).all().item())
        mean_rgb = float(rgb.mean().item())
        out = {
            "metric": "Mrays/s", "value": rays / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s_%dx%d_%dspp" % (workload, W, H, spp), "scene": scene_name,
                       "width": W, "height": H, "spp_per_gpu": spp if scaling == "weak" else spp / world, "spp_total": spp_total, "max_depth": 100,
                       "background": "sky" if background else "black", "bvh": has_bvh,
                       "integrator": "wavefront" if wavefront else "megakernel",
                       "wf_kernel": int(c.wf_kernel) if wavefront else None, "wf_pools": int(c.wf_pools) if wavefront else None,
                       "parallelism": ("spp-shard x%d (%s) + %s" % (world, scaling, "RCCL reduce" if backend == "nccl" else "gloo reduce (REHEARSAL, ranks share GPUs)")) if world > 1 else ("single GPU" + (" (collectives executed with one rank, %s)" % backend if use_dist else ""))},
            "samples_per_s": samples / dt, "rays_per_sample": rays / max(1.0, samples),
            "image_mean": mean_rgb, "image_finite": finite,
        }
        base_stats = None
        # rank 0 only; at N > 1 the other ranks wait at the next collective while rank 0 counts the algorithmic work
        o = entry.load_oracle() if not args.no_cpu_baseline else None
        if o is not None and want_cpu and world == 1:
            out["cpu_baseline"], base_stats = cpu_baseline(b, o, scene_name, param, background, has_bvh)
        elif o is not None:
            # no timed baseline for this line: a short instrumented oracle pass only counts the algorithmic work
            small = b.Scene(scene_name, 128, 72, param)
            _, base_stats = o.Oracle().render(small.flat, 128, 72, 0, 2, 419, 100, background, use_bvh=has_bvh, threads=16)
            out["cpu_baseline"] = None
        else:
            out["cpu_baseline"] = None
        # roofline of the dominant kernel, duration from HIP events recorded on its own stream inside the library.
        # megakernel: render_kernel is the launch, bound by VALU issue (its tables sit in the scalar cache).
        # wavefront: ONE persistent kernel per call (wf_path_kernel, rt_wavefront4.h), so the event pair is that kernel's
        # duration and agrees with the trace (profiles/r03_trace_tri_kernel_stats.csv); with RT_WF_KERNEL=1..3 it spans the
        # intersect/logic loop of the pools.  Three limits are reported under "limits" (DESIGN.md section 6): VALU issue of
        # the kernel (PMC pass: saturated in the one-kernel form), the
        # rate at which a CU's address unit takes divergent 16-byte lane loads (tools/ubench/gather.hip with an
        # L2-resident table: about one lane per clock per CU), and the rate at which L2 misses leave the L2s (same
        # ubench, 110 MB and 1 GB tables: 58 G row requests/s chip-wide).  The line's own bound is the tightest one,
        # VALU issue, stated like the megakernel's: algorithmic flops per ray against the fp32 vector
        # peak.  The algorithmic HBM-bytes fraction SURVEY 8(d) defines is carried under "hbm_algorithmic"; the BVH
        # sits in the Infinity Cache and HBM bandwidth does not bind.
        rays_per_launch = rays / world / max(1, c.render_launches)
        one_kernel = wavefront and int(c.wf_kernel) == 4
        kname = ("wf_path_kernel" if one_kernel else "wf_intersect_kernel + wf_logic_kernel loop") if wavefront else "render_kernel"
        pm = None
        tpath = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(tpath):
            try:
                table = json.load(open(tpath))
                pm = table.get(out["config"]["workload"])
                if pm is None:      # counted on a smaller frame of the same scene (the final scene: 1920x1080 x 64 spp); per-ray figures carry over
                    pm = next((e for e in table.values() if e.get("stands_for") == workload), None)
            except Exception:
                pm = None
        roof = {"bound": "valu", "kernel": kname, "kernel_ms": kernel_ms, "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s",
                "achieved": None, "frac": None, "traffic": None}
        if base_stats is not None and kernel_ms > 0:
            fpr = flops_per_ray(base_stats)
            bpr = bytes_per_ray(base_stats, W, H, spp, scene_bytes, wavefront)
            ach_f = rays_per_launch * fpr / (kernel_ms * 1e-3) / 1e12
            ach_b = rays_per_launch * bpr / (kernel_ms * 1e-3) / 1e9
            roof.update({"achieved": ach_f, "frac": ach_f / PEAK_VALU_TFLOPS, "flops_per_ray": fpr})
            if wavefront:
                rows = (pm or {}).get("rows_per_ray")
                hit = (pm or {}).get("l2_hit_rate")
                limits = {"valu_issue_intersect": {"busy": (pm or {}).get("valu_busy"), "lane_efficiency": (pm or {}).get("lane_efficiency"),
                                                   "source": "profiles/pmc_summary.json (rocprofv3 --pmc pass of this command)"}}
                if rows:
                    rays_s = rays_per_launch / (kernel_ms * 1e-3)
                    # 16-byte lane loads and stores per ray: 7 per four-wide node row, 4 per leaf row, and in the two-kernel
                    # forms the path state (refill 2 + hit 1 in the intersect kernel; 5 loads + 4 stores in the logic kernel)
                    lane_ops = 7.0 * rows["node4_128B"] + 4.0 * rows["leaf_64B"] + (0.0 if one_kernel else 12.0)
                    per_clk_cu = rays_s * lane_ops / (N_CUS * CLOCK_GHZ * 1e9)
                    limits["vector_memory_address_rate"] = {
                        "lane_16B_accesses_per_ray": lane_ops, "achieved_per_clock_per_CU": per_clk_cu, "peak_per_clock_per_CU": TA_LANES_PER_CLOCK,
                        "frac": per_clk_cu / TA_LANES_PER_CLOCK,
                        "source": "profiles/r02_gather_sizes.jsonl: L2-resident random rows, 76 G x 7 and 174 G x 4 lane loads/s = 0.87 and 1.13 per clock per CU"}
                    if hit is not None:
                        per_ray = rows["node4_128B"] + rows["leaf_64B"] + rows["state_64B"]
                        ach = rays_s * per_ray * (1.0 - hit) / 1e9
                        limits["l2_miss_requests"] = {"achieved_Grequests_per_s": ach, "peak_Grequests_per_s": L2_MISS_REQUESTS_PEAK,
                                                      "frac": ach / L2_MISS_REQUESTS_PEAK, "row_requests_per_ray": per_ray, "l2_hit_rate": hit,
                                                      "source": "profiles/r02_gather_rates.jsonl (110 MB and 1 GB tables)"}
                roof["limits"] = limits
                roof["hbm_algorithmic"] = {"bytes_per_ray": bpr, "achieved_GBs": ach_b, "peak_GBs": PEAK_HBM_GBS, "frac": ach_b / PEAK_HBM_GBS,
                                           "note": "SURVEY 8(d) formula (every node and leaf row counted as if fetched from HBM); a frac above 1 means the "
                                                   "L2s and the Infinity Cache absorbed the difference -- see fabric_measured for the bytes that left the L2s"}
            else:
                roof.update({"bytes_per_ray": bpr, "hbm_peak_GBs": PEAK_HBM_GBS, "hbm_achieved_GBs": ach_b, "hbm_frac": ach_b / PEAK_HBM_GBS})
        # counters cannot be read in-process: the per-call HBM bytes and the PMC-derived shares come from the
        # committed rocprofv3 passes of this same command (tools/prof_round.sh -> profiles/pmc_summary.json), which
        # carry the commit they were taken at: after a kernel change they describe THAT build until the passes are re-run
        if pm:
            kernel_s = kernel_ms * 1e-3
            if pm.get("traffic_bytes_per_ray") and kernel_s > 0:
                # measured bytes between the L2s and the fabric (FETCH_SIZE + WRITE_SIZE, corrected as tools/make_profiles.py says),
                # per ray from the counter passes x this run's rays per launch, over this run's kernel time
                roof["traffic"] = pm["traffic_bytes_per_ray"] * rays_per_launch
                fab = roof["traffic"] / kernel_s / 1e9
                roof["fabric_measured"] = {"bytes_per_ray": pm["traffic_bytes_per_ray"], "achieved_GBs": fab, "peak_GBs": PEAK_HBM_GBS, "frac": fab / PEAK_HBM_GBS,
                                           "counted_on": pm.get("counted_on")}
            if pm.get("lane_instructions_per_ray") and kernel_s > 0:
                # lanes x VALU instructions per second against what 1024 SIMDs x 16 lanes issue per clock: the share of the
                # chip's issue slots that carried a live lane (= VALU-busy share x lane efficiency)
                roof["issue_ceiling_frac"] = pm["lane_instructions_per_ray"] * rays_per_launch / kernel_s / (N_CUS * 4 * 16 * CLOCK_GHZ * 1e9)
            roof["pmc"] = {k: v for k, v in pm.items() if k not in ("traffic_bytes", "traffic_uncorrected_bytes", "rays")}
            # do the committed counter passes describe the sources this library was built from?  (tools/make_profiles.py hashes
            # cuda-raytracer_amd/csrc at the time of the passes; false = re-run tools/prof_round.sh)
            roof["pmc_matches_sources"] = pm.get("sources_sha16") == sources_sha16() if pm.get("sources_sha16") else None
        out["roofline"] = roof
        return out

    out = run(args.workload, args.width, args.height, args.spp, args.param, args.steps, args.warmup, True)
    if args.workload == "cornell" and not args.no_extras and not (args.width or args.height or args.spp):
        # The north_star target scene beside the headline one: BASELINE.json configs[3] as it stands (1M triangles,
        # 1920x1080, 256 spp: ~1.3 s per step on one GPU), STRONG-scaled when N > 1 (the 256 samples cut over the ranks),
        # which is what "spp-sharded at 1/2/4/8 GPUs" and the >= 1 Grays/s node-wide target are stated on.
        # ... and the other GPU configs at their full size, one step each, so that every BASELINE config has a line
        # timed by whoever runs this file: configs[4] the final scene with 100 x 100 boxes at 3840x2160 x 4096 spp
        # ("across 8xMI355X": strong as well; about a minute on one GPU, no warm-up step) and, on one GPU only,
        # configs[2] Cornell smoke (0.5 s).  Every rank takes part in every line.
        keys = ("value", "unit", "n_gpus", "scaling", "steps", "warmup", "ms_per_step", "config", "samples_per_s", "rays_per_sample", "roofline")
        extras = [("tri_mesh", 1, 1)] + ([("cornell_smoke", 1, 1)] if world == 1 else []) + [("final", 1, 0)]
        also = []
        for wl, steps, warm in extras:
            extra = run(wl, 0, 0, 0, -1, steps, warm, False)
            if rank == 0:
                also.append({k: extra[k] for k in keys})
                print("bench.py: %s done, %.0f Mrays/s" % (wl, extra["value"]), file=sys.stderr, flush=True)
        if rank == 0:
            out["also"] = also
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
