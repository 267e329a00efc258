"""bench.py end to end on a small frame: the JSON line's contract, and the image of its last step against a direct
render of the same sample range (torch's fills, the library's kernels and the resolve must be ordered on one stream)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


@pytest.mark.parametrize("workload,extra", [("cornell", []), ("tri_mesh", ["--param", "3000"])])
def test_bench_line_and_image(binding, gpu_ctx, workload, extra):
    import torch
    W, H, spp, steps, warmup = 64, 48, 8, 3, 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--width", str(W), "--height", str(H),
                        "--spp", str(spp), "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-extras"] + extra,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j
    assert j["n_gpus"] == 1 and j["steps"] == steps and j["warmup"] == warmup and j["value"] > 0 and j["image_finite"]
    assert j["config"]["workload"] == "%s_%dx%d_%dspp" % (workload, W, H, spp)
    # the last step rendered samples [(warmup + steps - 1) * spp, + spp) into a zeroed buffer
    name, param = ("cornell", 0) if workload == "cornell" else ("tri_mesh", 3000)
    sc = binding.Scene(name, W, H, param)
    gpu_ctx.upload(sc)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    s0 = (warmup + steps - 1) * spp
    gpu_ctx.render_accumulate(acc, W, H, s0, s0 + spp, 419, 100, sc.background)
    gpu_ctx.synchronize()
    rgb = gpu_ctx.resolve(acc, W, H, spp, gamma=True)
    gpu_ctx.synchronize()
    assert abs(float(rgb.mean()) - j["image_mean"]) <= 1e-5 * max(1.0, abs(j["image_mean"]))


def test_two_rank_flow_rehearsal():
    """bench.py --gpus 2 under torch.distributed.run with RT_BENCH_BACKEND=gloo: both ranks share this box's GPU and the
    collectives go through host copies, so only the flow is checked (sample shards, reduce onto rank 0, one JSON line
    from rank 0, an image that is the mean over both ranks' samples)."""
    W, H, spp = 64, 48, 8
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--width", str(W), "--height", str(H), "--spp", str(spp), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    # the plain command line, as the round-end driver may give it: no outer launcher, bench.py starts the ranks itself
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common,
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]       # ONE JSON line and nothing else on stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["image_finite"] and "REHEARSAL" in j["config"]["parallelism"]
    # ... and under the launcher (the contract's form): same line
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common,
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    jl = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert jl["n_gpus"] == 2 and abs(jl["image_mean"] - j["image_mean"]) <= 1e-7
    # the last step: rank r renders samples [(1 * 2 + r) * spp, + spp); together [2 spp, 4 spp) = one direct render
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ["--width", str(W), "--height", str(H), "--spp", str(2 * spp), "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stderr[-2000:]
    j1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][0])
    assert abs(j1["image_mean"] - j["image_mean"]) <= 1e-5       # same samples, summed in another order
    assert j["samples_per_s"] > 0 and abs(j["rays_per_sample"] - j1["rays_per_sample"]) < 1e-9
    # strong scaling (BASELINE configs[3]: "spp-sharded"): the total stays 2 spp... 16 samples, rank r renders
    # sample_range(r, 2, 16) of them; the image is the one GPU's image of the same 16 samples
    rs = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--workload", "tri_mesh", "--param", "20000",
                         "--width", str(W), "--height", str(H), "--spp", "16", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras"],
                        capture_output=True, text=True, timeout=300, env=env)
    assert rs.returncode == 0, rs.stderr[-2000:]
    js = json.loads(rs.stdout.splitlines()[-1])
    r1s = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tri_mesh", "--param", "20000", "--width", str(W), "--height", str(H),
                          "--spp", "16", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=300)
    assert r1s.returncode == 0, r1s.stderr[-2000:]
    j1s = json.loads(r1s.stdout.splitlines()[-1])
    assert js["scaling"] == "strong" and j1s["scaling"] == "strong" and js["config"]["spp_total"] == 16 and js["config"]["spp_per_gpu"] == 8
    assert abs(js["image_mean"] - j1s["image_mean"]) <= 1e-5 and abs(js["rays_per_sample"] - j1s["rays_per_sample"]) < 1e-9
    assert js["config"]["integrator"] == "wavefront"


def test_collectives_execute_through_rccl_with_one_rank():
    """No machine available to the build has two GPUs, so N > 1 over RCCL cannot run here.  What can: the same calls with
    ONE rank.  RT_BENCH_FORCE_DIST=1 makes bench.py create the nccl (= RCCL) process group at N = 1 and run every collective
    of the N > 1 flow -- reduce of the accumulation buffer onto rank 0 on the render stream, the two all-reduces of the
    counters, the barriers, the teardown -- under the launcher the contract names.  The image must be the plain run's."""
    W, H, spp = 64, 48, 8
    common = ["--width", str(W), "--height", str(H), "--spp", str(spp), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, RT_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RT_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common,
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 1 and "nccl" in j["config"]["parallelism"] and j["image_finite"]
    # ... and without a launcher (bench.py supplies the rendezvous itself)
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=300, env=env)
    assert r2.returncode == 0, r2.stderr[-2000:]
    j2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][0])
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0, plain.stderr[-2000:]
    jp = json.loads([ln for ln in plain.stdout.splitlines() if ln.startswith("{")][0])
    assert jp["config"]["parallelism"] == "single GPU"
    assert j["image_mean"] == jp["image_mean"] == j2["image_mean"]
    assert j["rays_per_sample"] == jp["rays_per_sample"]
