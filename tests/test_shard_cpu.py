"""spp sharding + reduce with world_size 2 on CPU (gloo).  The renderer on each rank is the CPU oracle;
what is under test is the product's shard plan and reduce (cuda-raytracer_amd/shard.py)."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as entry
import importlib.util
spec = importlib.util.spec_from_file_location("rt_shard", os.path.join(%r, "cuda-raytracer_amd", "shard.py"))
shard = importlib.util.module_from_spec(spec); spec.loader.exec_module(shard)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
b, o = entry.load_binding(), entry.load_oracle()
W, H, SPP = 24, 16, 7
sc = b.Scene("cornell", W, H)
s0, s1 = shard.sample_range(rank, world, SPP)
acc, st = o.Oracle().render(sc.flat, W, H, s0, s1, 419, 100, 0)
t = torch.from_numpy(acc.reshape(-1).copy())
shard.reduce_accum(t, dst=0)
if rank == 0:
    np.save(sys.argv[1], t.numpy().reshape(H, W, 3))
dist.barrier()
dist.destroy_process_group()
'''


def test_sample_range_partitions():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rt_shard", os.path.join(ROOT, "cuda-raytracer_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    for spp in (0, 1, 7, 256, 1000):
        for world in (1, 2, 3, 4, 8):
            r = [shard.sample_range(k, world, spp, first=5) for k in range(world)]
            assert r[0][0] == 5 and r[-1][1] == 5 + spp
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [e - s for s, e in r]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_reduce_equals_single_render(tmp_path, binding, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    out = tmp_path / "img.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29517", str(script), str(out)],
                          env=env, timeout=300)
    got = np.load(out)
    W, H, SPP = 24, 16, 7
    sc = binding.Scene("cornell", W, H)
    full, _ = oracle.render(sc.flat, W, H, 0, SPP, 419, 100, 0)
    # same samples, summed as (0..3)+(4..6) instead of 0..6: equal up to fp32 summation order
    assert np.allclose(got, full, rtol=1e-5, atol=1e-6)
    assert got.sum() > 0


def test_bench_step_ranges_weak_and_strong():
    """bench.py's sample ranges per (step, rank): weak = every rank its own `spp` fresh samples, strong = the step's `spp`
    samples cut over the ranks (BASELINE configs[3], "spp-sharded"); in both, no sample index is rendered twice"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rt_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for world in (1, 2, 4, 8):
        for scaling, per_step in (("weak", 32 * world), ("strong", 32)):
            seen = []
            for k in range(3):
                r = [bench.step_sample_range(scaling, k, rank, world, 32) for rank in range(world)]
                assert sum(e - b for b, e in r) == per_step
                seen += [i for b, e in r for i in range(b, e)]
            assert len(seen) == len(set(seen)) == 3 * per_step and min(seen) == 0 and max(seen) == 3 * per_step - 1
    assert bench.DEFAULT_SCALING["tri_mesh"] == "strong" and bench.DEFAULT_SCALING["cornell"] == "weak"


def test_bench_starts_its_own_ranks_and_passes_their_exit_code():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts torch.distributed.run as a child and
    hands on its exit code; in this container there is no GPU, so the ranks refuse to run and the parent fails with them,
    printing no JSON line"""
    env = dict(os.environ, RT_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras",
                        "--no-cpu-baseline", "--width", "16", "--height", "16", "--spp", "2"], capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0 and r.stdout.count("{\"metric\"") == 1
    else:
        assert r.returncode != 0 and "no GPU visible" in r.stderr and not r.stdout.strip()
