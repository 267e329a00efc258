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
