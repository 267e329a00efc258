"""rt_reduce_accum (include/rt_api.h): the RCCL exchange step of the multi-GPU path, as far as one GPU can show
it: librccl loads on first use, a communicator over this device is made and reused, the reduce of a single
shard leaves its buffer as it was, and argument errors never reach RCCL.  (N > 1: tests/test_shard_cpu.py runs
the same sharding with gloo at world size 2; the round-end 8-GPU bench runs it on RCCL.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def reduce(lib, ctxs, bufs, count, root):
    n = len(ctxs)
    a = (C.c_void_p * max(n, 1))(*[c.handle for c in ctxs])
    b = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in bufs])
    return lib.rt_reduce_accum(a, b, n, count, root)


def test_single_shard_reduce_and_argument_errors(binding, gpu_ctx):
    import torch
    lib = binding.lib()
    W, H = 64, 48
    sc = binding.Scene("cornell", W, H)
    gpu_ctx.upload(sc)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.render_accumulate(acc, W, H, 0, 4, 419, 100, sc.background)
    gpu_ctx.synchronize()
    before = acc.clone()
    for _ in range(2):                                   # second call reuses the communicator
        assert reduce(lib, [gpu_ctx], [acc], acc.numel(), 0) == 0, lib.rt_last_error()
        gpu_ctx.synchronize()
        assert torch.equal(acc, before) and float(acc.sum()) > 0
    # errors that must be caught before any RCCL call
    other = binding.Context(0)
    try:
        assert reduce(lib, [gpu_ctx, other], [acc, acc], acc.numel(), 0) == -1       # two contexts on one device
        assert b"one device" in lib.rt_last_error()
    finally:
        other.close()
    assert reduce(lib, [gpu_ctx], [acc], acc.numel(), 1) == -1                       # root out of range
    assert reduce(lib, [gpu_ctx], [acc], 0, 0) == -1
    assert reduce(lib, [], [], 16, 0) == -1
    assert torch.equal(acc, before)


def test_single_shard_reduce_scatter(binding, gpu_ctx):
    """rt_reduce_scatter_rows with one shard: its band is the whole frame and the buffer is left as it was"""
    import torch
    lib = binding.lib()
    W, H = 40, 30
    sc = binding.Scene("cornell", W, H)
    gpu_ctx.upload(sc)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.render_accumulate(acc, W, H, 0, 2, 419, 100, sc.background)
    gpu_ctx.synchronize()
    before = acc.clone()
    a = (C.c_void_p * 1)(gpu_ctx.handle)
    b = (C.c_void_p * 1)(acc.data_ptr())
    rows = (C.c_int32 * 2)(-1, -1)
    assert lib.rt_reduce_scatter_rows(a, b, 1, W, H, rows) == 0, lib.rt_last_error()
    gpu_ctx.synchronize()
    assert list(rows) == [0, H] and torch.equal(acc, before)
    assert lib.rt_reduce_scatter_rows(a, b, 1, W, 0, rows) == -1
    assert lib.rt_reduce_scatter_rows(a, b, 1, W, H, None) == -1
    assert lib.rt_reduce_scatter_rows(a, b, 0, W, H, rows) == -1
