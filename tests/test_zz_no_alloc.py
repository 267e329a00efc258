"""rt_render_accumulate_noalloc (include/rt_api.h): the entry point that allocates nothing must give the bits of
rt_render_accumulate in megakernel form, call after call, with the built-in scratch shared between calls."""
import pytest

pytestmark = pytest.mark.gpu


def _pair(binding, ctx, sc, W, H, s0, s1, seed):
    import torch
    a = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    b = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.set_integrator(binding.RT_INTEGRATOR_MEGAKERNEL)
    try:
        ctx.render_accumulate(a, W, H, s0, s1, seed, 100, sc.background)
    finally:
        ctx.set_integrator(binding.RT_INTEGRATOR_AUTO)
    ctx.synchronize()
    ctx.render_accumulate_noalloc(b, W, H, s0, s1, seed, 100, sc.background)
    ctx.synchronize()
    return a, b


def test_noalloc_matches_render_accumulate_twice(binding, gpu_ctx):
    import torch
    # first call: 16 runs of a 96x80 frame; second, right after it, a smaller frame with fewer runs and another
    # scene kind (BVH): whatever the first call left in the scratch must not show
    for name, param, W, H, s0, s1 in (("cornell_smoke", 0, 96, 80, 0, 1100), ("final", 8, 40, 24, 5, 5 + 200),
                                      ("cornell", 0, 33, 17, 0, 60)):
        sc = binding.Scene(name, W, H, param)
        gpu_ctx.upload(sc)
        gpu_ctx.reset_counters()
        a, b = _pair(binding, gpu_ctx, sc, W, H, s0, s1, 419)
        assert float(a.sum()) > 0
        assert torch.equal(a, b), name
        c = gpu_ctx.counters()
        assert c.samples == 2 * W * H * (s1 - s0) and c.render_launches == 1      # the no-alloc call is not timed
    # accumulation into a buffer that already holds sums, as the allocating call does
    W, H = 64, 64
    sc = binding.Scene("cornell", W, H)
    gpu_ctx.upload(sc)
    a, b = _pair(binding, gpu_ctx, sc, W, H, 0, 256, 7)
    gpu_ctx.set_integrator(binding.RT_INTEGRATOR_MEGAKERNEL)
    gpu_ctx.render_accumulate(a, W, H, 256, 512, 7, 100, sc.background)
    gpu_ctx.set_integrator(binding.RT_INTEGRATOR_AUTO)
    gpu_ctx.render_accumulate_noalloc(b, W, H, 256, 512, 7, 100, sc.background)
    gpu_ctx.synchronize()
    assert torch.equal(a, b)


def test_noalloc_limits(binding, gpu_ctx):
    import torch
    W, H = 1024, 1024
    sc = binding.Scene("cornell", W, H)
    gpu_ctx.upload(sc)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    with pytest.raises(binding.RtError) as e:
        gpu_ctx.render_accumulate_noalloc(acc, W, H, 0, 1024, 1, 100, 0)        # 16 runs x 1024^2: beyond the scratch
    assert "scratch" in str(e.value)
    gpu_ctx.render_accumulate_noalloc(acc, W, H, 0, 4, 1, 100, 0)               # one run: no scratch, any frame size
    gpu_ctx.synchronize()
    ref = torch.zeros_like(acc)
    gpu_ctx.set_integrator(binding.RT_INTEGRATOR_MEGAKERNEL)
    gpu_ctx.render_accumulate(ref, W, H, 0, 4, 1, 100, 0)
    gpu_ctx.set_integrator(binding.RT_INTEGRATOR_AUTO)
    gpu_ctx.synchronize()
    assert torch.equal(acc, ref)
    gpu_ctx.render_accumulate_noalloc(acc, W, H, 3, 3, 1, 100, 0)               # empty range
    with pytest.raises(binding.RtError):
        gpu_ctx.render_accumulate_noalloc(acc, W, H, 4, 3, 1, 100, 0)
    fresh = binding.Context(0)
    with pytest.raises(binding.RtError):
        fresh.render_accumulate_noalloc(acc, W, H, 0, 1, 1, 100, 0)              # no scene
    fresh.close()


def test_noalloc_under_graph_capture(binding):
    """the callers the entry point is for include code under HIP graph capture: the call is captured on the context's
    stream (nothing runs), the graph is instantiated and replayed twice, and the buffer then holds what two allocating
    calls onto one buffer leave, bit for bit"""
    import torch
    W, H, s0, s1 = 96, 80, 0, 300                      # three runs: the scratch is in use
    stream = torch.cuda.Stream()
    ctx = binding.Context(0, stream.cuda_stream)
    try:
        sc = binding.Scene("cornell", W, H)
        ctx.upload(sc)
        ref = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
        got = torch.zeros_like(ref)
        torch.cuda.synchronize()
        ctx.set_integrator(binding.RT_INTEGRATOR_MEGAKERNEL)
        ctx.render_accumulate(ref, W, H, s0, s1, 11, 100, sc.background)
        ctx.set_integrator(binding.RT_INTEGRATOR_AUTO)
        ctx.render_accumulate_noalloc(got, W, H, s0, s1, 11, 100, sc.background)     # eager once: modules loaded before the capture
        ctx.synchronize()
        assert torch.equal(got, ref)
        got.zero_()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            ctx.render_accumulate_noalloc(got, W, H, s0, s1, 11, 100, sc.background)
        torch.cuda.synchronize()
        assert float(got.abs().sum()) == 0.0           # captured, not run
        graph.replay()
        graph.replay()
        torch.cuda.synchronize()
        ctx.set_integrator(binding.RT_INTEGRATOR_MEGAKERNEL)
        ctx.render_accumulate(ref, W, H, s0, s1, 11, 100, sc.background)             # the allocating call, a second time onto the same sums
        ctx.set_integrator(binding.RT_INTEGRATOR_AUTO)
        ctx.synchronize()
        assert float(ref.sum()) > 0 and torch.equal(got, ref)
    finally:
        ctx.close()
