"""Final scene at 3840x2160: one 512-spp render as ONE call against the same samples in chunks of 256 / 128 / 64.
A measurement script for the GPU box (python3 tools/chunk_sweep.py); nothing imports it."""
import json
import os
import sys


def main():
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, root)
    import torch
    import __graft_entry__ as e
    b = e.load_binding()
    ctx = b.Context(0)
    W, H = 3840, 2160
    sc = b.Scene("final", W, H, 100)
    ctx.upload(sc)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.render_accumulate(acc, W, H, 0, 16, 419, 100, sc.background)
    ctx.synchronize()
    for total, chunk in ((512, 512), (512, 128), (512, 64), (512, 256)):
        ctx.reset_counters()
        for s0 in range(0, total, chunk):
            ctx.render_accumulate(acc, W, H, 1000 + s0, 1000 + s0 + chunk, 419, 100, sc.background)
        c = ctx.counters()
        print(json.dumps({"total": total, "chunk": chunk, "Grays_per_s": c.rays / (c.render_ms * 1e-3) / 1e9, "ms": c.render_ms}), flush=True)


if __name__ == "__main__":
    main()
