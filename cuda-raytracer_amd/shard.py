"""Sample-per-pixel sharding across the GPUs of a node (one process per GPU).

The reference's "multi-GPU" mode is two host threads that each rebuild the world and render one
image half (reference main.cu:128-174, 652-699).  Here every rank renders every pixel for its own
contiguous range of sample indices into an fp32 sum buffer, and the buffers are added onto the
root with one RCCL reduce over xGMI (torch.distributed backend "nccl"; "gloo" on CPU in tests).
Because the RNG is addressed by (pixel, global sample index), the image does not depend on the
number of ranks except for fp32 summation order.
"""


def sample_range(rank, world, spp_total, first=0):
    """contiguous [begin, end) of global sample indices for `rank`; the first spp_total % world
    ranks get one extra sample"""
    if world < 1 or not (0 <= rank < world) or spp_total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(spp_total, world)
    begin = first + rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def reduce_accum(accum, dst=0):
    """sum the per-rank accumulation buffers onto `dst` (no-op for a single process)"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
