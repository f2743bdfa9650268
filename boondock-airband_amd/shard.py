"""Multi-GPU host logic: independent device streams shard across ranks (one process per GPU); there is no
exchange step inside the path (the reference runs one demod thread per device, rtl_airband.cpp:1044-1078).
The only collective is the gather of decimated audio to rank 0 for output (BASELINE.json north_star),
issued through torch.distributed -- backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests."""


def stream_range(rank, world, nstreams):
    """Stream-major contiguous partition: rank r owns [lo, hi).  Per-channel state never migrates."""
    if world < 1 or not (0 <= rank < world) or nstreams < 0:
        raise ValueError("bad rank/world/nstreams")
    base, extra = divmod(nstreams, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_audio(local, dst=0, group=None):
    """Gather each rank's [streams_local][nch][n] audio tensor to `dst`.  Ranks may own different numbers of
    streams (uneven partition): shapes are exchanged first.  Returns the list of per-rank tensors on dst,
    None elsewhere.  Point-to-point under the hood (RCCL gather = grouped send/recv), so rank-0 ingress over
    xGMI is the bound, not HBM."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    shape = torch.tensor(list(local.shape), dtype=torch.int64, device=local.device)
    shapes = [torch.empty_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)
    if rank == dst:
        outs = [torch.empty(tuple(int(v) for v in s.tolist()), dtype=local.dtype, device=local.device) for s in shapes]
        outs[dst].copy_(local)
        reqs = [dist.irecv(outs[r], src=r, group=group) for r in range(world) if r != dst]
        for q in reqs:
            q.wait()
        return outs
    dist.send(local.contiguous(), dst=dst, group=group)
    return None
