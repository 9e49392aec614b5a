"""Stream sharding across ranks and the terminal exchange: no data-path collective — streams are
independent — only one all-reduce of counters / max elapsed and one all-gather of per-stream digests at the end.
Backend-agnostic: RCCL ("nccl") with CUDA tensors on the GPU box, "gloo" with CPU tensors in the CPU tests."""
import numpy as np


def shard_stream_ids(rank, world, streams_per_rank):
    """Global stream ids owned by `rank` (static placement: shard s -> rank s // streams_per_rank)."""
    return list(range(rank * streams_per_rank, (rank + 1) * streams_per_rank))


def terminal_exchange(dist, device, counters, elapsed, digests):
    """counters: list of floats to SUM; elapsed: float to MAX; digests: list of ints gathered in rank order.
    Returns (summed counters, max elapsed, digests of all ranks concatenated). dist=None -> single process."""
    import torch
    if dist is None:
        return [float(c) for c in counters], float(elapsed), [int(d) for d in digests]
    c = torch.tensor([float(x) for x in counters], dtype=torch.float64, device=device)
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    d = torch.tensor([int(x) & 0x7fffffffffffffff for x in digests], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [torch.zeros_like(d) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, d)
    return [float(x) for x in c.cpu()], float(t.cpu()[0]), [int(x) for g in gathered for x in g.cpu()]
