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


def host_plan(usable_cores, world, local_rank, cpu_ids=None, groups_override=0, streams_override=0):
    """Host resources of one rank (one rank's polling threads must not assume the whole box):
      cores_rank   the rank's share of the usable host cores
      groups       stream groups (one host thread + HIP stream each): 3 per core of the share — a group sleeps while its kernels run, and the
                   GPU only fills up at ~48 groups in flight (profiles/README.md, round-2 sweep: 32 -> 66 k, 48 -> 72 k, 64 -> 73 k frames/s) —
                   between 2 and 48; never 8 pollers on 2 cores
      streams      8 camera streams per group
      cpu_slice    the contiguous slice of the allowed CPU ids this rank pins itself to (None for a single rank: nothing to separate)
    """
    cores_rank = float(usable_cores) / float(max(1, world))
    groups = int(groups_override) if groups_override > 0 else int(max(2, min(48, 3 * round(cores_rank))))
    streams = int(streams_override) if streams_override > 0 else 8 * groups
    groups = max(1, min(groups, streams))
    cpu_slice = None
    if world > 1 and cpu_ids:
        ids = sorted(cpu_ids)
        per = max(1, len(ids) // world)
        cpu_slice = ids[local_rank * per:(local_rank + 1) * per] or ids
    return {"cores_rank": cores_rank, "groups": groups, "streams": streams, "cpu_slice": cpu_slice}
