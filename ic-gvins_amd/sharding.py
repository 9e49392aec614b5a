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


def gather_rank_rows(dist, device, row):
    """row: list of floats describing THIS rank (frames/s, busy host cores, engine code ...); -> the rows of all ranks in rank order.
    Part of the terminal exchange (one more all-gather of a few doubles): the gathered bench line shows what every rank did, not only
    rank 0 — VERDICT r4 item 6."""
    import torch
    if dist is None:
        return [[float(x) for x in row]]
    t = torch.tensor([float(x) for x in row], dtype=torch.float64, device=device)
    gathered = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, t)
    return [[float(x) for x in g.cpu()] for g in gathered]


ENGINE_CODES = {"table": 0, "object": 1, "core": 2, "device": 3}
STREAMS_PER_GPU = 768  # fixed work per GPU whatever the world size: "scaling": "weak" means exactly this


def host_plan(usable_cores, world, local_rank, cpu_ids=None, groups_override=0, streams_override=0, engine_override=None):
    """Host resources of one rank (one rank's polling threads must not assume the whole box):
      cores_rank   the rank's share of the usable host cores
      streams      768 camera streams per GPU, the same for every world size (weak scaling: per-GPU work is fixed)
      groups       stream groups (one host thread + HIP stream each).  With the track-table engine a frame costs the host ~20 us of logic, so
                   a group can carry 32-64 streams; round-3 sweep on one MI355X (profiles/archive/r03_group_sweep.txt): 48 x 8 -> 78 k, 24 x 16 -> 92 k,
                   16 x 24 -> 98.6 k, 12 x 32 -> 99.9 k, 8 x 48 -> 100.7 k frames/s with 3.3-3.9 host cores busy; larger launches have shorter
                   tails: 12 x 64 -> 111.0 k, 16 x 64 -> 111.7 k, 8 x 96 -> 105.5 k (4.2 cores busy).  4 groups per core of the share, between
                   4 and 12 (every thread confined to 2 CPUs, profiles/archive/r03_cpu_quota.md: 8 x 96 -> 44.4 k, 6 x 128 -> 42.1 k frames/s).
      cpu_slice    the contiguous slice of the allowed CPU ids this rank pins itself to (None for a single rank: nothing to separate)
    """
    cores_rank = float(usable_cores) / float(max(1, world))
    # engine.  The device-resident tracker (csrc/tracker.hip: the streams' state in HBM, one launch chain + one wait per step) is THE engine
    # since round 5, for every rank whatever its share of the host: at the driver's command (20 timed steps) 133.5 k frames/s with 0.56 host
    # cores busy against 127.6 k with 5.4 cores busy for the track table (profiles/archive/r05_call9: same box, same minute; 137.5 k with every
    # thread confined to 2 CPUs, 119.0 k on ONE, where the table collapses to 47.5 k); on 200-step runs the two are within +-3 %
    # (130.9 / 135.2 k).  Rounds 1-3's host engines (track table, object graph, tracker core on the host) stay selectable
    # (--engine / ICG_TRACK_ENGINE) and run as the bench's engine twin and in the tests: same results, state for state
    # (tests/test_gpu_device_tracker.py, tests/test_host_engines_cpu.py; the 2-rank bench self-test runs one rank count on each engine).
    engine = engine_override or "device"
    if engine == "device":
        # wide launches: the stage kernels cost the same whatever the number of streams (a wave per stream).  4 x 192 -> 106.9 k, 12 x 64 ->
        # 108.8 k, 8 x 192 (1536 streams) -> 111.8 k; 2 confined CPUs: 4 x 192 -> 106.7 k, 8 x 96 -> 106.1 k
        groups = int(groups_override) if groups_override > 0 else 4
    else:
        # round 5 (profiles/archive/r05_call7: with 12 groups no LK launch is in flight 24 % of the time; r05_call5 / r05_call8 sweeps on three boxes:
        # 12 x 64 -> 131.6-134.7 k, 16 x 48 -> 134.7 k, 16 x 64 -> 135.6-138.2 k, 24 x 32 -> 125.5 k, 24 x 48 -> 134.2 k, 32 x 32 -> 130.8 k)
        groups = int(groups_override) if groups_override > 0 else int(max(4, min(16, 4 * round(cores_rank))))
    streams = int(streams_override) if streams_override > 0 else STREAMS_PER_GPU
    groups = max(1, min(groups, streams))
    cpu_slice = None
    if world > 1 and cpu_ids:
        ids = sorted(cpu_ids)
        per = max(1, len(ids) // world)
        cpu_slice = ids[local_rank * per:(local_rank + 1) * per] or ids
    return {"cores_rank": cores_rank, "groups": groups, "streams": streams, "cpu_slice": cpu_slice, "engine": engine}
