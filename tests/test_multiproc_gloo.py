"""N>1 path on CPU: two gloo ranks each track their shard of the streams (oracle-backed host layer) and run the same
terminal exchange bench.py uses over RCCL; the gathered per-stream digests must equal a single-process run of all streams
(placement invariance) and the reduced counters must add up."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, nframes, per_rank, out_q):
    sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import sharding
    from stream_utils import ensure_oracle_host, run_streams
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = sharding.shard_stream_ids(rank, world, per_rank)
    rec, stats, _ = run_streams(ensure_oracle_host(), per_rank, 320, 240, nframes, 60, stream_ids=ids)
    tracked = sum(s["tracked_sum"] for s in stats)
    counters, tmax, digests = sharding.terminal_exchange(dist, "cpu", [per_rank * nframes, tracked], 0.5 + rank, [s["digest"] for s in stats])
    if rank == 0:
        out_q.put((counters, tmax, digests, tracked))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
    from stream_utils import ensure_oracle_host, run_streams
    ensure_oracle_host()
    world, per_rank, nframes = 2, 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, world, port, nframes, per_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    counters, tmax, digests, tracked0 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rec, stats, _ = run_streams(ensure_oracle_host(), world * per_rank, 320, 240, nframes, 60)
    assert digests == [s["digest"] & 0x7fffffffffffffff for s in stats]
    assert counters[0] == world * per_rank * nframes
    assert counters[1] == sum(s["tracked_sum"] for s in stats)
    assert tmax == 1.5  # MAX over ranks


def test_eight_rank_sharding_matches_single_process():
    """the 8-GPU placement (one shard per rank, BASELINE configs[4]) on gloo: eight ranks, one stream each, the terminal exchange of
    bench.py — the gathered digests equal ONE process owning all eight streams (placement invariance of 8 shards)"""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
    from stream_utils import ensure_oracle_host, run_streams
    ensure_oracle_host()
    world, per_rank, nframes = 8, 1, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, world, port, nframes, per_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    counters, tmax, digests, _ = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rec, stats, _ = run_streams(ensure_oracle_host(), world * per_rank, 320, 240, nframes, 60)
    assert digests == [s["digest"] & 0x7fffffffffffffff for s in stats]
    assert counters[0] == world * per_rank * nframes and counters[1] == sum(s["tracked_sum"] for s in stats)
    assert tmax == 7.5  # MAX over ranks (0.5 + rank)


def test_host_plan_scales_the_polling_threads_with_the_rank_share():
    """bench.py's per-rank host plan (sharding.host_plan): group threads follow the rank's share of the usable cores (never 8 pollers on
    2 cores), and the ranks of a node pin themselves to disjoint CPU slices"""
    sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
    import sharding
    one = sharding.host_plan(16, 1, 0, cpu_ids=range(256))
    # round 5: the device-resident tracker is the engine of every rank (it needs ~0.5 host cores per GPU); the host engines are overrides
    assert one["groups"] == 4 and one["engine"] == "device" and one["streams"] == 768 and one["cpu_slice"] is None
    tab = sharding.host_plan(16, 1, 0, cpu_ids=range(256), engine_override="table")
    assert tab["groups"] == 16 and tab["engine"] == "table"
    eight = [sharding.host_plan(16, 8, r, cpu_ids=range(256)) for r in range(8)]
    # weak scaling: the streams of a GPU do not shrink with the world size
    assert all(p["groups"] == 4 and p["engine"] == "device" and p["streams"] == 768 and abs(p["cores_rank"] - 2.0) < 1e-12 for p in eight)
    slices = [set(p["cpu_slice"]) for p in eight]
    assert all(len(s_) == 32 for s_ in slices) and len(set().union(*slices)) == 256  # disjoint, covering
    big = sharding.host_plan(128, 8, 3, cpu_ids=range(128))
    assert big["groups"] == 4 and big["engine"] == "device" and big["streams"] == 768 and big["cpu_slice"] == list(range(48, 64))
    assert sharding.host_plan(128, 8, 3, cpu_ids=range(128), engine_override="table")["groups"] == 16
    four = sharding.host_plan(16, 4, 1, cpu_ids=range(256))
    assert four["engine"] == "device" and four["groups"] == 4 and abs(four["cores_rank"] - 4.0) < 1e-12
    tiny = sharding.host_plan(4, 8, 0, cpu_ids=range(4))
    assert tiny["groups"] == 4 and tiny["engine"] == "device" and tiny["streams"] == 768 and tiny["cpu_slice"]
    assert sharding.host_plan(16, 8, 0, engine_override="table")["groups"] == 8
    assert sharding.host_plan(16, 1, 0, groups_override=8, streams_override=64)["groups"] == 8


def _run_bench_selftest(world, streams, port, details, engine="table"):
    """bench.py's own main() as the driver launches it (python -m torch.distributed.run ... bench.py --gpus N ...), in its CPU plumbing
    self-test mode (ICG_BENCH_SELFTEST_ORACLE=1: gloo + the oracle-backed checker build of the host layer; value is null by construction)"""
    import json
    import subprocess
    env = dict(os.environ, ICG_BENCH_SELFTEST_ORACLE="1", MASTER_ADDR="127.0.0.1")
    tiny = ["--steps", "3", "--warmup", "1", "--width", "320", "--height", "240", "--features", "60", "--streams", str(streams),
            "--groups", "2", "--ring", "6", "--prime", "4", "--details", details, "--engine", engine]
    bench = os.path.join(ROOT, "bench.py")
    if world == 1:
        cmd = [sys.executable, bench, "--gpus", "1"] + tiny
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), bench, "--gpus", str(world)] + tiny
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE json line
    return json.loads(lines[0])


def test_bench_main_runs_two_ranks_as_the_driver_launches_it(tmp_path):
    """the N>1 path of bench.py ITSELF — env ranks, per-rank CPU slice, ring sizing, priming, warm-up, the barrier-bracketed
    timed region, MAX over ranks, the terminal exchange (all-reduce of counters, all-gather of digests) and the contract line — executed as
    written with two gloo ranks, and checked for placement invariance against the same main() as ONE rank owning all four streams."""
    from stream_utils import ensure_oracle_host
    ensure_oracle_host()
    port = 29900 + (os.getpid() % 90)
    # the two ranks on the device engine (the tracker ABI, here on the shim's CPU backend: what an 8-rank run on a 16-CPU box selects), the
    # single rank on the track table: same four streams, same digests — placement AND engine invariance through bench.main itself
    two = _run_bench_selftest(2, 2, port, str(tmp_path / "two.json"), engine="device")
    one = _run_bench_selftest(1, 4, port + 1, str(tmp_path / "one.json"), engine="table")
    assert "device-resident" in two["config"]["engine"] and "track table" in one["config"]["engine"]
    for line, n in ((two, 2), (one, 1)):
        assert line["value"] is None and "selftest" in line  # never a measurement
        assert line["n_gpus"] == n and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
        assert line["selftest"]["frames"] == 4 * 3                       # all ranks' frames of the timed region, summed
        assert line["quality"]["tracking_state_fraction"] == 1.0
        assert line["ms_per_step"] > 0 and line["selftest"]["elapsed_max_s"] > 0
    # the single-rank run carries the engine twin: the same four streams on the other engine (device tracker on the shim's CPU backend), digest by digest
    assert one["engine_twin"]["engine"] == "device" and one["engine_twin"]["ok"] is True and one["engine_twin"]["digests_compared"] == 4
    assert two.get("engine_twin") is None  # (N > 1 measures the sharded front-end only)
    assert two["config"]["streams_per_gpu"] == 2 and two["config"]["frames_per_step"] == 4 and "pinned" in two["config"]["cpu_slice_per_rank"]
    # every rank reports itself in the gathered line: its own rate, busy host cores, engine
    assert [r["rank"] for r in two["ranks"]] == [0, 1] and all(r["engine"] == "device" and r["groups"] == 2 and r["frames_per_s"] > 0 for r in two["ranks"])
    assert all(r["cpu_cores_busy"] >= 0 for r in two["ranks"]) and [r["engine"] for r in one["ranks"]] == ["table"]
    # rank order of the gathered digests = global stream order: the same four streams, wherever they ran
    # (the gathered digests travel as int64: 63 bits)
    assert two["selftest"]["digests"] == [d & 0x7fffffffffffffff for d in one["selftest"]["digests"]]
    assert two["selftest"]["tracked_mappoints"] == one["selftest"]["tracked_mappoints"]
