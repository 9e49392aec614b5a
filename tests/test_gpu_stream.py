"""GPU end-to-end parity: the shipped host layer on libicgvins_hip.so vs the same host layer on the CPU oracle, frame by
frame — track states, map-point (track) ids and feature pixel bits must be identical (bit-exact track IDs / indices,
north_star)."""
import os

import numpy as np
import pytest

import harness as H
from stream_utils import ensure_oracle_host, run_streams

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [(640, 480, 100, 20), (1280, 720, 300, 14)])
def test_stream_parity_hip_vs_oracle(cfg):
    w, h, nfeat, nframes = cfg
    rec_o, stats_o, frames = run_streams(ensure_oracle_host(), 1, w, h, nframes, nfeat)
    rec_g, stats_g, _ = run_streams(H.HOST_LIB, 1, w, h, nframes, nfeat, scene_frames=frames)
    for k in range(nframes):
        so, ido, pxo = rec_o[k][0]
        sg, idg, pxg = rec_g[k][0]
        assert so == sg, (k, so, sg)
        assert np.array_equal(ido, idg), (k, len(ido), len(idg))
        assert np.array_equal(pxo.view(np.uint32), pxg.view(np.uint32)), k
    assert stats_o[0]["digest"] == stats_g[0]["digest"]
    assert stats_g[0]["mappoints_created"] > 40


def test_multi_stream_batch_on_gpu_equals_oracle():
    w, h, nfeat, nframes, ns = 640, 480, 100, 12, 4
    rec_o, stats_o, frames = run_streams(ensure_oracle_host(), ns, w, h, nframes, nfeat)
    rec_g, stats_g, _ = run_streams(H.HOST_LIB, ns, w, h, nframes, nfeat, scene_frames=frames, host_threads=2, groups=2)
    for s in range(ns):
        assert stats_o[s]["digest"] == stats_g[s]["digest"], s


@pytest.mark.parametrize("scenario", ["c1_640x480_100", "c2_1280x720_300", "c1_histgate", "c1_lost_and_reinit", "c1_lost_histgate", "c1_slow_second_new", "c4_1920x1080_500", "c1_bgr", "c1_long_160", "c2_long_60"])
def test_gpu_host_layer_matches_reference_tracker_golden(scenario):
    """icg::Tracking on the HIP kernels vs what the REFERENCE's own tracking.cc produced on the oracle primitives
    (tests/golden/tracking_ref_*.npz): track states, map-point ids, key-point float bits, window bookkeeping per frame."""
    import ref_tracking_utils as rt
    rt.compare_scenario(H.HOST_LIB, scenario)


def test_bench_terminal_exchange_runs_through_rccl_on_one_gpu(tmp_path):
    """SURVEY.md section 8(e) / VERDICT r5 item 6: the collective leg of the multi-GPU path — sharding.terminal_exchange (all-reduce of the
    counters, all-gather of the digests) and sharding.gather_rank_rows — executed by bench.main through RCCL (backend "nccl") on cuda:0,
    launched the way the driver launches N > 1 (`python -m torch.distributed.run --nproc-per-node 1`), and again as the plain N = 1
    command, whose exchange goes through a one-rank RCCL group by default.  Small shapes: this is about the exchange, not the rate."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    tiny = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--width", "640", "--height", "480", "--features", "100", "--streams", "8", "--groups", "2",
            "--ring", "6", "--prime", "4", "--no-reproj", "--no-c4", "--no-replay", "--no-cpu-baseline", "--no-engine-twin", "--no-profile-pass",
            "--force-dist"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    lines = []
    for k, launcher in enumerate(([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                                   "--master-port", str(29700 + os.getpid() % 200)], [sys.executable])):
        r = subprocess.run(launcher + [bench] + tiny + ["--details", str(tmp_path / f"d{k}.json")], env=env, cwd=root, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(out) == 1, r.stdout[-2000:]
        lines.append(json.loads(out[0]))
    for line in lines:
        assert line["exchange"].startswith("rccl"), line["exchange"]
        assert line["n_gpus"] == 1 and line["value"] > 0
        assert [r_["rank"] for r_ in line["ranks"]] == [0] and line["ranks"][0]["frames_per_s"] > 0 and line["ranks"][0]["engine"] == "device"
        assert line["parity"]["ok"] is True  # (the witness streams' digests equal the oracle-backed checker's)
    # the same streams either way: the quality block is computed from the all-reduced counters
    assert lines[0]["quality"] == lines[1]["quality"]
