"""Parity witness at the measured configuration (the parity gates must hold for any reported number).

GPU: >= 64 streams in >= 16 free-running stream groups through icgh_batch_run, frames resident in HBM, ICG_WAIT_POLL,
GPU_MAX_HW_QUEUES=20 — the path that produces bench.py's `value` — 640x480/100 for all streams plus 1280x720/300 for 8 streams,
16 frames each: every stream's digest, state series, last-frame map-point ids / key-point bits and candidate lists must equal the
oracle-backed host layer's (reference per-frame algorithm: tracking/tracking.cc:144-245).  The tool runs in its own process so that
GPU_MAX_HW_QUEUES is in place before the HIP runtime starts.

CPU: the same tool with both sides on the oracle-backed host layer (grouped + threaded vs one batch): the comparison logic itself."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tests", "tools", "parity_at_scale.py")


def _run(args, timeout):
    env = dict(os.environ, GPU_MAX_HW_QUEUES="20")
    r = subprocess.run([sys.executable, TOOL] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return r.returncode, json.loads(lines[-1])


@pytest.mark.gpu
def test_measured_configuration_matches_oracle():
    rc, out = _run(["--streams", "64", "--groups", "16", "--frames", "16", "--c2-streams", "8", "--c2-groups", "4"], 900)
    for c in out["cases"]:
        assert not c["mismatches"], c
        assert c["tracking_states"] > c["streams"] * (c["frames"] - 8), c  # the streams really track (not all PASSED / LOST)
        assert c["mappoints"] > 40 * c["streams"], c
    assert out["cases"][0]["streams"] >= 64 and out["cases"][0]["groups"] >= 16
    assert out["cases"][1]["config"].startswith("1280x720") and out["cases"][1]["streams"] >= 8
    assert rc == 0 and out["ok"]


def test_parity_tool_selftest_cpu():
    rc, out = _run(["--cpu-selftest", "--streams", "6", "--groups", "3", "--frames", "8", "--c2-streams", "0"], 600)
    assert rc == 0 and out["ok"], out
    assert out["cases"][0]["groups"] == 3 and out["cases"][0]["tracking_states"] > 0
