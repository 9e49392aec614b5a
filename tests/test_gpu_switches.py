"""Every run-time switch the product libraries read from the environment (`grep getenv ic-gvins_amd/csrc ic-gvins_amd/host`) under test on
the MI355X: tests/tools/switch_probe.py — tracker, window refinement, batched solver, batched marginalization — runs in a child process
per environment, and every combination below must reproduce the digests of the clean environment bit for bit.  The switches that select
an ENGINE or a solver path have their own tests (ICG_TRACK_ENGINE: test_gpu_device_tracker.py and the probe's second leg here;
ICG_SOLVER_DEVICE_CHOLESKY: test_gpu_solver.py; ICG_LOCKSTEP_MARG_BATCH: test_gpu_zz_marg_batch.py; ICG_TRACKING_LOG_DIR:
ref_tracking_utils.py; ICG_HOST_CHECK: conftest.py sets it for every test).  Kernel variants that used to hide behind switches
(ICG_LK_PAIR, ICG_PYRAMID_TILES, ICG_CLAHE_LEGACY, ICG_LK_REUSE, ICG_RANSAC_DEVICE_LOOP, ICG_MARG_DENSE) were deleted in round 6."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "tools", "switch_probe.py")

# scheduling: where a wait happens, how many host threads, how launches are grouped — placement only
SCHEDULING = {"ICG_WAIT_MODE": "poll:20", "ICG_TRACKER_WAIT": "block", "ICG_SOLVER_THREADS": "3", "ICG_GROUP_STAGGER": "0",
              "ICG_HOST_MALLOC_POLICY": "raise", "ICG_TRACKER_LOG_DRAIN": "30", "ICG_REFINE_PER_STREAM": "1"}
# diagnostics: text on stderr, nothing else
DIAGNOSTICS = {"ICG_ABI_DEBUG": "1", "ICG_SOLVER_DEBUG": "1", "ICG_MARG_DEBUG": "1", "ICG_DEBUG_TIMING": "1", "ICG_HOST_PROF": "1", "ICG_DEBUG_TRI": "1"}


def run_probe(extra):
    env = {k: v for k, v in os.environ.items() if not k.startswith("ICG_") or k in ("ICG_HOST_CHECK",)}
    env.update(extra)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests"), os.path.join(ROOT, "ic-gvins_amd"), env.get("PYTHONPATH", "")])
    r = subprocess.run([sys.executable, PROBE], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (extra, r.stderr[-2000:])
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


def switches_in_the_sources():
    found = set()
    for sub in ("csrc", "host"):
        d = os.path.join(ROOT, "ic-gvins_amd", sub)
        for name in os.listdir(d):
            if name.endswith((".hip", ".cc", ".h")):
                found |= set(re.findall(r'getenv\("(ICG_[A-Z_]+)"\)', open(os.path.join(d, name)).read()))
    return found


def test_every_switch_of_the_product_is_covered():
    covered = set(SCHEDULING) | set(DIAGNOSTICS) | {"ICG_TRACK_ENGINE", "ICG_SOLVER_DEVICE_CHOLESKY", "ICG_TRACKING_LOG_DIR", "ICG_HOST_CHECK"}
    assert switches_in_the_sources() <= covered, sorted(switches_in_the_sources() - covered)


def test_switches_leave_every_result_as_it_is():
    for engine in ("device", "table"):
        clean, _ = run_probe({"ICG_TRACK_ENGINE": engine})
        assert clean["engine"] == engine
        sched, _ = run_probe(dict(SCHEDULING, ICG_TRACK_ENGINE=engine))
        assert sched == clean, (engine, "scheduling", sched, clean)
        diag, err = run_probe(dict(DIAGNOSTICS, ICG_TRACK_ENGINE=engine))
        assert diag == clean, (engine, "diagnostics", diag, clean)
        for needle in ("[icg_reproj_schur]", "[WindowSolverBatch]", "[marginalization"):
            assert needle in err, (engine, needle)
    # the engines agree with each other on everything but the engine's own state text
    a, _ = run_probe({"ICG_TRACK_ENGINE": "device"})
    b, _ = run_probe({"ICG_TRACK_ENGINE": "table"})
    for key in ("track_states", "track_digests", "refine", "solve_batch", "marg_batch"):
        assert a[key] == b[key], key
