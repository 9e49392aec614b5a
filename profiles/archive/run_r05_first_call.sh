#!/bin/bash
# First GPU call of round 5: everything round 4 left verified on the CPU backend only, then the numbers that go with it.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash profiles/run_r05_first_call.sh'
# 1. the whole GPU suite (tests/test_gpu_zz_marg_batch.py last: MarginalizationBatch in its final form, the lock-step replay with shared
#    marginalizations)
# 2. (now inside the suite: tests/test_gpu_device_tracker.py::test_device_tracker_writes_the_reference_tracking_txt, and the 15-keyframe
#    lock-step replay in tests/test_gpu_zz_marg_batch.py)
# 3. MarginalizationBatch throughput (steady state: one batch object, three passes) at 16 / 64 / 256 windows, with the phase split
# 4. lock-step replay of 8 estimators in one group with and without ICG_LOCKSTEP_MARG_BATCH=1
# 5. the driver's command (engine_twin and marg.batched in the line)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5first
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 420 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -24 > $O/gputests.txt; cat $O/gputests.txt
ICG_MARG_DEBUG=1 timeout 120 python profiles/marg_batch_probe.py $O/marg_batch_probe.json > $O/marg_batch_probe.out 2> $O/marg_batch_probe.err
cat $O/marg_batch_probe.json; grep "batch\]" $O/marg_batch_probe.err | tail -3
timeout 300 python - > $O/lockstep_marg.txt 2>&1 <<'PY'
import ctypes as C, os, tempfile, json
import gvins_checks as gc, gvins_data as gd, harness as H
lib = C.CDLL(H.TOOLS_LIB)
root = tempfile.mkdtemp(prefix="r5lock_")
files = gd.Sequence(lib).write(root)
gc.run_replay(lib, files)
out = {}
for flag in ("0", "1", "0", "1"):
    os.environ["ICG_LOCKSTEP_MARG_BATCH"] = flag
    outs = [os.path.join(root, "l%s_%d" % (flag, k)) for k in range(8)]
    SS, wall, shared = gc.run_replay_lockstep(lib, files, outs, groups=1)
    out.setdefault(flag, []).append({"wall_s": round(wall, 3), "x_real_time": round(sum(s["data_seconds"] for s in SS) / wall, 1),
                                     "marg_batches_windows": gc.lockstep_marg_counts(lib), "shared": shared})
print(json.dumps(out))
PY
tail -2 $O/lockstep_marg.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/driver_details.json > $O/driver_line.json 2> $O/driver.err
python - <<PY
import json
d = json.loads(open("$O/driver_line.json").read().strip().splitlines()[-1])
print(d["value"], d["config"]["engine"], d["host"], (d.get("parity") or {}).get("ok"), d.get("engine_twin"), d["marg"], (d.get("roofline") or {}).get("traffic"))
PY
