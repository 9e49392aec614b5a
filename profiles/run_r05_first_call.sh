#!/bin/bash
# First GPU call of round 5: everything round 4 left verified on the CPU backend only, then the numbers that go with it.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash profiles/run_r05_first_call.sh'
# 1. the whole GPU suite (tests/test_gpu_zz_marg_batch.py last: MarginalizationBatch in its final form, the lock-step replay with shared
#    marginalizations)
# 2. tracking.txt of the device-resident tracker against the reference tracker's goldens ON THE DEVICE (round 4 compared it on the CPU backend)
# 3. MarginalizationBatch throughput (steady state: one batch object, three passes) at 16 / 64 / 256 windows, with the phase split
# 4. lock-step replay of 8 estimators in one group with and without ICG_LOCKSTEP_MARG_BATCH=1
# 5. the driver's command (engine_twin and marg.batched in the line)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5first
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/gputests.txt; cat $O/gputests.txt
timeout 200 python - > $O/device_log.txt 2>&1 <<'PY'
import ref_tracking_utils as rt, harness as H
for name in ("c1_640x480_100", "c1_lost_histgate", "c1_slow_second_new", "c2_long_60", "c4_1920x1080_500"):
    rt.compare_scenario(H.HOST_LIB, name, engine="device", with_log=True)
    print("ok", name, flush=True)
PY
tail -6 $O/device_log.txt
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
