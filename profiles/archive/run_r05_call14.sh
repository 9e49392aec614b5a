#!/bin/bash
# Round 5, call 14: second pass of the stage diet (wave-wide map look-ups, window keeper an entry per lane, parallel sweep, the three gather loops four chunks at a time) + the reaper thread of MarginalizationBatch: tests, light bench, marks, marg phases.
# list / corner assembly / ROI table, list ranking by pointer jumping, parallax terms read out of registers, block histogram in LDS):
# device-tracker and stream tests, light bench on the product library (exclusive stage times), the marks again on the profiling build;
# MarginalizationBatch phases with glibc's allocator defaults against the host layer's policy, and on one host thread.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c14
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 500 python -m pytest tests/test_gpu_device_tracker.py tests/test_gpu_stream.py tests/test_gpu_frontend.py tests/test_parity_at_scale.py -m gpu -x -q 2>&1 | tail -6 > $O/gputests.txt; cat $O/gputests.txt
LIGHT="--gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --no-reproj --no-parity --no-engine-twin"
timeout 300 python bench.py $LIGHT --details $O/dev_details.json > $O/dev_line.json 2> $O/dev.err
cp ic-gvins_amd/libicgvins_hip.so /tmp/libicgvins_hip_keep.so
cp ic-gvins_amd/_variants/libicgvins_hip_tctiming.so ic-gvins_amd/libicgvins_hip.so
timeout 300 python bench.py $LIGHT --details $O/timing_details.json > $O/timing_line.json 2> $O/timing.err
cp /tmp/libicgvins_hip_keep.so ic-gvins_amd/libicgvins_hip.so
grep "tc timing" $O/timing.err > $O/tc_timing.txt
for mode in "on 0"; do set -- $mode
  echo "== allocator policy $1, ICG_SOLVER_THREADS=$2 (0 = default)" >> $O/marg_phases.txt
  if [ "$2" = "0" ]; then unset ICG_SOLVER_THREADS; else export ICG_SOLVER_THREADS=$2; fi
  ICG_HOST_MALLOC_POLICY=$1 ICG_MARG_DEBUG=1 timeout 120 python profiles/marg_batch_probe.py --windows 64,256 2>&1 | grep -v "^\[schur\]" | tail -8 >> $O/marg_phases.txt
done
unset ICG_SOLVER_THREADS
python - <<PY
import json
d = json.loads(open("$O/dev_line.json").read().strip().splitlines()[-1]); dd = json.load(open("$O/dev_details.json")); kc = dd.get("kernel_ceiling") or {}
print(d["value"], d.get("value_200steps"), d["host"], kc.get("exclusive_us_per_frame"))
print({k: round(v["exclusive_us_per_launch"], 1) for k, v in (kc.get("kernels") or {}).items()})
PY
cat $O/marg_phases.txt; cat $O/tc_timing.txt
