#!/bin/bash
# Round 5, call 4: CLAHE diet (float LUT in LDS + magic rounding in k_clahe_apply; fixed lane mapping + DPP scan in k_clahe_lut), parity first,
# then light benches: default library, LK at 4 and 6 waves per SIMD (variant libraries swapped in on the box).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c4
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 400 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_stream.py tests/test_gpu_c4.py tests/test_parity_at_scale.py -m gpu -x -q 2>&1 | tail -8 > $O/gputests.txt; cat $O/gputests.txt
LIGHT="--gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --no-reproj --no-parity --no-engine-twin"
timeout 300 python bench.py $LIGHT --details $O/new_details.json > $O/new_line.json 2> $O/new.err
cp ic-gvins_amd/libicgvins_hip.so /tmp/libicgvins_hip_keep.so
for v in lkw4 lkw6; do
  cp ic-gvins_amd/_variants/libicgvins_hip_$v.so ic-gvins_amd/libicgvins_hip.so
  timeout 300 python bench.py $LIGHT --details $O/${v}_details.json > $O/${v}_line.json 2> $O/$v.err
done
cp /tmp/libicgvins_hip_keep.so ic-gvins_amd/libicgvins_hip.so
timeout 300 python bench.py $LIGHT --engine device --details $O/new_dev_details.json > $O/new_dev_line.json 2> $O/new_dev.err
python - <<PY
import json
for tag in ("new", "lkw4", "lkw6", "new_dev"):
    try:
        d = json.loads(open("$O/%s_line.json" % tag).read().strip().splitlines()[-1])
        dd = json.load(open("$O/%s_details.json" % tag))
        kc = dd.get("kernel_ceiling") or {}
        print(tag, d["value"], d["config"]["engine"], d["host"].get("cpu_cores_busy"), kc.get("exclusive_us_per_frame"), kc.get("ceiling_frames_per_s"))
        print("  ", {k: [v.get("launches_per_step"), round(v["exclusive_us_per_launch"], 1)] for k, v in (kc.get("kernels") or {}).items()})
    except Exception as e:
        print(tag, "failed", e)
PY
