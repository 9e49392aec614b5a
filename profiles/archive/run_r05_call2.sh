#!/bin/bash
# Round 5, call 2: the LK instruction diet + full-exec epilogue, k_subpix sums on all lanes, wave-cooperative RANSAC subset draws.
#   gpurun --timeout 900 -- 'bash profiles/run_r05_call2.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c2
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
# 1. isolated LK launch, base (HEAD) vs new: same output hash, kernel time
ICG_LK_LIB=$R/ic-gvins_amd/_variants/libicgvins_hip_base.so timeout 120 python profiles/run_lk_only.py 32 560 5 > $O/lk_only_base.txt 2>&1
timeout 120 python profiles/run_lk_only.py 32 560 5 > $O/lk_only_new.txt 2>&1
ICG_LK_LIB=$R/ic-gvins_amd/_variants/libicgvins_hip_base.so timeout 120 python profiles/run_lk_only.py 64 300 5 >> $O/lk_only_base.txt 2>&1
timeout 120 python profiles/run_lk_only.py 64 300 5 >> $O/lk_only_new.txt 2>&1
tail -4 $O/lk_only_base.txt; tail -4 $O/lk_only_new.txt
# 2. parity
timeout 400 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_geometry.py tests/test_gpu_device_tracker.py tests/test_gpu_stream.py tests/test_gpu_c4.py tests/test_parity_at_scale.py -m gpu -x -q 2>&1 | tail -8 > $O/gputests.txt; cat $O/gputests.txt
# 3. the driver's command
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/driver_details.json > $O/driver_line.json 2> $O/driver.err
python - <<PY
import json
d = json.loads(open("$O/driver_line.json").read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(d["value"], d["config"]["engine"], d["host"], (d.get("parity") or {}).get("ok"), (d.get("engine_twin") or {}).get("value"), r.get("exclusive_us"), r.get("exclusive_us_per_frame_all_kernels"), r.get("ceiling_frames_per_s"))
PY
python - <<PY
import json
d = json.load(open("$O/driver_details.json"))
kc = d.get("kernel_ceiling") or {}
print(json.dumps(kc.get("per_kernel_us_per_frame") or kc, indent=0)[:1500])
PY
