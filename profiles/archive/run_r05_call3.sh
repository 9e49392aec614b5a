#!/bin/bash
# Round 5, call 3: the row-streaming pyrDown kernel (k_pyrdown_rows) against the tile kernel (ICG_PYRAMID_TILES=1), parity first.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c3
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 400 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_stream.py tests/test_gpu_c4.py tests/test_gpu_device_tracker.py tests/test_parity_at_scale.py -m gpu -x -q 2>&1 | tail -8 > $O/gputests.txt; cat $O/gputests.txt
LIGHT="--gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --no-reproj --no-parity --no-engine-twin"
ICG_PYRAMID_TILES=1 timeout 300 python bench.py $LIGHT --details $O/tiles_details.json > $O/tiles_line.json 2> $O/tiles.err
timeout 300 python bench.py $LIGHT --details $O/rows_details.json > $O/rows_line.json 2> $O/rows.err
timeout 300 python bench.py $LIGHT --engine device --details $O/rows_dev_details.json > $O/rows_dev_line.json 2> $O/rows_dev.err
python - <<PY
import json
for tag in ("tiles", "rows", "rows_dev"):
    try:
        d = json.loads(open("$O/%s_line.json" % tag).read().strip().splitlines()[-1])
        dd = json.load(open("$O/%s_details.json" % tag))
        kc = dd.get("kernel_ceiling") or {}
        print(tag, d["value"], d["config"]["engine"], d["host"].get("cpu_cores_busy"), kc.get("exclusive_us_per_frame"), kc.get("ceiling_frames_per_s"))
        print("  ", {k: [v.get("launches_per_step"), round(v["exclusive_us_per_launch"], 1)] for k, v in (kc.get("kernels") or {}).items()})
    except Exception as e:
        print(tag, "failed", e)
PY
