#!/bin/bash
# Round 6, call 12: RCCL exchange on one GPU (test + driver command), C4 roofline / parity witness / preint baseline in the line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c12
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -q -x -k "rccl" 2>&1 | tail -15 > $O/rccl_test.txt; cat $O/rccl_test.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/driver_details.json > $O/driver_line.json 2> $O/driver.err
tail -3 $O/driver.err
python3 - <<PY
import json
d = json.loads(open("$O/driver_line.json").read().strip().splitlines()[-1])
print(d["value"], d.get("value_200steps"), d["exchange"], d.get("ranks"))
print(d["solve"]["batched"], d["marg"])
print(json.dumps(d["c4"])[:1500])
PY
