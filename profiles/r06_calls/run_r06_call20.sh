#!/bin/bash
# Round 6, call 20: k_min_eig_nms with resident workgroups walking the tile grid at a fixed stride (no counters; the tree's library:
# 192 workgroups per XCD; variants: 96 and 384) against the one-workgroup-per-block launch (product variant) — detector tests first,
# then the driver's shape interleaved, 100 timed steps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c20
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 900 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_device_tracker.py tests/test_gpu_stream.py tests/test_parity_at_scale.py -m gpu -q -x 2>&1 | tail -6 | tee $O/tests.txt
LIGHT="--gpus 1 --steps 100 --warmup 10 --no-reproj --no-c4 --no-engine-twin --no-cpu-baseline --no-replay --no-dist"
cp ic-gvins_amd/libicgvins_hip.so $O/tree.so
run() {
  tag=$1; lib=$2
  cp $lib ic-gvins_amd/libicgvins_hip.so
  timeout 500 python bench.py $LIGHT --details $O/${tag}_details.json > $O/${tag}_line.json 2> $O/${tag}.err
  python3 - <<PY
import json
d = json.loads([l for l in open("$O/${tag}_line.json").read().splitlines() if l.startswith("{")][-1])
dd = json.load(open("$O/${tag}_details.json"))
k = dd.get("kernels") or {}
ce = (dd.get("kernel_ceiling") or {}).get("kernels", {}).get("detect_min_eig_nms", {})
print("$tag", d["value"], d["ms_per_step"], (d.get("parity") or {}).get("ok"), {n: round(v["avg_us"], 1) for n, v in k.items() if n in ("lk_track_fb", "clahe_apply", "detect_min_eig_nms", "detect_select")}, "alone:", ce.get("exclusive_us_per_launch"))
PY
}
run res192_a $O/tree.so
run product_a ic-gvins_amd/_variants/libicgvins_hip_product.so
run res96_a ic-gvins_amd/_variants/libicgvins_hip_res96.so
run res384_a ic-gvins_amd/_variants/libicgvins_hip_res384.so
run res192_b $O/tree.so
run product_b ic-gvins_amd/_variants/libicgvins_hip_product.so
cp $O/tree.so ic-gvins_amd/libicgvins_hip.so; rm -f $O/tree.so
