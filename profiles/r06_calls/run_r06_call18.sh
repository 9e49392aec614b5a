#!/bin/bash
# Round 6, call 18: does a streaming kernel that fits the 32 VGPRs five LK waves leave free on a SIMD (512 - 5 x 96) overlap better?
# k_pyrdown_rows capped at 32 VGPRs (34 in the product; no spill) against the product, interleaved; per-kernel HIP-event durations in the timed region
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c18
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
LIGHT="--gpus 1 --steps 100 --warmup 10 --no-reproj --no-c4 --no-engine-twin --no-cpu-baseline --no-parity --no-replay --no-dist"
cp ic-gvins_amd/libicgvins_hip.so $O/product.so
run() {
  tag=$1; lib=$2
  cp $lib ic-gvins_amd/libicgvins_hip.so
  timeout 400 python bench.py $LIGHT --details $O/${tag}_details.json > $O/${tag}_line.json 2> $O/${tag}.err
  python3 - <<PY
import json
d = json.loads([l for l in open("$O/${tag}_line.json").read().splitlines() if l.startswith("{")][-1])
k = json.load(open("$O/${tag}_details.json")).get("kernels") or {}
print("$tag", d["value"], d["ms_per_step"], {n: round(v["avg_us"], 1) for n, v in k.items() if n in ("lk_track_fb", "clahe_apply", "clahe_lut", "pyrdown_rows", "detect_min_eig_nms")})
PY
}
run product_a $O/product.so
run pyr32_a ic-gvins_amd/_variants/libicgvins_hip_pyr32.so
run product_b $O/product.so
run pyr32_b ic-gvins_amd/_variants/libicgvins_hip_pyr32.so
cp $O/product.so ic-gvins_amd/libicgvins_hip.so; rm -f $O/product.so
