#!/bin/bash
# Round 6, call 17: occupancy of k_lk_track_fb now that it has no scratch — the product (96 VGPRs, 5 waves per SIMD) against variants capped
# at 4 waves per SIMD (amdgpu_waves_per_eu(4,4): 128 VGPRs free per SIMD for the other groups' kernels) and pushed to 6 (80 VGPRs, 68 B of
# scratch per lane); variant libraries built in ic-gvins_amd/_variants (git-ignored), swapped in on the box's scratch copy; interleaved
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c17
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
print("$tag", d["value"], d["ms_per_step"], {n: round(v["avg_us"], 1) for n, v in k.items() if n in ("lk_track_fb", "clahe_apply", "detect_min_eig_nms", "fm_ransac_sets")})
PY
}
run w5_a $O/product.so
run max4_a ic-gvins_amd/_variants/libicgvins_hip_lkmax4.so
run w6_a ic-gvins_amd/_variants/libicgvins_hip_lk6.so
run w5_b $O/product.so
run max4_b ic-gvins_amd/_variants/libicgvins_hip_lkmax4.so
run w6_b ic-gvins_amd/_variants/libicgvins_hip_lk6.so
cp $O/product.so ic-gvins_amd/libicgvins_hip.so; rm -f $O/product.so
