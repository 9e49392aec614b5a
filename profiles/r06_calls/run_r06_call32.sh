#!/bin/bash
# Round 6, call 32: compile-time variants of the final kernels against the tree's library, the driver's shape, interleaved:
#   lkw6     k_lk_track_fb built for 6 waves per SIMD (80 VGPRs, 14 spilled: 60 B of scratch)
#   detr160 / detr320   k_min_eig_nms with 160 (its residency at 91 VGPRs: 5 waves per SIMD) and 320 workgroups per XCD instead of 192
#   detg3 / detg10      runs of rows closed over gaps of up to 3 and up to 10 masked rows instead of 6
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c32
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
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
ce = (dd.get("kernel_ceiling") or {}).get("kernels", {})
print("$tag", d["value"], d["ms_per_step"], (d.get("parity") or {}).get("ok"), {n: round(v["avg_us"], 1) for n, v in k.items() if n in ("lk_track_fb", "detect_min_eig_nms")}, "alone:", {n: ce.get(n, {}).get("exclusive_us_per_launch") for n in ("lk_track_fb", "detect_min_eig_nms")})
PY
}
V=ic-gvins_amd/_variants
for rep in a b; do
  run tree_$rep $O/tree.so
  run lkw6_$rep $V/libicgvins_hip_lkw6.so
  run detr160_$rep $V/libicgvins_hip_detr160.so
  run detr320_$rep $V/libicgvins_hip_detr320.so
  run detg3_$rep $V/libicgvins_hip_detg3.so
  run detg10_$rep $V/libicgvins_hip_detg10.so
done
run tree_c $O/tree.so
cp $O/tree.so ic-gvins_amd/libicgvins_hip.so; rm -f $O/tree.so
