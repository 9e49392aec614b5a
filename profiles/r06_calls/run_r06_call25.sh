#!/bin/bash
# Round 6, call 25: the Gauss-Newton loop of k_lk_track_fb with the float convergence guard (FP64 only inside a band around eps^2) and
# inline-zero accumulators — front-end tests on the tree's library, then the driver's shape against the library of the commit before (c24)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c25
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_device_tracker.py tests/test_gpu_stream.py tests/test_parity_at_scale.py tests/test_gpu_c4.py -m gpu -q -x 2>&1 | tail -6 | tee $O/tests.txt
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
print("$tag", d["value"], d["ms_per_step"], (d.get("parity") or {}).get("ok"), {n: round(v["avg_us"], 1) for n, v in k.items() if n in ("lk_track_fb", "lk_finish", "clahe_apply", "detect_min_eig_nms")}, "alone:", {n: ce.get(n, {}).get("exclusive_us_per_launch") for n in ("lk_track_fb", "detect_min_eig_nms")})
PY
}
V=ic-gvins_amd/_variants
run tree_a $O/tree.so
run c24_a $V/libicgvins_hip_c24.so
run tree_b $O/tree.so
run c24_b $V/libicgvins_hip_c24.so
run tree_c $O/tree.so
run c24_c $V/libicgvins_hip_c24.so
cp $O/tree.so ic-gvins_amd/libicgvins_hip.so; rm -f $O/tree.so
