#!/bin/bash
# Round 6, call 22: k_min_eig_nms with one wave per 60 x 64 BLOCK (one mask pass per block, runs of needed rows streamed in chunks; the
# tree's library: 8-row chunks, 192 workgroups per XCD) against the resident 60 x 16 tile form of the commit before (tile192); variants:
# 576 workgroups per XCD, no cap (one block per wave), 4-row chunks (93 VGPRs, 5 waves per SIMD instead of 106 / 4)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c22
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 900 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_device_tracker.py tests/test_gpu_stream.py tests/test_parity_at_scale.py tests/test_gpu_frontend.py -m gpu -q -x 2>&1 | tail -6 | tee $O/tests.txt
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
V=ic-gvins_amd/_variants
run tile192_a $V/libicgvins_hip_tile192.so
run blk192_a $O/tree.so
run blkall_a $V/libicgvins_hip_blkall.so
run blk576_a $V/libicgvins_hip_blk576.so
run blkch4_192_a $V/libicgvins_hip_blkch4_192.so
run blkch4_all_a $V/libicgvins_hip_blkch4_all.so
run tile192_b $V/libicgvins_hip_tile192.so
run blk192_b $O/tree.so
cp $O/tree.so ic-gvins_amd/libicgvins_hip.so; rm -f $O/tree.so
