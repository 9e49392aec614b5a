#!/bin/bash
# Round 6, call 30: k_preint (the tree: 237 VGPRs, 2 waves per SIMD) against builds capped at 3 and 4 waves per SIMD (168 / 128 VGPRs, 102 / 196 registers spilled)
# : the preintegration tests on the tree's library, then the C4 preintegration leg of bench.py
# (3 840 intervals x 40 samples, Earth variant) for each library, three times interleaved
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c30
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 900 python -m pytest tests/test_gpu_preint.py tests/test_gpu_c4.py tests/test_gpu_backend.py -m gpu -q -x 2>&1 | tail -6 | tee $O/tests.txt
cp ic-gvins_amd/libicgvins_hip.so $O/tree.so
run() {
  tag=$1; lib=$2
  cp $lib ic-gvins_amd/libicgvins_hip.so
  timeout 300 python - <<PY
import json, sys
sys.path.insert(0, "$R/tests"); sys.path.insert(0, "$R/ic-gvins_amd")
import icgvins, preint_data as pdz
b = pdz.bench_block(icgvins, 0, n_streams=256, n_intervals=15, n_samples=40)
b2 = pdz.bench_block(icgvins, 0, n_streams=16, n_intervals=15, n_samples=40)
b3 = pdz.bench_block(icgvins, 0, n_streams=1, n_intervals=1, n_samples=200)
print("$tag", "3840 x 40:", b["kernel_us"], "us", round(b["value"] / 1e6, 1), "M samples/s | 240 x 40:", b2["kernel_us"], "us | 1 x 200:", b3["kernel_us"], "us")
PY
}
V=ic-gvins_amd/_variants
for rep in a b c; do
  run tree_$rep $O/tree.so
  run w3_$rep $V/libicgvins_hip_w3.so
  run w4_$rep $V/libicgvins_hip_w4.so
done
cp $O/tree.so ic-gvins_amd/libicgvins_hip.so; rm -f $O/tree.so
