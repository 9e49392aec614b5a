#!/bin/bash
# Round 6, call 31: the detection mask as a bit plane per frame (k_mask_bits draws the discs once per frame, k_min_eig_nms reads two words
# per block row instead of testing the disc list per block; the tree's library) against the library of the commit before (c28): detector and
# tracker tests, the driver's shape interleaved, then the SQ instruction counters of the tree
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c31
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 900 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_device_tracker.py tests/test_gpu_stream.py tests/test_parity_at_scale.py tests/test_gpu_c4.py -m gpu -q -x 2>&1 | tail -6 | tee $O/tests.txt
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
print("$tag", d["value"], d["ms_per_step"], (d.get("parity") or {}).get("ok"), {n: round(v["avg_us"], 1) for n, v in k.items() if n in ("lk_track_fb", "detect_mask_bits", "clahe_apply", "detect_min_eig_nms")}, "alone:", {n: ce.get(n, {}).get("exclusive_us_per_launch") for n in ("detect_mask_bits", "detect_min_eig_nms")})
PY
}
V=ic-gvins_amd/_variants
run tree_a $O/tree.so
run c28_a $V/libicgvins_hip_c28.so
run tree_b $O/tree.so
run c28_b $V/libicgvins_hip_c28.so
run tree_c $O/tree.so
run c28_c $V/libicgvins_hip_c28.so
SHORT="--no-cpu-baseline --no-reproj --no-engine-twin --prime 24 --warmup 2 --steps 6 --no-profile-pass --no-dist --no-c4 --no-replay"
pmc() {
  tag=$1; lib=$2
  cp $lib $R/ic-gvins_amd/libicgvins_hip.so
  (cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_$tag -o p -- python $R/bench.py $SHORT > /dev/null 2> $O/pmc_$tag.err)
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob("$O/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in ("k_min_eig_nms", "k_mask_bits"):
    if k in acc:
        print("$tag", k, {c: round(v / cnt[k][c] / 1e6, 3) for c, v in acc[k].items()}, "launches", cnt[k]["SQ_WAVES"])
PY
  rm -rf $O/pmc_$tag
}
pmc tree $O/tree.so
cp $O/tree.so ic-gvins_amd/libicgvins_hip.so; rm -f $O/tree.so
