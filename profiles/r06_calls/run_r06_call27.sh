#!/bin/bash
# Round 6, call 27: the new detector test (row slivers between disc bands) on the tree's library, and the split of the block detector's
# vector instructions: SQ counters of the tree against a variant that leaves every block after the mask phase (-DFE_MASK_ONLY: no parity,
# instruction accounting only)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c27
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 900 python -m pytest tests/test_gpu_geometry.py -m gpu -q -x 2>&1 | tail -6 | tee $O/tests.txt
cp ic-gvins_amd/libicgvins_hip.so $O/tree.so
SHORT="--no-cpu-baseline --no-reproj --no-engine-twin --no-parity --prime 24 --warmup 2 --steps 6 --no-profile-pass --no-dist --no-c4 --no-replay"
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
for k in ("k_min_eig_nms", "k_select", "k_subpix"):
    if k in acc:
        print("$tag", k, {c: round(v / cnt[k][c] / 1e6, 3) for c, v in acc[k].items()}, "launches", cnt[k]["SQ_WAVES"])
PY
  rm -rf $O/pmc_$tag
}
pmc tree $O/tree.so
pmc maskonly ic-gvins_amd/_variants/libicgvins_hip_maskonly.so
cp $O/tree.so ic-gvins_amd/libicgvins_hip.so; rm -f $O/tree.so
