#!/bin/bash
# Round 6, call 11: the LK launch on a lowest-priority stream of its own (ICG_LK_STREAM_PRIORITY=low), or the group's
# stream raised above it (=high), against the default — same box, interleaved, 100 timed steps each
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c11
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
LIGHT="--gpus 1 --steps 100 --warmup 10 --no-reproj --no-c4 --no-engine-twin --no-cpu-baseline --no-parity --no-replay"
run() {
  tag=$1; shift
  env "$@" timeout 400 python bench.py $LIGHT --details $O/${tag}_details.json > $O/${tag}_line.json 2> $O/${tag}.err
  python3 - <<PY
import json
d = json.loads(open("$O/${tag}_line.json").read().strip().splitlines()[-1])
k = (d.get("kernels") or {})
print("$tag", d["value"], d["ms_per_step"], {n: (round(v.get("avg_us", 0), 1)) for n, v in k.items() if n in ("lk_track_fb", "clahe_apply", "min_eig_nms")} if k else "")
PY
}
run default1 ICG_X=0
run low1 ICG_LK_STREAM_PRIORITY=low
run high1 ICG_LK_STREAM_PRIORITY=high
run default2 ICG_X=0
run low2 ICG_LK_STREAM_PRIORITY=low
run high2 ICG_LK_STREAM_PRIORITY=high
