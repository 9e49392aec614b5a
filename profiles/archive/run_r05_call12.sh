#!/bin/bash
# (profiling build: `make -C ic-gvins_amd/csrc timing` before the call)
# Round 5, call 12: (a) the whole GPU suite on the tree with the restructured symmetricEigen (host) and the fused list compaction of the tracker
# stages, (b) the driver's command, (c) where the tracker stages' latency chains spend their time: the profiling build of the library
# (-DTC_TIMING: wall-clock marks inside the stage bodies, host/track_core.h) under a light bench.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c12
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/gputests.txt; cat $O/gputests.txt
timeout 800 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/driver_details.json > $O/driver_line.json 2> $O/driver.err
LIGHT="--gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --no-reproj --no-parity --no-engine-twin"
cp ic-gvins_amd/libicgvins_hip.so /tmp/libicgvins_hip_keep.so
cp ic-gvins_amd/_variants/libicgvins_hip_tctiming.so ic-gvins_amd/libicgvins_hip.so
timeout 300 python bench.py $LIGHT --details $O/timing_details.json > $O/timing_line.json 2> $O/timing.err
cp /tmp/libicgvins_hip_keep.so ic-gvins_amd/libicgvins_hip.so
grep "tc timing" $O/timing.err > $O/tc_timing.txt; wc -l $O/tc_timing.txt
python - <<PY
import json
d = json.loads(open("$O/driver_line.json").read().strip().splitlines()[-1])
print(d["value"], d.get("value_200steps"), d["config"]["engine"][:14], d["host"], (d.get("parity") or {}).get("ok"))
r = d["roofline"]; print(r["exclusive_us_per_frame_all_kernels"], r["trk_stage"]["exclusive_us_per_frame"], r["frac_whole_path"])
print(json.dumps({k: d.get(k) for k in ("marg", "solve", "replay")})[:1500]); print(json.dumps(d["c4"])[:900])
dd = json.load(open("$O/driver_details.json")); kc = dd.get("kernel_ceiling") or {}
print({k: round(v["exclusive_us_per_launch"], 1) for k, v in (kc.get("kernels") or {}).items()})
PY
cat $O/tc_timing.txt
