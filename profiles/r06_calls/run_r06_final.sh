#!/bin/bash
# Round 6, final tree: the whole GPU suite, smoke(), the driver's command exactly as the driver runs it, and the kernel trace of the headline region alone
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6final
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/gputests.txt; cat $O/gputests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/driver_details.json > $O/driver_line.json 2> $O/driver.err
python3 - <<PY
import json
d = json.loads([l for l in open("$O/driver_line.json").read().splitlines() if l.startswith("{")][-1])
print(d["value"], d.get("value_200steps"), d["exchange"], (d.get("parity") or {}).get("ok"), d["host"])
print(json.dumps(d.get("roofline"))[:1500])
print(json.dumps({k: d.get(k) for k in ("engine_twin", "solve", "marg", "replay", "pcie_inclusive", "cpu_baseline", "cpu_baseline_allcores")})[:2500])
print(json.dumps(d.get("c4"))[:2000])
print(len(open("$O/driver_line.json").read()))
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reproj --no-parity --no-engine-twin --no-profile-pass --details $O/hr_details.json > $O/hr_line.json 2> $O/hr_err.txt
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_headline_region.csv \;
rm -rf $O/kt
python3 - <<PY
import csv, json
d = json.loads([l for l in open("$O/hr_line.json").read().splitlines() if l.startswith("{")][-1])
print("bench under the tracer:", d["value"], d["ms_per_step"])
for r in list(csv.DictReader(open("$O/kernel_stats_headline_region.csv")))[:10]:
    print(r["Name"][:36], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
