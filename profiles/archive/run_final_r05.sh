#!/bin/bash
# Round 5, final tree: the whole GPU suite, smoke(), and the driver's command exactly as the driver runs it (device-resident tracker by default)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5final
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/gputests.txt; cat $O/gputests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 800 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/driver_details.json > $O/driver_line.json 2> $O/driver.err
python - <<PY
import json
d = json.loads(open("$O/driver_line.json").read().strip().splitlines()[-1])
print(d["value"], d.get("value_200steps"), d["config"]["engine"][:14], d["config"]["groups_per_gpu"], d["host"], (d.get("parity") or {}).get("ok"))
print(json.dumps(d.get("roofline"))[:1800])
print(json.dumps({k: d.get(k) for k in ("engine_twin", "c4", "marg", "solve", "replay", "cpu_baseline", "cpu_baseline_allcores", "speedup_vs_cpu_baseline", "pcie_inclusive", "ms_per_step")})[:3500])
print(len(open("$O/driver_line.json").read()))
PY
