#!/bin/bash
# Round-4 evidence, one gpurun call: the driver's command (device engine, all blocks, parity witness), the same command on the track table,
# the CPU-quota rows, then profiles/collect.sh (rocprofv3 kernel stats + queue view + counter passes of the default command).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4final
mkdir -p $O
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/driver_details.json > $O/driver_line.json 2> $O/driver.err
Q="--no-reproj --no-cpu-baseline --no-profile-pass --no-parity"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $Q --engine table --details $O/table20_details.json > $O/table20.line 2> $O/table20.err
timeout 300 python bench.py --steps 200 --warmup 30 $Q --details $O/dev200_details.json > $O/dev200.line 2> $O/dev200.err
timeout 300 python bench.py --steps 200 --warmup 30 $Q --engine table --details $O/table200_details.json > $O/table200.line 2> $O/table200.err
for N in 1 2; do
  ICG_BENCH_TIMED_CPUS=$N timeout 300 python bench.py --steps 60 --warmup 5 $Q --details $O/q${N}_dev_details.json > $O/q${N}_dev.line 2> $O/q${N}_dev.err
done
ICG_BENCH_TIMED_CPUS=2 timeout 300 python bench.py --steps 60 --warmup 5 $Q --engine table --groups 8 --details $O/q2_table_details.json > $O/q2_table.line 2> $O/q2_table.err
ICG_HOST_PROF=cpu ICG_BENCH_TIMED_CPUS=2 timeout 300 python bench.py --steps 60 --warmup 5 $Q --details $O/q2_prof_details.json > $O/q2_prof.line 2> $O/q2_prof.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.line")) + ["$O/driver_line.json"]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["steps"], d["config"]["groups_per_gpu"], d["config"]["streams_per_gpu"], d["config"]["engine"][:12], d["host"], (d.get("parity") or {}).get("ok"))
    except Exception as e:
        print(f, "failed", e)
PY
bash profiles/collect.sh r04 > $O/collect.log 2>&1
tail -5 $O/collect.log
