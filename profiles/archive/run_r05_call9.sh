#!/bin/bash
# Round 5, call 9: the driver's command on the current tree (all new blocks: value_200steps, ranks, c4.solve_batched / marg_batched, replay.lockstep64,
# roofline.trk_stage on the twin), then the same on the device engine and with every thread confined to 2 / 1 CPUs (profiles/r05_cpu_quota.md)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c9
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/driver_details.json > $O/driver_line.json 2> $O/driver.err
tail -3 $O/driver.err
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --engine device --no-reproj --no-cpu-baseline --no-engine-twin --details $O/device_details.json > $O/device_line.json 2> $O/device.err
ICG_BENCH_TIMED_CPUS=2 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --engine device --no-reproj --no-cpu-baseline --no-profile-pass --no-parity --no-engine-twin --details $O/q2_details.json > $O/q2_line.json 2> $O/q2.err
ICG_BENCH_TIMED_CPUS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --engine device --no-reproj --no-cpu-baseline --no-profile-pass --no-parity --no-engine-twin --details $O/q1_details.json > $O/q1_line.json 2> $O/q1.err
ICG_BENCH_TIMED_CPUS=2 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --engine table --no-reproj --no-cpu-baseline --no-profile-pass --no-parity --no-engine-twin --details $O/q2t_details.json > $O/q2t_line.json 2> $O/q2t.err
python - <<PY
import json
for f in ("driver", "device", "q2", "q1", "q2t"):
    try:
        d = json.loads(open("$O/%s_line.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d.get("value_200steps"), d["config"]["engine"][:12], d["config"]["groups_per_gpu"], d["host"], (d.get("parity") or {}).get("ok"))
    except Exception as e:
        print(f, "failed", e)
d = json.loads(open("$O/driver_line.json").read().strip().splitlines()[-1])
print(json.dumps({k: d.get(k) for k in ("engine_twin", "c4", "marg", "solve", "replay", "reproj")})[:3000])
print(json.dumps(d.get("roofline"))[:1500])
print(len(open("$O/driver_line.json").read()))
PY
