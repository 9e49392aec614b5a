#!/bin/bash
# Experiments queued at the end of round 3 (the round's GPU minutes ran out): one gpurun call, ~5 GPU-minutes.  Every bench run prints
# one line; the summary at the end lists value / cores busy.  (A bench run needs ~30 s on a fresh box: timeouts are 90 s.)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
Q="--steps 60 --warmup 5 --no-reproj --no-cpu-baseline --no-profile-pass --no-parity"
# (a) wider jobs: does 12 x 128 / 16 x 96 reach the 116 k of the 128-wide device-only replay?
timeout 120 python bench.py $Q --groups 12 --streams 1536 > $O/w12x128.json 2> $O/w12x128.err
timeout 120 python bench.py $Q --groups 16 --streams 1536 > $O/w16x96.json 2> $O/w16x96.err
# (b) one rank's share of a 16-CPU / 8-GPU box (2 CPUs, every thread confined): coarser polls, fewer group threads
for W in 100 300 1000; do
  ICG_WAIT_MODE=poll:$W ICG_BENCH_TIMED_CPUS=2 timeout 90 python bench.py $Q --groups 8 --streams 768 > $O/q2_poll$W.json 2> $O/q2_poll$W.err
done
ICG_BENCH_TIMED_CPUS=2 timeout 90 python bench.py $Q --groups 4 --streams 768 > $O/q2_g4.json 2> $O/q2_g4.err
python - <<PY
import json
for n in ("w12x128", "w16x96", "q2_poll100", "q2_poll300", "q2_poll1000", "q2_g4"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], (d.get("host") or {}).get("cpu_cores_busy"))
    except Exception as e:
        print(n, "failed", e)
PY
