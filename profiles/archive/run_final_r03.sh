#!/bin/bash
# Last GPU call of round 3 (≈4 GPU-minutes): the whole -m gpu suite on the final host layer, one rank's share of a 16-core / 8-GPU box
# (2 cores, the world-8 host plan: 6 groups x 128 streams), and the kernel-only ceiling at 8 / 32 / 128 streams per launch.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
timeout 150 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc $?" ; tail -2 $O/gpu_tests.txt
Q="--steps 60 --warmup 5 --no-reproj --no-cpu-baseline --no-profile-pass --no-parity"
ICG_BENCH_TIMED_CPUS=2 timeout 100 python bench.py $Q --groups 6 --streams 768 --details $O/q2_6x128_details.json > $O/q2_6x128.json 2> $O/q2_6x128.err
ICG_BENCH_TIMED_CPUS=2 timeout 100 python bench.py $Q --groups 8 --streams 768 --details $O/q2_8x96_details.json > $O/q2_8x96.json 2> $O/q2_8x96.err
C="--steps 20 --warmup 5 --no-reproj --no-cpu-baseline --no-parity"
timeout 80 python bench.py $C --groups 12 --streams 96 --details $O/ceil_8_details.json > $O/ceil_8.json 2> $O/ceil_8.err
timeout 80 python bench.py $C --groups 12 --streams 384 --details $O/ceil_32_details.json > $O/ceil_32.json 2> $O/ceil_32.err
timeout 100 python bench.py $C --groups 6 --streams 768 --details $O/ceil_128_details.json > $O/ceil_128.json 2> $O/ceil_128.err
python - <<PY
import json, os
for n in ("q2_6x128", "q2_8x96", "ceil_8", "ceil_32", "ceil_128"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(n, d["value"], (d.get("host") or {}).get("cpu_cores_busy"), r.get("exclusive_us_per_frame_all_kernels"), r.get("ceiling_frames_per_s"))
    except Exception as e:
        print(n, "failed", e)
PY
