#!/bin/bash
# kernel trace of a short default-configuration front-end run + the queue-level analysis (profiles/analyze_trace.py)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace --output-format csv -d /tmp/kts -o kt -- python $R/bench.py --no-cpu-baseline --no-reproj --no-profile-pass --steps 40 --warmup 10 > $OUT/trace_short_bench.json 2> $OUT/trace_short.err
f=$(find /tmp/kts -name "*kernel_trace.csv" | head -1)
head -1 "$f" > $OUT/trace_header.txt
python $R/profiles/analyze_trace.py "$f" > $OUT/${1:-r02}_queue_view.json
tail -c 300 $OUT/trace_short.err
