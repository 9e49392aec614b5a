#!/bin/bash
# Round 6, call 28 (as calls 21 and 26): the final tree — counters and kernel trace re-collected (collect.sh r06), the fresh
# counter summary put where bench.py reads it, then the final-tree evidence run (run_r06_final.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 1500 bash profiles/collect.sh r06 2>&1 | tail -12
cp gpurun_out/r06_pmc_summary.json profiles/r06_pmc_summary.json
cd $R
timeout 2400 bash profiles/r06_calls/run_r06_final.sh
