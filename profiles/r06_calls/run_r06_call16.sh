#!/bin/bash
# Round 6, call 16: GPU suite + marg probe after the reaper thread went
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c16
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/gputests.txt; cat $O/gputests.txt
ICG_MARG_DEBUG=1 timeout 200 python profiles/marg_batch_probe.py $O/probe.json > $O/probe.out 2> $O/probe.err
cat $O/probe.json; grep "batch\] 256" $O/probe.err | tail -2
