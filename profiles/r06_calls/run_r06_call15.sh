#!/bin/bash
# Round 6, call 15: where the 13 ms outside the four phases of a 256-window MarginalizationBatch go (layout: factor upload, partition, assembly plan)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c15
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
ICG_ABI_DEBUG=1 ICG_MARG_DEBUG=1 timeout 200 python profiles/marg_batch_probe.py $O/probe.json --windows 256 > $O/probe.out 2> $O/probe.err
cat $O/probe.json; grep "batch\] 256\|set_windows\] W=256" $O/probe.err | tail -6
