#!/bin/bash
# gpurun --timeout 300 -- 'bash profiles/ubench/run.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $R/gpurun_out/ubench
cd $R/profiles/ubench
UB_FROM=${UB_FROM:-add_u32} timeout 180 ./valu_cost > $R/gpurun_out/ubench/valu_cost.txt 2>&1
cat $R/gpurun_out/ubench/valu_cost.txt
