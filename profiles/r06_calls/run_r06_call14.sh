#!/bin/bash
# Round 6, call 14: MarginalizationBatch phase split at 256 C2 windows on the round-6 tree (allocator policy default = glibc's, and =raise)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c14
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
for pol in default raise; do
  ICG_HOST_MALLOC_POLICY=$pol ICG_MARG_DEBUG=1 timeout 200 python profiles/marg_batch_probe.py $O/probe_$pol.json > $O/probe_$pol.out 2> $O/probe_$pol.err
  echo "policy $pol"; cat $O/probe_$pol.json; grep "batch\] 256" $O/probe_$pol.err | tail -3
done
