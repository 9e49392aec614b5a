#!/bin/bash
# Stall / LDS counters of the front-end kernels at the bench configuration (one short run, one counter group): where the wave-cycles
# that do not issue an instruction go.  profiles/collect_stalls.sh r02  -> gpurun_out/<tag>_pmc_stalls.json
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="--no-cpu-baseline --no-reproj --prime 24 --warmup 2 --steps 6 --no-profile-pass"
timeout 280 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/${TAG}_pmc_st -o p -- python $R/bench.py $SHORT > /dev/null 2> $OUT/${TAG}_pmc_st.err
timeout 280 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/${TAG}_pmc_st2 -o p -- python $R/bench.py $SHORT > /dev/null 2> $OUT/${TAG}_pmc_st2.err
python $R/profiles/summarize_pmc.py $OUT/${TAG}_pmc_st $OUT/${TAG}_pmc_st2 > $OUT/${TAG}_pmc_stalls.json
tail -2 $OUT/${TAG}_pmc_st.err $OUT/${TAG}_pmc_st2.err
rm -rf $OUT/${TAG}_pmc_st $OUT/${TAG}_pmc_st2
