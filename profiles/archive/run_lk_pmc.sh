#!/bin/bash
# Counters of the isolated LK launch (profiles/run_lk_only.py 32 560) for both lane mappings (ICG_LK_PAIR=0: one feature per wave,
# 1: two features per wave): instruction counts, issue activity, stalls.  -> gpurun_out/<tag>_lk_pmc_pair{0,1}.json
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export ICG_LK_PAIR=$v
  timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_lk_a$v -o p -- python $R/profiles/run_lk_only.py 32 560 3 > /dev/null 2> $OUT/${TAG}_lk_a$v.err
  timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d $OUT/${TAG}_lk_b$v -o p -- python $R/profiles/run_lk_only.py 32 560 3 > /dev/null 2> $OUT/${TAG}_lk_b$v.err
  timeout 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES --output-format csv -d $OUT/${TAG}_lk_c$v -o p -- python $R/profiles/run_lk_only.py 32 560 3 > /dev/null 2> $OUT/${TAG}_lk_c$v.err
  python $R/profiles/summarize_pmc.py $OUT/${TAG}_lk_a$v $OUT/${TAG}_lk_b$v $OUT/${TAG}_lk_c$v > $OUT/${TAG}_lk_pmc_pair$v.json
  tail -1 $OUT/${TAG}_lk_a$v.err
  rm -rf $OUT/${TAG}_lk_a$v $OUT/${TAG}_lk_b$v $OUT/${TAG}_lk_c$v
done
