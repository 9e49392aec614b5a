#!/bin/bash
# Profile collection for one round (run on the GPU box through gpurun):
#   profiles/collect.sh r01
# 1) rocprofv3 --kernel-trace --stats of the DEFAULT bench.py command  -> gpurun_out/<tag>_kt/
# 2) separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ issue counters) of a short bench.py run -> gpurun_out/<tag>_pmc_*/
# (never --pmc together with a trace domain other than kernel dispatches; one counter group per pass)
# Copy the summaries produced by profiles/summarize_pmc.py and the *_kernel_stats.csv into profiles/ afterwards.
#   profiles/collect.sh r05 table     (round 5: a second argument names a NON-default engine; outputs carry it as a suffix, the kernel
#                                      trace of the full bench is skipped for it)
set -u
TAG=${1:-r02}
ENGINE=${2:-}
ESUF=""
EARG=""
if [ -n "$ENGINE" ]; then ESUF="_$ENGINE"; EARG="--engine $ENGINE"; fi
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -z "$ENGINE" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o kt -- python $R/bench.py --no-cpu-baseline --no-replay --details $OUT/${TAG}${ESUF}_kt_bench_details.json > $OUT/${TAG}${ESUF}_kt_bench.json 2> $OUT/${TAG}${ESUF}_kt.err
else
python $R/bench.py $EARG --no-cpu-baseline --no-reproj --no-parity --no-engine-twin --steps 60 --warmup 10 --details $OUT/${TAG}${ESUF}_kt_bench_details.json > $OUT/${TAG}${ESUF}_kt_bench.json 2> $OUT/${TAG}${ESUF}_kt.err
fi
# the counter passes run the BENCH CONFIGURATION (default streams / groups of this box), shortened: 24 priming + 2 warm-up + 6 timed frames per stream
SHORT="$EARG --no-cpu-baseline --no-reproj --no-engine-twin --prime 24 --warmup 2 --steps 6 --no-profile-pass"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}${ESUF}_pmc_fetch -o p -- python $R/bench.py $SHORT > /dev/null 2> $OUT/${TAG}${ESUF}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}${ESUF}_pmc_write -o p -- python $R/bench.py $SHORT > /dev/null 2> $OUT/${TAG}${ESUF}_pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/${TAG}${ESUF}_pmc_sq -o p -- python $R/bench.py $SHORT > /dev/null 2> $OUT/${TAG}${ESUF}_pmc_sq.err
# k_reproj_eval (the `reproj` block's launch shape) in its own passes
for CNT in $( [ -z "$ENGINE" ] && echo FETCH_SIZE WRITE_SIZE ); do
  rocprofv3 --pmc $CNT --output-format csv -d $OUT/${TAG}_pmc_reproj_$CNT -o p -- python $R/profiles/run_reproj_only.py > /dev/null 2> $OUT/${TAG}_pmc_reproj_$CNT.err
done
# (the contract line is the LAST line that starts with "{": native libraries may print banners before it)
export ICG_PMC_STREAMS_PER_LAUNCH=$(python -c "import json; c=json.loads([l for l in open('$OUT/${TAG}${ESUF}_kt_bench.json').read().splitlines() if l.startswith('{')][-1])['config']; print(c['streams_per_gpu'] / c['groups_per_gpu'])")
export ICG_PMC_LK_ACTIVE_POINTS=$(python -c "import json; print(json.load(open('$OUT/${TAG}${ESUF}_kt_bench_details.json'))['roofline']['units_per_launch'])")
python $R/profiles/summarize_pmc.py $OUT/${TAG}${ESUF}_pmc_fetch $OUT/${TAG}${ESUF}_pmc_write $OUT/${TAG}${ESUF}_pmc_sq $OUT/${TAG}_pmc_reproj_FETCH_SIZE $OUT/${TAG}_pmc_reproj_WRITE_SIZE > $OUT/${TAG}_pmc_summary${ESUF}.json
[ -d $OUT/${TAG}_kt ] && find $OUT/${TAG}_kt -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
# queue-level view of the same trace (hardware-queue occupancy, kernels in flight, per-kernel duration under load)
KT=$( [ -d $OUT/${TAG}_kt ] && find $OUT/${TAG}_kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python $R/profiles/analyze_trace.py "$KT" > $OUT/${TAG}_queue_view.json
ls -la $OUT | grep ${TAG}
# the raw per-dispatch tables are large: keep only the summaries (gpurun merges at most 64 MiB back)
rm -rf $OUT/${TAG}${ESUF}_pmc_fetch $OUT/${TAG}${ESUF}_pmc_write $OUT/${TAG}${ESUF}_pmc_sq $OUT/${TAG}_pmc_reproj_FETCH_SIZE $OUT/${TAG}_pmc_reproj_WRITE_SIZE $OUT/${TAG}_kt
