R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o sb -- python $R/profiles/run_solve_batch_only.py > $OUT/sb.out 2> $OUT/sb.err
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
echo "file: $f"
if [ -n "$f" ]; then cp "$f" $OUT/sb_kernel_stats.csv; fi
tail -3 $OUT/sb.out; tail -3 $OUT/sb.err
