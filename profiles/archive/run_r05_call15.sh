#!/bin/bash
# Round 5, call 15 (diagnostics): where WindowSolverBatch and the 64-estimator lock-step replay spend their wall time
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c15
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
ICG_SOLVER_DEBUG=1 timeout 200 python profiles/run_solve_batch_only.py > $O/solve.out 2> $O/solve.err
grep -v "^$" $O/solve.err | tail -40; cat $O/solve.out
ICG_GVINS_DEBUG=1 timeout 300 python profiles/run_lockstep_probe.py 64 4 > $O/lock.out 2> $O/lock.err
cat $O/lock.out
grep "gvins-phase" $O/lock.err | awk '{k=$2; for(i=3;i<NF-1;i++) k=k" "$i; s[k]+=$(NF-1); n[k]++} END {for (k in s) printf "%-26s %10.1f ms over %d estimators\n", k, s[k], n[k]}' | sort > $O/lock_phases.txt
cat $O/lock_phases.txt
grep -c . $O/lock.err
timeout 300 python profiles/run_lockstep_probe.py 64 16 > $O/lock16.out 2>/dev/null; cat $O/lock16.out
rm -f $O/lock.err
