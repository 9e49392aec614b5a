#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4d
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_device_tracker.py -m gpu -x -q 2>&1 | tail -30 > $O/devtrk.txt; cat $O/devtrk.txt
