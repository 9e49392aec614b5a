#!/bin/bash
# Race detection for the host layer (not part of pytest: ~10 minutes).  Builds the checker's all-in-one host library (product host layer +
# tools on the CPU shim of the C ABI) with -fsanitize=$SAN in a scratch copy and runs, under ThreadSanitizer:
#   1. block_pool.cc        BlockPool / PoolAllocator: allocate on one thread, free on another
#   2. frontend_groups.cc   StreamGroups: 6 streams in 3 groups x 2 host threads, 30 frames each (worker threads, HostPool, pooled objects)
#   3. icg_replay_oracle    one estimator with the host-factor helper thread (WindowSolver::setHostFactorOverlap)
#   4. icg_replay_oracle --streams 3 --lockstep-groups 1   WindowSolverBatch with its persistent pool
#   7. marg_batch.cc        MarginalizationBatch: 12 windows x 3 passes, per-window phases on 4 pool threads (round 4)
#   8. icg_replay_oracle --streams 3 --lockstep-groups 1 with ICG_LOCKSTEP_MARG_BATCH=1: the marginalizations of a tick in one batch (round 4)
# Expected: 0 "WARNING: ThreadSanitizer" in every log (round 2: all four clean).
# SAN=address,undefined tests/tsan/run.sh /tmp/icg_asan runs the same four under AddressSanitizer + UBSan, plus 5. below (round 3: all clean).
set -eu
SAN=${SAN:-thread}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=${1:-/tmp/icg_tsan}
rm -rf $W && mkdir -p $W/src
python3 $ROOT/tests/tsan/make_inputs.py $W > $W/inputs.txt
read CFG IMU GNSS IMG < $W/inputs.txt
cp -r $ROOT/include $ROOT/ic-gvins_amd $ROOT/oracle $W/src/
find $W/src -name "*.o" -delete; rm -f $W/src/oracle/*.so $W/src/oracle/icg_replay_oracle
make -s -C $W/src/oracle -j8 CXXFLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-psabi -pthread -fsanitize=$SAN" liboracle.so libicgvins_host_oracle.so icg_replay_oracle
export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" ASAN_OPTIONS="detect_leaks=0"
g++ -std=c++17 -O1 -g -fsanitize=$SAN -pthread $ROOT/tests/tsan/block_pool.cc -o $W/block_pool && $W/block_pool > $W/1_block_pool.log 2>&1
g++ -std=c++17 -O1 -g -fsanitize=$SAN -pthread $ROOT/tests/tsan/frontend_groups.cc -o $W/frontend_groups -L$W/src/oracle -licgvins_host_oracle -loracle -Wl,-rpath,$W/src/oracle
$W/frontend_groups > $W/2_frontend_groups.log 2>&1
(cd $W/src/oracle && LD_LIBRARY_PATH=. ./icg_replay_oracle --config $CFG --imu $IMU --gnss $GNSS --images $IMG --output $W/out1 > $W/3_replay_single.log 2>&1)
mkdir -p $W/out3
(cd $W/src/oracle && LD_LIBRARY_PATH=. ./icg_replay_oracle --config $CFG --imu $IMU --gnss $GNSS --images $IMG --output $W/out3 --streams 3 --lockstep-groups 1 > $W/4_replay_lockstep.log 2>&1)
# 8. (round 4) the lock-step replay with the marginalizations of a tick shared through one MarginalizationBatch
mkdir -p $W/out8
(cd $W/src/oracle && ICG_LOCKSTEP_MARG_BATCH=1 LD_LIBRARY_PATH=. ./icg_replay_oracle --config $CFG --imu $IMU --gnss $GNSS --images $IMG --output $W/out8 --streams 3 --lockstep-groups 1 > $W/8_replay_lockstep_marg.log 2>&1)
# 7. (round 4) MarginalizationBatch: 12 windows (one on the dense path) x 3 passes on one batch object, per-window phases on 4 pool threads
g++ -std=c++17 -O1 -g -fsanitize=$SAN -pthread $ROOT/tests/tsan/marg_batch.cc -o $W/marg_batch -L$W/src/oracle -licgvins_host_oracle -loracle -Wl,-rpath,$W/src/oracle
$W/marg_batch $W/marg_problem.bin > $W/7_marg_batch.log 2>&1
# 5. (address/undefined only) the lazily built object view of the track table and its write-back — culling and window refinement through
#    the C entry points on both engines — driven from python with the sanitizer runtime preloaded
case "$SAN" in *address*)
  LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) PYTHONPATH=$ROOT/ic-gvins_amd:$ROOT:$ROOT/tests python3 - > $W/5_object_view.log 2>&1 <<PY
import oracle_lib, cull_checks as cc, refine_checks as rc
LIB = "$W/src/oracle/libicgvins_host_oracle.so"
oracle = oracle_lib.load()
for eng in ("table", "object"):
    cc.check_window_culling(LIB, oracle, engine=eng)
    rc.check_refinement(LIB, n_frames=24, engine=eng)
print("done")
PY
  # 6. (round 4) the tracker core — fixed-capacity blocks, the source of the device-resident tracker's stage kernels — as the host engine
  #    ("core") and behind the tracker ABI's CPU backend ("device": download / import / view / absorb / export / upload, history drains with
  #    a low threshold): culling, refinement and a 70-frame stream whose window rolls over, under AddressSanitizer + UBSan
  ICG_TRACKER_LOG_DRAIN=30 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) PYTHONPATH=$ROOT/ic-gvins_amd:$ROOT:$ROOT/tests python3 - > $W/6_tracker_core.log 2>&1 <<PY
import oracle_lib, cull_checks as cc, refine_checks as rc
import ctypes as C, numpy as np, harness as H
LIB = "$W/src/oracle/libicgvins_host_oracle.so"
oracle = oracle_lib.load()
for eng in ("core", "device"):
    cc.check_window_culling(LIB, oracle, engine=eng)
    rc.check_refinement(LIB, n_frames=24, engine=eng)
    w, h = 640, 480
    cam = H.camera_for(w, h)
    sb = H.StreamBatch(LIB, 2, w, h, cam, max_features=100, engine=eng)
    scene = H.SynthScene(sb.lib, w, h, cam, tex_size=1024, threads=2)
    for k in range(70):
        frames = [scene.render(k, stream=50 + s) if not (30 <= k < 33) else np.full((h, w), 90, np.uint8) for s in range(2)]
        poses = np.stack([H.pose12(*scene.ins_pose(k, stream=50 + s)) for s in range(2)])
        sb.step([f.ctypes.data for f in frames], w, np.full(2, 100.0 + k / 20.0), poses)
        if k % 9 == 0:
            sb.dump(0, 0), sb.features(1)
    sb.close()
print("done")
PY
  ;;
esac
for f in $W/[1-8]_*.log; do echo "$(basename $f): $(grep -c 'WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error' $f || true) reports; $(tail -1 $f | cut -c1-140)"; done
