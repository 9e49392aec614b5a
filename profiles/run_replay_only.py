"""The f2 workload alone (one synthetic GNSS + IMU + camera sequence through icg::GVINS on the HIP-backed host layer), for
`rocprofv3 --kernel-trace --stats -- python profiles/run_replay_only.py`.  ICG_GVINS_DEBUG=1 prints the estimator's log."""
import ctypes as C
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
import gvins_checks as gc  # noqa: E402
import gvins_data as gd  # noqa: E402
import harness as H  # noqa: E402

lib = C.CDLL(H.TOOLS_LIB)
seq = gd.Sequence(lib)
files = seq.write(tempfile.mkdtemp(prefix="prof_replay_"))
for _ in range(3):
    S = gc.run_replay(lib, files)
    print({k: S[k] for k in ("data_seconds", "wall_seconds", "frames_tracked", "keyframes", "optimizations", "marginalizations", "ins_launches")})
