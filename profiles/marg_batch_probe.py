"""Throughput of MarginalizationBatch against one MarginalizationInfo::marginalization() after the other on the product libraries
(host/marg_batch.h; the bench's marg.batched block runs the same measurement).  usage: python profiles/marg_batch_probe.py [out.json]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "ic-gvins_amd"), ROOT]
import backend_utils as bu  # noqa: E402
import harness as H  # noqa: E402
import marg_data as md  # noqa: E402

hl = C.CDLL(os.environ.get("ICG_PROBE_HOST_LIB") or H.HOST_LIB)
Pm = md.make_problem(n_lm=300, n_kf=10, seed=2)
out = {"factors_per_window": int(Pm["obs"].shape[1])}
bu.backend_marginalize_batch(hl, Pm, 8, 0)
for nmb in (16, 64, 256):
    one = bu.backend_marginalize_batch(hl, Pm, nmb, 1, reps=2)
    bat = bu.backend_marginalize_batch(hl, Pm, nmb, 0, reps=3)
    scale = np.abs(one["Hp"]).max(axis=(1, 2))
    out[str(nmb)] = {"batch_ms": round(bat["seconds"] * 1e3, 3), "one_by_one_ms": round(one["seconds"] * 1e3, 3),
                     "windows_per_s": round(nmb / bat["seconds"], 1), "windows_per_s_one_by_one": round(nmb / one["seconds"], 1),
                     "structured_dense": [bat["structured"], bat["dense"]],
                     "max_rel_diff_Hp": float((np.abs(bat["Hp"] - one["Hp"]).max(axis=(1, 2)) / scale).max())}
txt = json.dumps(out)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt + "\n")
