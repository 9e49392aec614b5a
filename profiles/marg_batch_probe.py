"""Throughput of MarginalizationBatch (host/marg_batch.h) against one MarginalizationInfo::marginalization() after the other, on the product
libraries.  bench.py runs this file as a child process for its marg.batched block (a block outside the headline path must not be able to
take the bench line down); `python profiles/marg_batch_probe.py [out.json] [--windows 16,64,256] [--lm 500 --kf 15]` runs it alone.
ICG_PROBE_HOST_LIB: another build of the host layer (the oracle-backed one for a CPU dry run)."""
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


def measure(windows, n_lm=300, n_kf=10):
    hl = C.CDLL(os.environ.get("ICG_PROBE_HOST_LIB") or H.HOST_LIB)
    Pm = md.make_problem(n_lm=n_lm, n_kf=n_kf, seed=2)  # (300, 10): the C2 window of the bench's marg block; (500, 15): C4
    out = {"factors_per_window": int(Pm["obs"].shape[1])}
    bu.backend_marginalize_batch(hl, Pm, 8, 0)  # (contexts, pool, code paged in)
    for nmb in windows:
        one = bu.backend_marginalize_batch(hl, Pm, nmb, 1, reps=2)
        bat = bu.backend_marginalize_batch(hl, Pm, nmb, 0, reps=3)
        scale = np.abs(one["Hp"]).max(axis=(1, 2))
        out[str(nmb)] = {"batch_ms": round(bat["seconds"] * 1e3, 3), "one_by_one_ms": round(one["seconds"] * 1e3, 3),
                         "windows_per_s": round(nmb / bat["seconds"], 1), "windows_per_s_one_by_one": round(nmb / one["seconds"], 1),
                         "structured_dense": [bat["structured"], bat["dense"]],
                         "max_rel_diff_Hp": float((np.abs(bat["Hp"] - one["Hp"]).max(axis=(1, 2)) / scale).max())}
    return out


if __name__ == "__main__":
    argv = sys.argv[1:]
    windows = (16, 64, 256)
    if "--windows" in argv:
        k = argv.index("--windows")
        windows = tuple(int(x) for x in argv[k + 1].split(","))
        del argv[k:k + 2]
    shape = {}
    for flag, key in (("--lm", "n_lm"), ("--kf", "n_kf")):
        if flag in argv:
            k = argv.index(flag)
            shape[key] = int(argv[k + 1])
            del argv[k:k + 2]
    txt = json.dumps(measure(windows, **shape))
    print(txt)
    if argv:
        open(argv[0], "w").write(txt + "\n")
