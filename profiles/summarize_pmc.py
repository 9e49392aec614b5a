#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (counter_collection.csv under each given directory).

Output JSON: {kernel: {"launches": n, "<COUNTER>": mean per launch, "grid_threads": mean grid size, ...}}.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; "hbm_bytes_per_launch" applies the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request: doubled; WRITE_SIZE taken as is, uncalibrated)."""
import collections
import csv
import glob
import json
import sys


def main(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.Counter())
    grid = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k][r["Counter_Name"]] += 1
                grid[k].append(int(r["Grid_Size"]))
    out = {}
    for k in sorted(acc):
        e = {}
        for c, v in acc[k].items():
            e[c] = v / cnt[k][c]
            e.setdefault("launches", cnt[k][c])
        e["grid_threads"] = sum(grid[k]) / len(grid[k])
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = 2.0 * 1024.0 * e.get("FETCH_SIZE", 0.0) + 1024.0 * e.get("WRITE_SIZE", 0.0)
        out[k] = e
    # provenance (bench.py checks it before it quotes these counters next to a fresh measurement): a hash of the kernel sources the
    # counters were collected on and the launch shape of the bench configuration (streams per launch)
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha1()
    per_file = {}
    src = os.path.join(root, "ic-gvins_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h")):
            data = open(os.path.join(src, name), "rb").read()
            h.update(name.encode())
            h.update(data)
            per_file[name] = hashlib.sha1(data).hexdigest()
    # (round 5) the tracker kernel's stage bodies live in the host layer's header, compiled into tracker.hip: part of that kernel's sources
    core = os.path.join(root, "ic-gvins_amd", "host", "track_core.h")
    if os.path.exists(core):
        data = open(core, "rb").read()
        h.update(b"../host/track_core.h")
        h.update(data)
        per_file["../host/track_core.h"] = hashlib.sha1(data).hexdigest()
    lk = out.get("k_lk_track_fb")
    pre = out.get("k_pyramid3")
    out["_meta"] = {"csrc_sha1": h.hexdigest(), "csrc_files": per_file, "streams_per_launch": os.environ.get("ICG_PMC_STREAMS_PER_LAUNCH"),
                    "lk_points_per_launch": (lk["grid_threads"] / 64.0) if lk else None,
                    # segmented launches (device-resident tracker) size the grid by capacity: the points actually tracked per launch, from the
                    # bench line of the same configuration (roofline.units_per_launch)
                    "lk_active_points_per_launch": float(os.environ["ICG_PMC_LK_ACTIVE_POINTS"]) if os.environ.get("ICG_PMC_LK_ACTIVE_POINTS") else None,
                    "note": "csrc_sha1 = sha1 over the names and contents of ic-gvins_amd/csrc/*.{hip,h} and host/track_core.h at collection time; csrc_files = the sha1 of each "
                            "of them (bench.py accepts the summary for a kernel while the files that kernel is compiled from are unchanged)"}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
