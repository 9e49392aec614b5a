#!/usr/bin/env python3
"""Queue-level view of a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): how busy every hardware queue is, how many kernels run side by
side, and how long each kernel takes while the others run.  Restricted to the densest part of the run (the quarter of the k_lk_track_fb
dispatches that lie closest together = the steady front-end).  usage: analyze_trace.py <kernel_trace.csv>  -> JSON on stdout"""
import csv
import json
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], r.get("Queue_Id", "0")))
rows.sort()
lk = [r for r in rows if r[2].startswith("k_lk_track_fb")]
if len(lk) < 40:
    print(json.dumps({"error": "too few k_lk_track_fb dispatches", "n": len(lk)}))
    sys.exit(0)
# the densest run of consecutive k_lk_track_fb dispatches (a quarter of them, at most 2500): the steady state of the main front-end block
N = max(20, min(2500, len(lk) // 4))
best = min(range(len(lk) - N), key=lambda i: lk[i + N][0] - lk[i][0])
t0, t1 = lk[best][0], lk[best + N][0]
win = [r for r in rows if r[0] >= t0 and r[1] <= t1]
span = float(t1 - t0)
per_q = defaultdict(list)
per_k = defaultdict(list)
for s, e, name, q in win:
    per_q[q].append((s, e))
    per_k[name].append(e - s)
queues = {}
for q, iv in per_q.items():
    iv.sort()
    busy = sum(e - s for s, e in iv)
    gaps = [iv[i + 1][0] - iv[i][1] for i in range(len(iv) - 1)]
    queues[q] = {"dispatches": len(iv), "busy_frac": round(busy / span, 3), "median_gap_us": round(sorted(gaps)[len(gaps) // 2] / 1e3, 1) if gaps else None}
# concurrency profile: sweep over start/end events
ev = sorted([(s, 1) for s, e, _, _ in win] + [(e, -1) for s, e, _, _ in win])
cur, last, hist = 0, t0, defaultdict(float)
for t, d in ev:
    hist[cur] += t - last
    cur += d
    last = t
tot = sum(hist.values())
# how often the chip-filling kernel runs at all (round 5): LK dispatches in flight over the window
evl = sorted([(s, 1) for s, e, nm, _ in win if nm.startswith("k_lk_track_fb")] + [(e, -1) for s, e, nm, _ in win if nm.startswith("k_lk_track_fb")])
curl, lastl, histl = 0, t0, defaultdict(float)
for t, d in evl:
    histl[curl] += t - lastl
    curl += d
    lastl = t
totl = sum(histl.values()) or 1.0
kernels = {k: {"n": len(v), "mean_us": round(sum(v) / len(v) / 1e3, 1), "share_of_queue_time": round(sum(v) / sum(sum(x) for x in per_k.values()), 3)}
           for k, v in sorted(per_k.items(), key=lambda kv: -sum(kv[1]))}
out = {"window_ms": round(span / 1e6, 2), "lk_launches_in_window": sum(1 for r in win if r[2].startswith("k_lk_track_fb")),
       "queues_seen": len(queues), "mean_queue_busy_frac": round(sum(q["busy_frac"] for q in queues.values()) / max(1, len(queues)), 3),
       "mean_kernels_in_flight": round(sum(k * v for k, v in hist.items()) / tot, 2),
       "in_flight_histogram": {str(k): round(v / tot, 3) for k, v in sorted(hist.items())},
       "lk_in_flight_histogram": {str(k): round(v / totl, 3) for k, v in sorted(histl.items())},
       "queues": queues, "kernels": kernels}
print(json.dumps(out, indent=1))
