#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) kernel trace: per-kernel calls/avg/total and GPU busy (union) time."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    kd = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch_%'")][0]
    ks = kd.replace("rocpd_kernel_dispatch_", "rocpd_info_kernel_symbol_")
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    stats = {}
    for name, a, b in rows:
        name = name.split("(")[0]
        e = stats.setdefault(name, [0, 0])
        e[0] += 1
        e[1] += b - a
    total = sum(v[1] for v in stats.values())
    # union of busy intervals
    busy, cur_a, cur_b = 0, None, None
    for _, a, b in rows:
        if cur_a is None:
            cur_a, cur_b = a, b
        elif a <= cur_b:
            cur_b = max(cur_b, b)
        else:
            busy += cur_b - cur_a
            cur_a, cur_b = a, b
    if cur_a is not None:
        busy += cur_b - cur_a
    span = rows[-1][2] - rows[0][1] if rows else 0
    lines = ["kernel,calls,total_ms,avg_us,pct_of_kernel_time"]
    for name, (n, t) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{name},{n},{t / 1e6:.3f},{t / n / 1e3:.3f},{100.0 * t / max(1, total):.2f}")
    lines.append(f"# sum_of_kernel_time_ms={total / 1e6:.3f} gpu_busy_union_ms={busy / 1e6:.3f} first_to_last_span_ms={span / 1e6:.3f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
