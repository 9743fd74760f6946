#!/usr/bin/env python3
"""Per-kernel time inside ONE steady-state step of a rocprofv3 rocpd trace: a window between
two consecutive launches of a marker kernel (default: the large-scene FPS kernel, once per
step) -- the one with the MEDIAN wall time among the windows shorter than 1.5x the shortest
(warm-up and bench.py's fully instrumented detail steps are longer).
Usage: rocpd_step.py results.db [marker-substring] [out.md] [markers-per-step]"""
import os
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else "fps_bucket_kernel"
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    marks = [s for (n, s, e) in rows if marker in n]
    per_step = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    marks = marks[::per_step]
    if len(marks) < 2:
        raise SystemExit("marker %r seen %d times" % (marker, len(marks)))
    wins = sorted((b - a, a, b) for a, b in zip(marks[:-1], marks[1:]))
    if os.environ.get("ROCPD_WINDOW") != "median":   # (median of ALL windows: pipelined loop,
        wins = [w for w in wins if w[0] <= 1.5 * wins[0][0]]   # whose shortest window is the prologue)
    _, t0, t1 = wins[len(wins) // 2]
    agg = {}
    for n, s, e in rows:
        if t0 <= s < t1:
            a = agg.setdefault(n, [0, 0, 0])
            a[0] += 1
            a[1] += e - s
            a[2] = max(a[2], e - s)
    busy = sum(v[1] for v in agg.values())
    lines = ["step window: %.3f ms wall, %.3f ms kernel-busy, %d dispatches" % (
        (t1 - t0) / 1e6, busy / 1e6, sum(v[0] for v in agg.values())), "",
        "| kernel | calls | total us | % of busy | longest call us |", "|---|---|---|---|---|"]
    for n, (cnt, tot, longest) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = n if len(n) <= 100 else n[:97] + "..."
        lines.append("| `%s` | %d | %.1f | %.2f | %.1f |" % (short, cnt, tot / 1e3,
                                                            100.0 * tot / busy, longest / 1e3))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
