#!/usr/bin/env python3
"""Dispatch-by-dispatch timeline of ONE steady-state step of a rocprofv3 rocpd trace (same
window choice as rocpd_step.py): start offset, duration, gap to the previous end on any queue,
queue id, kernel.  Usage: rocpd_timeline.py results.db [marker-substring] [out.txt]"""
import re
import os
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else "fps_bucket_kernel"
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    sel = "select %s, start, end%s from kernels order by start" % (
        name_col, (", " + qcol) if qcol else "")
    rows = c.execute(sel).fetchall()
    marks = [r[1] for r in rows if marker in r[0]]
    wins = sorted((b - a, a, b) for a, b in zip(marks[:-1], marks[1:]))
    if os.environ.get("ROCPD_WINDOW") != "median":   # (median of ALL windows: pipelined loop,
        wins = [w for w in wins if w[0] <= 1.5 * wins[0][0]]   # whose shortest window is the prologue)
    _, t0, t1 = wins[len(wins) // 2]
    out = ["step window %.3f ms" % ((t1 - t0) / 1e6), "start_us   dur_us   idle_us  queue  kernel"]
    last_end = t0
    idle_total = 0
    for r in rows:
        n, s, e = r[0], r[1], r[2]
        if not (t0 <= s < t1):
            continue
        q = r[3] if qcol else 0
        idle = max(0, s - last_end)
        idle_total += idle
        short = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", ""))
        short = short.replace("void ", "").replace("at::native::", "")[:90]
        out.append("%8.1f %8.1f %8.1f  %5s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, idle / 1e3,
                                                   q, short))
        last_end = max(last_end, e)
    out.append("idle (no kernel on any queue): %.3f ms" % (idle_total / 1e6))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
