#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace into a per-kernel stats table (what
`--stats` prints as CSV in older rocprofv3 builds).  Usage: rocpd_stats.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute(
        "select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by %s order by sum(end-start) desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |",
             "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) <= 110 else name[:107] + "..."
        lines.append("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.2f |" % (
            short, n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    text = "\n".join(lines) + "\n\ntotal kernel time: %.3f ms over %d dispatches\n" % (
        total / 1e6, sum(r[1] for r in rows))
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
