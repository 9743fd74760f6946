#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database.
Usage: rocpd_pmc.py results.db [out.md]   (counter values are per dispatch; FETCH_SIZE and
WRITE_SIZE are in KiB -- MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads HALF the bytes of a
wide coalesced read stream, so the corrected read estimate is 2x)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    rows = c.execute(
        "select kernel_name, counter_name, count(*), avg(value), sum(value), avg(duration) "
        "from counters_collection group by kernel_name, counter_name "
        "order by sum(value) desc").fetchall()
    lines = ["| kernel | counter | dispatches | avg per dispatch | avg dispatch us |",
             "|---|---|---|---|---|"]
    for name, counter, n, avg, tot, dur in rows:
        short = name if len(name) <= 90 else name[:87] + "..."
        lines.append("| `%s` | %s | %d | %.1f | %.1f |" % (short, counter, n, avg, (dur or 0) / 1e3))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
