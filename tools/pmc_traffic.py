#!/usr/bin/env python3
"""Combine the FETCH_SIZE and WRITE_SIZE rocprofv3 passes into profiles/pmc_traffic.json:
HBM bytes per launch for the hand-written kernels, keyed by (kernel, grid size) so that the
SA1-sized launches can be told apart.  FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950
FETCH_SIZE counts a wide coalesced read stream at half its bytes (MI355X_MICROARCH.md, HBM),
so `read_bytes_corrected` = 2 x FETCH_SIZE x 1024 is the upper estimate and
`read_bytes_raw` the lower one.
The workload the passes ran (bench.py --workload / --points / --batch) is recorded too: bench.py
quotes a counter only on the launch shape it was measured on.
Usage: pmc_traffic.py fetch.db write.db out.json [workload points batch]"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(path):
    c = sqlite3.connect(path).cursor()
    out = {}
    for name, grid, n, avg in c.execute(
            "select kernel_name, grid_size, count(*), avg(value) from counters_collection "
            "where kernel_name like '%btr::%' group by kernel_name, grid_size"):
        short = name.split("(")[0].replace("void ", "")
        out[(short, int(grid))] = (int(n), float(avg))
    return out


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    res = {}
    for key in sorted(set(fetch) | set(write)):
        f = fetch.get(key, (0, 0.0))
        w = write.get(key, (0, 0.0))
        res["%s@grid%d" % key] = {
            "dispatches": max(f[0], w[0]),
            "read_bytes_raw": f[1] * 1024.0,
            "read_bytes_corrected": 2.0 * f[1] * 1024.0,
            "write_bytes": w[1] * 1024.0,
        }
    from backtoreality_amd import build
    wl = sys.argv[4] if len(sys.argv) > 4 else "fsb"
    points = int(sys.argv[5]) if len(sys.argv) > 5 else 40000
    batch = int(sys.argv[6]) if len(sys.argv) > 6 else 8
    json.dump({"build_id": build.build_id(),
               "workload": {"workload": wl, "points": points, "batch": batch},
               "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate "
                         "passes) -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline",
               "kernels": res}, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
