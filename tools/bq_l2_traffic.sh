#!/bin/bash
# L2-side traffic of the bucket ball query (bqb_query_kernel): requests the vector L1s send to the L2
# and what the L2 fetches from memory, per launch.  tools/bq_l2_traffic.sh > gpurun_out/bq_l2.md
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bql2a /tmp/bql2b
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d /tmp/bql2a -o r -- python3 $ROOT/tools/bq_ab.py > /tmp/bql2a.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TOTAL_CACHE_ACCESSES_sum -d /tmp/bql2b -o r -- python3 $ROOT/tools/bq_ab.py > /tmp/bql2b.log 2>&1
cd $ROOT
python3 - $(find /tmp/bql2a -name "*.db" | head -1) $(find /tmp/bql2b -name "*.db" | head -1) <<'PY'
import sqlite3, sys
agg = {}
for db in sys.argv[1:]:
    try:
        c = sqlite3.connect(db).cursor()
        for name, grid, cn, v, dur in c.execute(
                "select kernel_name, grid_size, counter_name, avg(value), avg(duration) from "
                "counters_collection where kernel_name like '%bqb_query%' group by kernel_name, "
                "grid_size, counter_name"):
            agg.setdefault(grid, {"dur": dur})[cn] = v
    except Exception as e:
        print("(", db, e, ")")
print("| grid | us (profiled) | counter | per launch |")
print("|---|---|---|---|")
for g, d in sorted(agg.items()):
    for k, v in sorted(d.items()):
        if k != "dur":
            print("| %d | %.1f | %s | %.0f |" % (g, d["dur"] / 1e3, k, v))
PY
tail -3 /tmp/bql2a.log /tmp/bql2b.log 1>&2
