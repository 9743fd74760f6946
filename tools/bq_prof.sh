#!/bin/bash
# per-kernel time of the SA1 ball query, bucket path and grid path (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rm -rf /tmp/bqp$mode
  BTR_BQ_BUCKETS=$mode rocprofv3 --kernel-trace --stats -d /tmp/bqp$mode -o r -- python3 $GRAFT_REPO_ROOT/tools/bq_ab.py > /dev/null 2>&1
  DB=$(find /tmp/bqp$mode -name "*.db" | head -1)
  echo "== BTR_BQ_BUCKETS=$mode"
  python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB | grep -i "bq\|bqb" | cut -c1-160
done
