#!/bin/bash
# The ball query over the FPS buckets, first form against the variants of the second
# (csrc/ball_query_bucket.hip, BTR_BQ_FORM): stand-alone times at four shapes, checksums, and the
# kernel's own duration from a rocprofv3 trace.   tools/bq_form_ab.sh > gpurun_out/bq_form_ab.txt
cd ${GRAFT_REPO_ROOT:-.}
for f in 1 2; do
  echo "== BTR_BQ_FORM=$f"
  BTR_BQ_FORM=$f python tools/bq_ab.py 2>&1 | grep "B="
done
cd /tmp && export TMPDIR=/tmp
for f in 1 2; do
  rm -rf /tmp/bqf$f
  BTR_BQ_FORM=$f rocprofv3 --kernel-trace --stats -d /tmp/bqf$f -o r -- python3 $GRAFT_REPO_ROOT/tools/bq_ab.py > /dev/null 2>&1
  DB=$(find /tmp/bqf$f -name "*.db" | head -1)
  echo "== kernel trace, BTR_BQ_FORM=$f"
  python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB | grep -i "bqb_query" | cut -c1-170
done
