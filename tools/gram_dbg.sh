#!/bin/bash
# producer-only / consumer-only times of sa_bwd_gram_ws_kernel (BTR_GRAM_DBG): stdout
cd /tmp && export TMPDIR=/tmp
for D in 0 1 2; do
  rm -rf /tmp/pg$D
  BTR_GRAM_DBG=$D rocprofv3 --kernel-trace --stats -d /tmp/pg$D -o r -- python3 $GRAFT_REPO_ROOT/tools/bwd_gram_ab.py > /tmp/pg$D.log 2>&1
  DB=$(find /tmp/pg$D -name "*.db" | head -1)
  echo "== BTR_GRAM_DBG=$D"
  python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB | grep "gram_ws\|sa_bwd_fused_kernel<4, 1" | cut -c1-160
done
