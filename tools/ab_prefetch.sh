#!/bin/bash
# Same-box A/B of where the next batch's sampling pyramid starts: before this step's forward
# (round 3), behind its SA level 2 / 1 / 3 (round 4), or before its backward.
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do
  for v in forward sa2 sa1 sa3 sa4 backward; do
    export BTR_PREFETCH_AT=$v; unset BTR_FORK_LEVEL
    case $v in sa1) export BTR_PREFETCH_AT=sa2 BTR_FORK_LEVEL=1;; sa3) export BTR_PREFETCH_AT=sa2 BTR_FORK_LEVEL=3;; sa4) export BTR_PREFETCH_AT=sa2 BTR_FORK_LEVEL=4;; esac
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequential 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('prefetch at $v  %.3f ms  host %.2f | fps in-loop %.3f ms | gemm family %.3f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], r['avg_ms'], d['mlp_roofline']['ms_per_step']))"
  done
done
