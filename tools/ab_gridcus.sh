#!/bin/bash
# Same-box A/B of BTR_GRID_CUS (CUs the one-round grids are sized for; no CU partition).
cd ${GRAFT_REPO_ROOT:-.}
for r in ${1:-256 248 240}; do
  echo "== BTR_GRID_CUS=$r"; BTR_GRID_CUS=$r python tools/fps_interference.py 2>&1 | tail -2
done
for i in 1 2 3; do
  for r in ${1:-256 248 240}; do
    BTR_GRID_CUS=$r python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('BTR_GRID_CUS=$r  %.3f ms  seq %.3f ms  host %.2f' % (d['ms_per_step'], d.get('sequential_ms_per_step') or 0, d['host_enqueue_ms_per_step']))"
  done
done
