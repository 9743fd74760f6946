#!/bin/bash
# Same-box A/B of BTR_GRID_ROUNDS (rounds of resident workgroups per row-chunked launch).
cd ${GRAFT_REPO_ROOT:-.}
for r in 1 2 4; do
  echo "== BTR_GRID_ROUNDS=$r"; BTR_GRID_ROUNDS=$r python tools/fps_interference.py 2>&1 | tail -2
done
for i in 1 2 3; do
  for r in 1 2 4; do
    BTR_GRID_ROUNDS=$r python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('BTR_GRID_ROUNDS=$r  %.3f ms  seq %.3f ms  host %.2f' % (d['ms_per_step'], d.get('sequential_ms_per_step') or 0, d['host_enqueue_ms_per_step']))"
  done
done
