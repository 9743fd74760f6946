#!/bin/bash
# Same-box A/B: BTR_FPS_PRIO=0 (the large-scene FPS kernel without raised wave priority) vs default.
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do
  for v in default 0; do
    if [ $v = default ]; then unset BTR_FPS_PRIO; else export BTR_FPS_PRIO=0; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('BTR_FPS_PRIO=$v  %.3f ms  seq %.3f ms  host %.2f | fps in-loop %.3f ms alone %.3f' % (d['ms_per_step'], d.get('sequential_ms_per_step') or 0, d['host_enqueue_ms_per_step'], r['avg_ms'], r.get('avg_ms_running_alone') or 0))"
  done
done
