#!/bin/bash
# Same-box A/B of BTR_FPS_LDS_KB (unused dynamic LDS the large-scene FPS launch asks for, to keep
# other streams' workgroups off its CUs).
cd ${GRAFT_REPO_ROOT:-.}
for r in ${1:-0 64 120 150}; do
  echo "== BTR_FPS_LDS_KB=$r"; BTR_FPS_LDS_KB=$r python tools/fps_interference.py 2>&1 | tail -2
done
for i in 1 2 3; do
  for r in ${1:-0 64 120 150}; do
    BTR_FPS_LDS_KB=$r python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('BTR_FPS_LDS_KB=$r  %.3f ms  seq %.3f ms  host %.2f | fps in-loop %.3f ms alone %.3f' % (d['ms_per_step'], d.get('sequential_ms_per_step') or 0, d['host_enqueue_ms_per_step'], r['avg_ms'], r.get('avg_ms_running_alone') or 0))"
  done
done
