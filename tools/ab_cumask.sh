#!/bin/bash
# Same-box A/B of the CU partition (BTR_CU_MASK = CUs per XCD reserved for the large-scene FPS).
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do
  for c in 0 1 2; do
    if [ $c = 0 ]; then unset BTR_CU_MASK; else export BTR_CU_MASK=$c; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; m=d['mlp_roofline']
print('BTR_CU_MASK=$c  %.3f ms  seq %.3f ms  host %.2f | fps in-loop %.3f ms alone %.3f | gemm family %.3f ms (event-timed) | bq %.1f us' % (d['ms_per_step'], d.get('sequential_ms_per_step') or 0, d['host_enqueue_ms_per_step'], r['avg_ms'], r.get('avg_ms_running_alone') or 0, m['ms_per_step'], 1e3*d['ball_query_roofline']['avg_ms']))"
  done
done
