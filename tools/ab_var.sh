#!/bin/bash
# Same-box A/B of one environment variable over several values:  tools/ab_var.sh VAR "v1 v2 v3" [rounds]
cd ${GRAFT_REPO_ROOT:-.}
VAR=$1
for i in $(seq 1 ${3:-2}); do
  for v in $2; do
    env $VAR=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$v  %.3f ms  seq %.3f ms  host %.2f ms' % (d['ms_per_step'], d.get('sequential_ms_per_step') or 0, d['host_enqueue_ms_per_step']))"
  done
done
