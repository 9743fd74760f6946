#!/bin/bash
# Same-box A/B of the flat gradient sinks (pointnet2/grad_sink.py) on the two-forward workloads.
cd ${GRAFT_REPO_ROOT:-.}
run() {
  python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline --no-sequential 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 $2  %.3f ms  host %.2f ms' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))"
}
for w in ${WORKLOADS:-gfbr cr br}; do
  for i in 1 2; do
    BTR_GRAD_SINK=0 run $w "sinks off"
    BTR_GRAD_SINK=1 run $w "sinks on "
  done
done
