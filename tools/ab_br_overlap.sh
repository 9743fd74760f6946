#!/bin/bash
# Same-box A/B: the source branch's head beside the target branch's backbone (train.two_forwards).
cd ${GRAFT_REPO_ROOT:-.}
run() {
  python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline --no-sequential 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 $2  %.3f ms  host %.2f ms' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))"
}
for w in ${WORKLOADS:-br cr}; do
  for i in 1 2; do
    BTR_BR_OVERLAP=0 run $w "one stream     "
    BTR_BR_OVERLAP=1 run $w "head || backbone"
  done
done
