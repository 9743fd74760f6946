#!/bin/bash
# Same-box A/B of the BatchNorm ticket protocol:  full fences / write-through partials / write-through
# partials for every statistics GEMM (no row limit) / no ticket at all.
cd ${GRAFT_REPO_ROOT:-.}
WL=${WL:-fsb}
run() { python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-sequential ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1  %.3f ms  host %.2f ms' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))"; }
for i in 1 2 3; do
  BTR_BN_TICKET_FENCE=1 run "fence      "
  run "default    "
  BTR_BN_TICKET_MAX_ROWS=100000000 run "all rows   "
  BTR_BN_TICKET=0 run "no ticket  "
done
