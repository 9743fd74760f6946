#!/bin/bash
# Same-box A/B: which weight-gradient GEMMs run on the library's second stream.
cd ${GRAFT_REPO_ROOT:-.}
run() {
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1  %.3f ms  seq %.3f ms  host %.2f ms' % (d['ms_per_step'], d.get('sequential_ms_per_step') or 0, d['host_enqueue_ms_per_step']))"
}
for i in 1 2; do
  BTR_WGRAD_STREAM=1 run "both on the side stream (old default)"
  BTR_WGRAD_STREAM=0 run "all on the callers stream         "
done
