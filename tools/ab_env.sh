#!/bin/bash
# Same-box A/B of one environment switch on the FSB bench:  tools/ab_env.sh VAR [rounds]
# alternates VAR=0 / default, prints pipelined and sequential ms per step of every run.
VAR=$1; N=${2:-3}
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq 1 $N); do
  for v in 0 1; do
    if [ $v = 0 ]; then export $VAR=0; else unset $VAR; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=%s  %.3f ms  seq %.3f ms  host %.2f ms' % ('0' if $v == 0 else 'default', d['ms_per_step'], d.get('sequential_ms_per_step') or 0, d['host_enqueue_ms_per_step']))"
  done
done
