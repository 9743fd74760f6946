#!/bin/bash
# A/B of environment switches on the eager GroupFree3D step (and the FSB step): each setting twice,
# alternating (the boxes' host speed drifts between runs).  usage: tools/ab_gf.sh "A=1" "B=0 C=2" ...
WL=${WL:-gf}
for i in 1 2; do
for cfg in "X=1" "$@"; do
  echo "== $WL $cfg"
  env $cfg python bench.py --workload $WL --no-graph --no-cpu-baseline --no-sequential ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'host', round(d['host_enqueue_ms_per_step'],2), d.get('chain_paths'))"
done; done
