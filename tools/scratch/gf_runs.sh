#!/bin/bash
# tools/scratch/gf_runs.sh "<env settings>" ...   -- one GroupFree3D bench line per argument
cd "$(dirname "$0")/../.."
for e in "$@"; do
  env $e python bench.py --workload gf --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', round(d['ms_per_step'],3), round(d['host_enqueue_ms_per_step'],3))"
done
