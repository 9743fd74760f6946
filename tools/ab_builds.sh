#!/bin/bash
# Same-box A/B of two builds: the working tree against a second checkout (default _ab_prev, e.g.
# `git worktree add _ab_prev HEAD` + build there).  usage: tools/ab_builds.sh [dir]  (WL, BENCH_ARGS as ab_gf.sh)
OTHER=${1:-_ab_prev}
WL=${WL:-fsb}
for i in 1 2 3; do
for d in . $OTHER; do
  echo "== $WL $d"
  (cd $d && python bench.py --workload $WL --no-cpu-baseline --no-sequential ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'host', round(d['host_enqueue_ms_per_step'],2))")
done; done
