#!/bin/bash
# The bench lines of tools/final_profiles.sh alone (boxes differ in host speed: re-run when the
# profile job landed on a slow one):  tools/final_benches.sh <tag>
TAG=${1:-final}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
python bench.py 2>/dev/null | tail -1 > $O/bench.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_steps20.json
for wl in br cr gf gfbr; do
  python bench.py --workload $wl 2>/dev/null | tail -1 > $O/bench_$wl.json
done
python bench.py --workload gf --graph 2>/dev/null | tail -1 > $O/bench_gf_graph.json
python -c "
import json
for f in ('bench','bench_steps20','bench_br','bench_cr','bench_gf','bench_gf_graph','bench_gfbr'):
    d = json.load(open('$O/%s.json' % f))
    print(f, round(d['value'], 1), d['unit'], round(d['ms_per_step'], 3), 'ms host', round(d['host_enqueue_ms_per_step'], 2), d.get('chain_paths'), 'mlp', d.get('mlp_roofline', {}).get('frac'))
"
