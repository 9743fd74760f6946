for i in 1 2; do
for cfg in "X=1" "BTR_CHAIN_MIN_ROWS=0"; do
  echo "== gf graph $cfg"; env $cfg python bench.py --workload gf --no-cpu-baseline --no-sequential 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'host', round(d['host_enqueue_ms_per_step'],2), d.get('chain_paths'), d['hip_graph'])"
done; done
for cfg in "X=1" "BTR_CHAIN_MIN_ROWS=0"; do
  echo "== gfbr graph $cfg"; env $cfg python bench.py --workload gfbr --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'host', round(d['host_enqueue_ms_per_step'],2), d.get('chain_paths'), d['hip_graph'])"
done
