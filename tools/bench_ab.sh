#!/bin/bash
# bench.py under a list of environment settings, back to back on ONE box (boxes differ by a
# few percent):  tools/bench_ab.sh "A=1 B=0" "A=0" ...   prints value / ms per step per setting
for setting in "$@"; do
  out=$(env $setting python bench.py --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$setting :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f scenes/s  %.2f ms/step  sequential %.2f ms' % (d['value'], d['ms_per_step'], d.get('sequential',{}).get('ms_per_step',0)))")"
done
