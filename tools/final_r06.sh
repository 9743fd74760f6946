#!/bin/bash
# The judged set of the final round-6 build, on the GPU box:  tools/final_r06.sh <tag>
# (then here: tools/collect_profiles.sh <tag>).  ~8 GPU-minutes.
TAG=${1:-r06_z}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -m gpu -q --durations=10 > $O/gputests.txt 2>&1
grep -E "passed|failed" $O/gputests.txt
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
bash tools/profile_pipelined.sh ${TAG}_pipe > $O/profile_pipelined.log 2>&1
python tools/step_table.py gpurun_out/${TAG}_pipe/timeline.txt > $O/step_table.md 2>/dev/null
cp gpurun_out/${TAG}_pipe/one_step.md $O/pipelined_one_step.md
cp gpurun_out/${TAG}_pipe/timeline.txt $O/pipelined_timeline.txt
cp gpurun_out/${TAG}_pipe/kernel_stats.md $O/pipelined_kernel_stats.md
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_steps20.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_steps20_b.json
for wl in c5 br cr gf gfbr; do
  python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$wl.json
done
python tools/eval_throughput.py 2>&1 | grep -v amdgpu.ids > $O/eval_times.txt
bash tools/gf_lanes_ab.sh > $O/gf_lanes.txt 2>&1
python tools/fps_lds_ab.py > $O/fps_lds_ab.txt 2>&1
python tools/fps_prof.py > $O/fps_prof.txt 2>&1
{ python tools/phase_times.py 2>&1 | tail -11; } > $O/phase_times.txt
python -c "
import json
for f in ('bench','bench_steps20','bench_steps20_b','bench_c5','bench_br','bench_cr','bench_gf','bench_gfbr'):
    try:
        d = json.load(open('$O/%s.json' % f))
        print(f, round(d['value'], 1), d['unit'], round(d['ms_per_step'], 3), 'ms host', round(d['host_enqueue_ms_per_step'], 2), 'seq', d.get('sequential_ms_per_step'))
    except Exception as e:
        print(f, 'ERR', e)
"
head -1 $O/one_step.md gpurun_out/${TAG}_pipe/one_step.md
cat $O/roofline_check.md | tail -5
