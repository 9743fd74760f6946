#!/bin/bash
# The whole judged set of one build, on the GPU box:  tools/final_profiles.sh <tag>
# (then, here: tools/collect_profiles.sh <tag>).  ~10 GPU-minutes.
TAG=${1:-final}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
bash tools/profile_pipelined.sh ${TAG}_pipe > $O/profile_pipelined.log 2>&1
bash tools/probe/gemm_pmc.sh > $O/gemm_sq_counters.txt 2>&1
bash tools/profile_gf.sh ${TAG}_gf_eager GF_ARGS=--no-graph > $O/profile_gf_eager.log 2>&1
bash tools/profile_gf.sh ${TAG}_gf_graph GF_ARGS=--graph > $O/profile_gf_graph.log 2>&1
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_steps20.json
for wl in c5 br cr gf gfbr; do
  python bench.py --workload $wl 2>/dev/null | tail -1 > $O/bench_$wl.json
done
python bench.py --workload gf --graph 2>/dev/null | tail -1 > $O/bench_gf_graph.json
python tools/gemm_tn_ab.py > $O/gemm_tn_ab.txt 2>&1
python tools/gemm_nt_small_ab.py > $O/gemm_nt_small_ab.txt 2>&1
python tools/bwd_fused_ab.py > $O/bwd_fused_ab.txt 2>&1
python tools/fps_prefix_ab.py > $O/fps_prefix_ab.txt 2>&1
python tools/bwd_gram_ab.py > $O/bwd_gram_ab.txt 2>&1
python tools/gemm_sm_ab.py > $O/gemm_sm_ab.txt 2>&1
bash tools/pmc_kernels.sh tools/bwd_fused_ab.py sa_bwd_fused > $O/fused_sq_counters.md 2>/dev/null
bash tools/pmc_kernels.sh tools/bwd_gram_ab.py sa_bwd_gram > $O/gram_sq_counters.md 2>/dev/null
# what the step's streams cost each other
{ python tools/fps_interference.py 2>&1 | tail -2
  for m in "8 0 2000" "8 1 2000" "8 2 2000"; do
    echo "occupant (workgroups, mode 0 sleep / 1 VALU spin / 2 L2 pointer chase, us): $m"
    python tools/fps_interference.py --occupant $m 2>&1 | tail -1
  done
  echo "BTR_FPS_LDS_KB=0:"; BTR_FPS_LDS_KB=0 python tools/fps_interference.py 2>&1 | tail -2; } > $O/fps_interference.txt
{ python tools/phase_times.py 2>&1 | tail -11; echo; python tools/phase_times.py --sequential 2>&1 | tail -11; } > $O/phase_times.txt
{ bash tools/ab_env2.sh "BTR_FPS_LDS_KB" 2; bash tools/ab_side.sh; BENCH_ARGS= bash tools/ab_gridcus.sh "256 248" | tail -6; } > $O/streams_ab.txt 2>&1
{ bash tools/ab_sink.sh; bash tools/ab_br_overlap.sh; } > $O/two_branch_ab.txt 2>&1
python tools/fps_lds_ab.py > $O/fps_lds_ab.txt 2>&1
python -c "
import json
for f in ('bench','bench_steps20','bench_c5','bench_br','bench_cr','bench_gf','bench_gf_graph','bench_gfbr'):
    try:
        d = json.load(open('$O/%s.json' % f))
        print(f, round(d['value'], 1), d['unit'], round(d['ms_per_step'], 3), 'ms host', round(d['host_enqueue_ms_per_step'], 2), d.get('chain_paths'))
    except Exception as e:
        print(f, 'ERR', e)
"
head -1 $O/one_step.md gpurun_out/${TAG}_pipe/one_step.md gpurun_out/${TAG}_gf_eager/one_step.md gpurun_out/${TAG}_gf_graph/one_step.md
cat $O/roofline_check.md | tail -5
