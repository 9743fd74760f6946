#!/bin/bash
# Copy the summaries tools/profile_round.sh / profile_pipelined.sh / profile_gf.sh /
# probe/gemm_pmc.sh left under gpurun_out/ into profiles/ (tracked):
#   tools/collect_profiles.sh <tag>     e.g. r03_h  (expects gpurun_out/<tag>, <tag>_pipe, ...)
set -e
T=$1
S=gpurun_out
cd "$(dirname "$0")/.."
for f in kernel_stats.md one_step.md pmc_FETCH_SIZE.md pmc_WRITE_SIZE.md pmc_traffic.json \
         bench.json bench_under_profiler.json roofline_check.md bench_br.json bench_cr.json \
         bench_gf.json bench_gf_eager.json bench_gf_graph.json bench_gfbr.json bench_steps20.json bench_c5.json gemm_tn_ab.txt gemm_nt_small_ab.txt bwd_fused_ab.txt fps_prefix_ab.txt bwd_gram_ab.txt gemm_sm_ab.txt two_branch_ab.txt fps_lds_ab.txt fps_prof.txt gputests.txt step_table.md pipelined_one_step.md pipelined_timeline.txt pipelined_kernel_stats.md bench_steps20_b.json fused_sq_counters.md gram_sq_counters.md fps_interference.txt phase_times.txt streams_ab.txt eval_times.txt gf_lanes.txt; do
  [ -f $S/$T/$f ] && cp $S/$T/$f profiles/${T}_$f
done
[ -f $S/$T/pmc_traffic.json ] && cp $S/$T/pmc_traffic.json profiles/pmc_traffic.json
[ -f $S/$T/gemm_sq_counters.txt ] && { echo '```'; grep -v "^W20\|simple_timer" $S/$T/gemm_sq_counters.txt; echo '```'; } > profiles/${T}_gemm_sq_counters_raw.md
for f in kernel_stats.md one_step.md timeline.txt bench.json roofline_check.md; do
  [ -f $S/${T}_pipe/$f ] && cp $S/${T}_pipe/$f profiles/${T}_pipelined_$f
done
for m in eager graph; do
  for f in one_step.md bench.json; do
    [ -f $S/${T}_gf_$m/$f ] && cp $S/${T}_gf_$m/$f profiles/${T}_gf_${m}_$f
  done
done
ls profiles | grep "^${T}_" | wc -l
