#!/bin/bash
# Refresh the judged profiles on the GPU box:  tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of `bench.py` -> kernel_stats.md + one_step.md
#   2. two SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, as the guide
#      prescribes) -> pmc_traffic.json (+ the per-kernel tables)
# Everything is summarised on the box into gpurun_out/<tag>/ (the databases are too large to
# travel); copy what should be judged into profiles/.
set -e
# every kernel alone on the chip: sequential loop, and the library's second (weight-gradient)
# stream off -- overlapped kernels stretch each other's durations
export BTR_WGRAD_STREAM=0
TAG=${1:-round}
ARGS="--steps 3 --warmup 2 --no-cpu-baseline --sequential"
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/$TAG.*
rocprofv3 --kernel-trace --stats -d /tmp/$TAG.kt -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --sequential > $OUT/bench_under_profiler.json 2>/tmp/$TAG.kt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/$TAG.fetch -o r -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2>/tmp/$TAG.fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/$TAG.write -o r -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2>/tmp/$TAG.write.err
cd $GRAFT_REPO_ROOT
KT=$(find /tmp/$TAG.kt -name "*.db" | head -1)
FE=$(find /tmp/$TAG.fetch -name "*.db" | head -1)
WR=$(find /tmp/$TAG.write -name "*.db" | head -1)
python tools/rocpd_step.py $KT fps_bucket_kernel $OUT/one_step.md
python tools/rocpd_stats.py $KT $OUT/kernel_stats.md
python tools/rocpd_pmc.py $FE $OUT/pmc_FETCH_SIZE.md
python tools/rocpd_pmc.py $WR $OUT/pmc_WRITE_SIZE.md
python tools/pmc_traffic.py $FE $WR $OUT/pmc_traffic.json
python tools/recompute_roofline.py $OUT/bench_under_profiler.json $OUT/one_step.md > $OUT/roofline_check.md
# the headline loop (software-pipelined) and the sequential loop beside it, no profiler attached
unset BTR_WGRAD_STREAM
python bench.py > $OUT/bench.json 2>/dev/null
head -3 $OUT/one_step.md; cat $OUT/roofline_check.md; tail -c 600 $OUT/bench.json
