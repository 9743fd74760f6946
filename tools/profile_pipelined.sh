#!/bin/bash
# Timeline of one step of the software-pipelined loop (the bench headline):
#   tools/profile_pipelined.sh <tag>  ->  gpurun_out/<tag>/{one_step.md,timeline.txt,bench.json}
set -e
TAG=${1:-pipe}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/$TAG
rocprofv3 --kernel-trace --stats -d /tmp/$TAG -o r -- python3 $GRAFT_REPO_ROOT/bench.py ${PIPE_ARGS:---steps 16 --warmup 3} --no-cpu-baseline --no-sequential > /tmp/$TAG.json 2>/tmp/$TAG.err
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
DB=$(find /tmp/$TAG -name "*.db" | head -1)
export ROCPD_WINDOW=median
python tools/rocpd_step.py $DB fps_bucket_kernel gpurun_out/$TAG/one_step.md
python tools/rocpd_timeline.py $DB fps_bucket_kernel gpurun_out/$TAG/timeline.txt
grep "^{" /tmp/$TAG.json | tail -1 > gpurun_out/$TAG/bench.json
head -1 gpurun_out/$TAG/one_step.md; tail -1 gpurun_out/$TAG/timeline.txt
python tools/rocpd_stats.py $DB gpurun_out/$TAG/kernel_stats.md
python tools/recompute_roofline.py gpurun_out/$TAG/bench.json gpurun_out/$TAG/one_step.md > gpurun_out/$TAG/roofline_check.md
cat gpurun_out/$TAG/roofline_check.md
