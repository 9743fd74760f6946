#!/bin/bash
# Kernel statistics of the inference forward (tools/eval_throughput.py):  tools/profile_eval.sh <tag>
set -e
TAG=${1:-eval}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/$TAG
EVAL_ONLY=forward rocprofv3 --kernel-trace --stats -d /tmp/$TAG -o r -- python3 $GRAFT_REPO_ROOT/tools/eval_throughput.py > /tmp/$TAG.out 2>/tmp/$TAG.err
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
cp /tmp/$TAG.out gpurun_out/$TAG/times.txt
DB=$(find /tmp/$TAG -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/$TAG/kernel_stats.md
ROCPD_WINDOW=median python tools/rocpd_timeline.py $DB fps_bucket_kernel gpurun_out/$TAG/timeline.txt
head -40 gpurun_out/$TAG/kernel_stats.md
