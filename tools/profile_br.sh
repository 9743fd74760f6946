#!/bin/bash
# Kernel statistics of the Back-to-Reality step on the GPU box -> gpurun_out/<tag>/kernel_stats.md
# (per-kernel calls and time over 6 timed + 3 warm-up steps of the pipelined loop, both queues)
set -e
TAG=${1:-prof_br}
WL=${2:-br}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/$TAG
rocprofv3 --kernel-trace --stats -d /tmp/$TAG -o r -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 6 --warmup 3 --no-cpu-baseline --no-sequential > /tmp/$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
DB=$(find /tmp/$TAG -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/$TAG/kernel_stats.md
head -40 gpurun_out/$TAG/kernel_stats.md
