#!/bin/bash
# One-step kernel profile of the Back-to-Reality step on the GPU box -> gpurun_out/<tag>/
set -e
TAG=${1:-prof_br}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/$TAG
rocprofv3 --kernel-trace -d /tmp/$TAG -o r -- python3 $GRAFT_REPO_ROOT/bench.py --workload br --steps 6 --warmup 3 --no-cpu-baseline > /tmp/$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
DB=$(find /tmp/$TAG -name "*.db" | head -1)
python tools/rocpd_step.py $DB fps_sortm_scan_kernel gpurun_out/$TAG/one_step.md 2
head -3 gpurun_out/$TAG/one_step.md
