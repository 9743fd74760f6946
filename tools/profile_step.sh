#!/bin/bash
# One-step kernel profile on the GPU box:  tools/profile_step.sh <tag>  ->  gpurun_out/<tag>/
# (rocprofv3 --kernel-trace over a short bench run, summarised on the box: the rocpd database
# is too large to travel).
set -e
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/$TAG
rocprofv3 --kernel-trace -d /tmp/$TAG -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --sequential > /tmp/$TAG.json 2>/tmp/$TAG.err
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
DB=$(find /tmp/$TAG -name "*.db" | head -1)
python tools/rocpd_step.py $DB fps_bucket_kernel gpurun_out/$TAG/one_step.md
python tools/rocpd_stats.py $DB > gpurun_out/$TAG/kernel_stats.md 2>/dev/null || true
python tools/rocpd_timeline.py $DB fps_bucket_kernel gpurun_out/$TAG/timeline.txt || true
grep "^{" /tmp/$TAG.json | tail -1 > gpurun_out/$TAG/bench.json
head -3 gpurun_out/$TAG/one_step.md
