#!/bin/bash
# Kernel timeline of one GroupFree3D step of tools/gf_stack_times.py (no GEMM-trace pass, so the
# decoder stack's replayed graphs and its two lanes are what is seen):
#   tools/profile_gf_stack.sh <tag> [env settings]
set -e
TAG=${1:-gfs}; shift || true
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/$TAG
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace -d /tmp/$TAG -o r -- python3 $GRAFT_REPO_ROOT/tools/gf_stack_times.py 8 > /tmp/$TAG.out 2>/tmp/$TAG.err
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
cp /tmp/$TAG.out gpurun_out/$TAG/times.txt
DB=$(find /tmp/$TAG -name "*.db" | head -1)
ROCPD_WINDOW=median python tools/rocpd_timeline.py $DB fps_bucket_kernel gpurun_out/$TAG/timeline.txt
python tools/step_table.py gpurun_out/$TAG/timeline.txt > gpurun_out/$TAG/step_table.md
head -3 gpurun_out/$TAG/step_table.md
