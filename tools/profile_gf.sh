#!/bin/bash
# One-step kernel table of the GroupFree3D workload:  tools/profile_gf.sh <tag> [env settings]
set -e
TAG=${1:-gf}; shift || true
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/$TAG
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace -d /tmp/$TAG -o r -- python3 $GRAFT_REPO_ROOT/bench.py --workload gf --steps ${GF_STEPS:-6} --warmup 3 --no-cpu-baseline ${GF_ARGS:-} > /tmp/$TAG.json 2>/tmp/$TAG.err
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
DB=$(find /tmp/$TAG -name "*.db" | head -1)
ROCPD_WINDOW=median python tools/rocpd_step.py $DB fps_bucket_kernel gpurun_out/$TAG/one_step.md
grep "^{" /tmp/$TAG.json | tail -1 > gpurun_out/$TAG/bench.json
head -1 gpurun_out/$TAG/one_step.md
ROCPD_WINDOW=median python tools/rocpd_timeline.py $DB fps_bucket_kernel gpurun_out/$TAG/timeline.txt
