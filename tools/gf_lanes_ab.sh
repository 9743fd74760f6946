#!/bin/bash
# The GroupFree3D decoder stack's replayed graphs and lanes, same box, same build:
#   tools/gf_lanes_ab.sh > gpurun_out/<tag>_gf_lanes.txt
# step time and the main lane's GPU time of the two library calls for: per-call allocations and
# single launches (round 5's path), slots with single launches, replayed graphs on 1 / 2 / 3
# streams; then the per-segment GPU times of two steps on three lanes (BTR_LANE_DEBUG).
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python tools/gf_stack_times.py 2>&1 | grep -v amdgpu.ids | tail -4; }
run BTR_GRAPHS=0 BTR_GF_SLOTS=0
run BTR_GRAPHS=0
run BTR_GF_LANES=1
run BTR_GF_LANES=2
run BTR_GF_LANES=3
echo "== BTR_LANE_DEBUG=1 (three lanes; the report waits for the lanes after every call)"
BTR_LANE_DEBUG=1 python tools/gf_stack_times.py 2 2>&1 | grep -v amdgpu.ids | tail -48 | head -44
