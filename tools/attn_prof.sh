#!/bin/bash
# kernel durations of tools/attn_bench.py (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/attn_prof
rocprofv3 --kernel-trace --stats -d /tmp/attn_prof -o r -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py > /tmp/attn_prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/attn_prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB /tmp/attn_stats.md
grep -E "attn_" /tmp/attn_stats.md | sed 's/`\([^`]\{0,60\}\)[^`]*`/\1/'
