#!/bin/bash
# per-kernel times of tools/bwd_gram_ab.py:  tools/prof_gram.sh -> stdout
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg
rocprofv3 --kernel-trace --stats -d /tmp/pg -o r -- python3 $GRAFT_REPO_ROOT/tools/bwd_gram_ab.py > /tmp/pg.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/pg -name "*.db" | head -1)
python tools/rocpd_stats.py $DB | cut -c1-200 | head -30
