#!/bin/bash
# Same-box A/B of several environment switches, one after the other:  tools/ab_env2.sh "VAR1 VAR2" [rounds]
cd ${GRAFT_REPO_ROOT:-.}
for V in $1; do bash tools/ab_env.sh $V ${2:-3}; done
