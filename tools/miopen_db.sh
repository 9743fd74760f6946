#!/bin/bash
# (Re)generates the shipped MIOpen find-db: runs every bench workload once with
# torch.backends.cudnn.benchmark on and MIOPEN_USER_DB_PATH pointing at the package's
# miopen_db/, so that MIOpen's solver search for the stock 1x1 convolutions (FP / voting /
# proposal / GroupFree3D head layers) is stored; copy gpurun_out/miopen_db_new/*.txt over
# backtoreality_amd/miopen_db/.  Usage (GPU box): bash tools/miopen_db.sh
cd $GRAFT_REPO_ROOT
export MIOPEN_USER_DB_PATH=$PWD/backtoreality_amd/miopen_db
for w in fsb br cr gf gfbr; do
  s=$(date +%s)
  python bench.py --workload $w --no-cpu-baseline --sequential 2>gpurun_out/at_$w.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$w', d['value'], d['ms_per_step'])"
  echo "$w took $(( $(date +%s) - s )) s"
done
mkdir -p gpurun_out/miopen_db_new
cp backtoreality_amd/miopen_db/*.txt gpurun_out/miopen_db_new/
ls -la gpurun_out/miopen_db_new
