#!/bin/bash
# The N>1 code path on one GPU: a one-rank RCCL process group + DistributedDataParallel around
# the model (BTR_FORCE_DDP=1), against the bare module.  Usage: tools/ddp_one_rank.sh
cd $GRAFT_REPO_ROOT
echo "bare module:"; python bench.py --no-cpu-baseline --sequential 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"
echo "FlatGradParallel (1 rank, RCCL):"; BTR_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 python bench.py --no-cpu-baseline --sequential 2>/tmp/ddp.err | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" || tail -20 /tmp/ddp.err
echo "FlatGradParallel BR:"; BTR_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 python bench.py --workload br --no-cpu-baseline 2>/tmp/ddp.err | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" || tail -20 /tmp/ddp.err
echo "DistributedDataParallel (1 rank, RCCL):"; BTR_DP=ddp BTR_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 python bench.py --no-cpu-baseline --sequential 2>/tmp/ddp.err | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" || tail -20 /tmp/ddp.err
echo "DistributedDataParallel BR:"; BTR_DP=ddp BTR_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29514 python bench.py --workload br --no-cpu-baseline 2>/tmp/ddp.err | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" || tail -20 /tmp/ddp.err
