#!/bin/bash
# SQ counters of the kernels whose name contains $2, run by `python3 $1`: where do the
# wave-cycles go?  Two --pmc passes (8 SQ slots each).  Output: markdown on stdout.
#   bash tools/pmc_kernels.sh tools/bwd_fused_ab.py sa_bwd_fused > gpurun_out/x.md
TOOL=$1; PAT=$2; TOOLARGS=${3:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kpmc1 /tmp/kpmc2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES -d /tmp/kpmc1 -o r -- python3 $ROOT/$TOOL $TOOLARGS > /tmp/kpmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d /tmp/kpmc2 -o r -- python3 $ROOT/$TOOL $TOOLARGS > /tmp/kpmc2.log 2>&1
cd $ROOT
python3 - "$PAT" $(find /tmp/kpmc1 -name "*.db" | head -1) $(find /tmp/kpmc2 -name "*.db" | head -1) <<'PY'
import sqlite3, sys
pat = sys.argv[1]
agg = {}
for db in sys.argv[2:]:
    c = sqlite3.connect(db).cursor()
    for name, grid, cn, v, dur, n in c.execute(
            "select kernel_name, grid_size, counter_name, avg(value), avg(duration), count(*) "
            "from counters_collection where kernel_name like ? group by kernel_name, grid_size, "
            "counter_name", ('%' + pat + '%',)):
        import re
        m = re.search(r'(\w+_kernel(<[^>]*>)?)', name)
        short = m.group(1) if m else name.split('(')[0].replace('void ', '').replace('btr::', '')
        agg.setdefault((short, grid), {'n': n}).setdefault('dur', dur)
        agg[(short, grid)][cn] = v
print("| kernel | grid | us (profiled) | waves | wait_any | wait_inst | active | lds_stall | "
      "mfma_busy / busy | VALU insts/wave | LDS insts/wave | LDS conflict / active | VMEM rd+wr /wave |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for (k, g), d in sorted(agg.items(), key=lambda kv: -kv[1]['dur']):
    wc = max(d.get('SQ_WAVE_CYCLES', 1), 1)
    w = max(d.get('SQ_WAVES', 1), 1)
    print("| `%s` | %d | %.1f | %.0f | %.0f%% | %.0f%% | %.0f%% | %.0f%% | %.2f | %.0f | %.0f | %.2f | %.0f |" % (
        k, g, d['dur'] / 1e3, w, 100 * d.get('SQ_WAIT_ANY', 0) / wc,
        100 * d.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * d.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        100 * d.get('SQ_WAIT_INST_LDS', 0) / wc,
        d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1, d.get('SQ_BUSY_CYCLES', 1)),
        d.get('SQ_INSTS_VALU', 0) / w, d.get('SQ_INSTS_LDS', 0) / w,
        d.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, d.get('SQ_LDS_IDX_ACTIVE', 1)),
        (d.get('SQ_INSTS_VMEM_RD', 0) + d.get('SQ_INSTS_VMEM_WR', 0)) / w))
PY
tail -2 /tmp/kpmc1.log /tmp/kpmc2.log 1>&2
