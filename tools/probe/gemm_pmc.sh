#!/bin/bash
# SQ counters of the SA GEMM kernels (tools/gemm_ab.py shapes): where do the wave-cycles go?
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gpmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES -d /tmp/gpmc -o r -- python3 $GRAFT_REPO_ROOT/tools/gemm_ab.py > /tmp/gpmc.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/gpmc -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]).cursor()
rows = c.execute("select kernel_name, grid_size, counter_name, avg(value), avg(duration), count(*) from counters_collection where kernel_name like '%gemm_nt%' group by kernel_name, grid_size, counter_name").fetchall()
agg = {}
for name, grid, cn, v, dur, n in rows:
    key = (name.split('(')[0][-40:], grid)
    agg.setdefault(key, {'dur': dur, 'n': n})[cn] = v
for k, d in sorted(agg.items(), key=lambda kv: -kv[1]['dur']):
    wc = d.get('SQ_WAVE_CYCLES', 1)
    print("%s grid %d: %.1f us x%d | wave_cyc %.3g busy_cyc %.3g | wait_any %.0f%% wait_inst %.0f%% active %.0f%% lds_stall %.0f%% | mfma_busy/busy %.2f waves %.0f" % (
        k[0], k[1], d['dur'] / 1e3, d['n'], wc, d.get('SQ_BUSY_CYCLES', 0),
        100 * d.get('SQ_WAIT_ANY', 0) / wc, 100 * d.get('SQ_WAIT_INST_ANY', 0) / wc,
        100 * d.get('SQ_ACTIVE_INST_ANY', 0) / wc, 100 * d.get('SQ_WAIT_INST_LDS', 0) / wc,
        d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1, d.get('SQ_BUSY_CYCLES', 1)), d.get('SQ_WAVES', 0)))
PY
tail -3 /tmp/gpmc.log
