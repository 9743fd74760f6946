#!/usr/bin/env python3
"""Per-kernel table of a bench.py JSON line (its `kernels` object: the instrumented steps):
   python tools/bench_kernels.py bench.json [other.json]  -- second file: side-by-side A/B."""
import json
import re
import sys


def load(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    out = {}
    for k, v in d['kernels'].items():
        m = re.match(r'(\w+)\[(.*)\]', k)
        key = tuple(int(x) for x in m.group(2).split(',')) if m.group(2) else ()
        out[(m.group(1), key)] = (v['avg_ms'] * 1e3, v['launches_per_step'])
    return d, out


d, a = load(sys.argv[1])
b = load(sys.argv[2])[1] if len(sys.argv) > 2 else None
rows = sorted(a.items(), key=lambda kv: -kv[1][0] * kv[1][1])
tot = [0.0, 0.0]
for (op, key), (us, n) in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 70]:
    tf = ''
    if 'gemm' in op and len(key) >= 3:
        tf = '%5.0f TF' % (2.0 * key[0] * key[1] * key[2] / us / 1e6)
    other = ''
    if b is not None and (op, key) in b:
        other = '   | %8.1f us' % (b[(op, key)][0] * b[(op, key)][1])
        tot[1] += b[(op, key)][0] * b[(op, key)][1]
    tot[0] += us * n
    print('%8.1f us  %-22s %-30s x%.0f %s%s' % (us * n, op, list(key), n, tf, other))
print('sum of the rows shown: %.1f us%s' % (tot[0], ('   | %.1f us' % tot[1]) if b else ''))
