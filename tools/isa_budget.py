#!/usr/bin/env python3
"""Static instruction budget of a kernel from the compiler's ISA listing: per basic block the
number of VALU / MFMA / SALU / LDS / vector-memory / wait instructions, blocks ranked by size, and
the loop structure (a block that branches backwards closes a loop).

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o /tmp/sa_mlp.s backtoreality_amd/csrc/sa_mlp.hip
    tools/isa_budget.py /tmp/sa_mlp.s 'sa_bwd_fused_kernelILi8ELi1ELb0' > profiles/<tag>_isa_budget.md
"""
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(pat), l))
    name = lines[start].split(":")[0]
    blocks, cur, order = {}, "entry", ["entry"]
    blocks[cur] = []
    for l in lines[start + 1:]:
        if "s_endpgm" in l:
            blocks[cur].append("s_endpgm")
            break
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if m and not l.strip().startswith((";", ".")):
            blocks[cur].append(l.strip())
    idx = {b: i for i, b in enumerate(order)}
    kinds = ("valu", "mfma", "salu", "lds", "vmem", "wait")
    tot = dict.fromkeys(kinds + ("other",), 0)
    rows = []
    for b in order:
        c = dict.fromkeys(kinds + ("other",), 0)
        back = []
        for ins in blocks[b]:
            op = ins.split()[0]
            c[classify(op)] += 1
            m = re.match(r"s_cbranch\w*\s+(\.LBB[0-9_]+)|s_branch\s+(\.LBB[0-9_]+)", ins)
            if m:
                t = m.group(1) or m.group(2)
                if t in idx and idx[t] <= idx[b]:
                    back.append(t)
        for k in tot:
            tot[k] += c[k]
        rows.append((b, c, back, len(blocks[b])))
    print("Static ISA budget of `%s`\n" % name)
    print("whole kernel: " + ", ".join("%s %d" % (k, tot[k]) for k in kinds) + "\n")
    print("| block | instructions | VALU | MFMA | SALU | LDS | VMEM | wait / barrier | closes a loop to |")
    print("|---|---|---|---|---|---|---|---|---|")
    for b, c, back, n in sorted(rows, key=lambda r: -r[3])[:14]:
        print("| %s | %d | %d | %d | %d | %d | %d | %d | %s |" % (
            b, n, c["valu"], c["mfma"], c["salu"], c["lds"], c["vmem"], c["wait"],
            ", ".join(back) or "-"))


if __name__ == "__main__":
    main()
