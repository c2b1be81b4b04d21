#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing.
usage: asm_blocks.py file.s <kernel-name-substring> [min_instrs]"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().endswith(l.split(":")[0] + ":") is False and ":" in l and name in l.split(":")[0])
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur, label = [], Counter(), "entry"
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((label, cur))
            cur, label = Counter(), m.group(1)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        cur[classify(op)] += 1
        if op.startswith("v_pk_"):
            cur["(pk)"] += 1
    blocks.append((label, cur))
    tot = Counter()
    for label, c in blocks:
        tot.update(c)
        n = sum(v for k, v in c.items() if not k.startswith("("))
        if n >= min_n:
            print(f"{label:12s} n={n:5d}  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))
    print("TOTAL", dict(tot))
    for l in lines[end:end + 60]:
        if any(k in l for k in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize", "NumSgprs")):
            print(l.strip())


if __name__ == "__main__":
    main()
