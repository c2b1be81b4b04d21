#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (CSV).
usage: trace_gaps.py p_kernel_trace.csv [first_fraction_to_skip]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = ev[int(len(ev) * skip):]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = []
last_end = ev[0][1]
for s, e, n in ev[1:]:
    gaps.append((s - last_end, n))
    last_end = max(last_end, e)
pos = [g for g, _ in gaps if g > 0]
print(f"kernels {len(ev)} span {span/1e3:.0f} us busy {busy/1e3:.0f} us idle {sum(pos)/1e3:.0f} us "
      f"({100*sum(pos)/span:.1f} %), overlapped launches {sum(1 for g,_ in gaps if g <= 0)}")
pos.sort()
print("gap ns: median", pos[len(pos)//2], "p90", pos[int(len(pos)*.9)], "max", pos[-1])
by = {}
for g, n in gaps:
    if g > 0:
        k = n.split("(")[0][-60:]
        by.setdefault(k, [0, 0])
        by[k][0] += g
        by[k][1] += 1
for k, (t, c) in sorted(by.items(), key=lambda x: -x[1][0])[:12]:
    print(f"  before {k:60s} total {t/1e3:8.0f} us  n {c:5d}  avg {t/c:7.0f} ns")
