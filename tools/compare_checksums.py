#!/usr/bin/env python3
"""First probed tensor whose checksum differs between two runs of tools/probe_step.py (same seed, so every tensor
should match bit for bit):  python tools/compare_checksums.py quiet.json contended.json"""
import json
import sys

a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
if "losses" in a and "losses" in b:
    d = [i for i, (u, v) in enumerate(zip(a["losses"], b["losses"])) if u != v]
    print("losses:", "identical in all %d steps" % len(a["losses"]) if not d else "first differ at step %d (%s vs %s)" % (
        d[0], float.fromhex(a["losses"][d[0]]), float.fromhex(b["losses"][d[0]])))
n = min(len(a["sums"]), len(b["sums"]))
for i in range(n):
    if a["labels"][i] != b["labels"][i]:
        print("label sequence differs at", i, a["labels"][i], b["labels"][i])
        break
    if a["sums"][i] != b["sums"][i] and "wgrad_raw" not in a["labels"][i][1]:
        print("first difference at probe", i, a["labels"][i])
        for j in range(max(0, i - 5), min(n, i + 6)):
            print("   ", j, a["labels"][j], "same" if a["sums"][j] == b["sums"][j] else "DIFFERENT")
        break
else:
    print("all", n, "checksums equal (weight-gradient probes skipped: their reduction is deferred)")
