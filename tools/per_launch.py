"""Per-launch durations (us) of the kernels whose name contains one of the given keys, for the last step of a
rocprofv3 kernel trace: python tools/per_launch.py <kernel_trace.csv> <steps in the trace> key [key ...]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
d = collections.defaultdict(list)
for r in rows:
    for key in sys.argv[3:]:
        if key in r["Kernel_Name"]:
            d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    per = len(v) // steps
    print(k, per, " ".join(f"{t:.0f}" for t in v[-per:]))
