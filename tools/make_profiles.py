#!/usr/bin/env python3
"""Copies one profiling run (gpurun_out/<tag>: bench.json, prof/p_kernel_stats.csv, pmc/pass{1,2})
into profiles/ under this round's names.   python tools/make_profiles.py v9 r1 [steps_in_profile=28]"""
import csv
import os
import re
import shutil
import subprocess
import sys

tag, rnd = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 28
src = os.path.join("gpurun_out", tag)
shutil.copy(os.path.join(src, "bench.json"), f"profiles/{rnd}_bench_{tag}.json")
if os.path.exists(os.path.join(src, "pmc/pass1/p_results.db")):       # (a run without counter passes keeps the other run's file)
    with open(f"profiles/{rnd}_pmc_traffic.json", "w") as f:
        subprocess.check_call([sys.executable, "tools/pmc_traffic.py", os.path.join(src, "pmc/pass1/p_results.db"),
                               os.path.join(src, "pmc/pass2/p_results.db")], stdout=f)
rows = list(csv.DictReader(open(os.path.join(src, "prof/p_kernel_stats.csv"))))
out, tot = [], 0.0
for r in rows:
    name = re.sub(r"^void ", "", r["Name"]).replace("(anonymous namespace)::", "")
    calls, t = int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3
    out.append((name[:160], calls, round(t, 1), round(t / calls, 2), round(float(r["MinNs"]) / 1e3, 2),
                round(float(r["MaxNs"]) / 1e3, 2), r["Percentage"], round(t / steps, 1), round(calls / steps, 1)))
    tot += t / steps
with open(f"profiles/{rnd}_rocprofv3_kernel_stats_{tag}.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent", "us_per_step", "calls_per_step"])
    w.writerows(out)
    w.writerow([f"TOTAL ({steps} steps: warm-up + timed + 3 instrumented)", "", "", "", "", "", "", round(tot, 1), ""])
fam = {"weight gradient": 0.0, "conv forward/data gradient": 0.0, "bn_relu_bwd_dz": 0.0, "pool": 0.0, "upsample": 0.0, "1x1": 0.0}
for o in out:
    n = o[0]
    if "wgrad" in n: fam["weight gradient"] += o[7]
    elif "conv3x3" in n: fam["conv forward/data gradient"] += o[7]
    elif "bn_relu_bwd_dz" in n: fam["bn_relu_bwd_dz"] += o[7]
    elif "maxpool" in n: fam["pool"] += o[7]
    elif "up_" in n: fam["upsample"] += o[7]
    elif "conv1x1" in n: fam["1x1"] += o[7]
print("kernel sum per step (us):", round(tot, 1), {k: round(v, 1) for k, v in fam.items()})
