#!/bin/bash
# Kernel-trace profile of the default bench command: bash tools/prof_bench.sh <tag> [bench args]
# -> gpurun_out/<tag>/bench.json and gpurun_out/<tag>/prof/p_kernel_stats.csv
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/bench.err
echo "rc=$?"
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("$OUT/prof/p_kernel_stats.csv")))
steps=28
tot=0
for r in rows[:45]:
    n=re.sub(r"^void ","",r["Name"]).replace("(anonymous namespace)::","")[:90]
    c=int(r["Calls"]); t=float(r["TotalDurationNs"])/1e3
    tot+=t/steps
    print(f"{n:90s} {c/steps:6.1f}/step {t/c:8.2f} us avg {t/steps:8.1f} us/step")
print("sum (top rows) us/step", round(tot,1))
PY
