#!/bin/bash
# Kernel-trace profile of tools/bench_aunet.py: bash tools/prof_model.sh <tag> <a_unet|rdam_unet> [steps]
# -> gpurun_out/<tag>/prof/p_kernel_stats.csv and a per-step table on stdout
TAG=$1; MODEL=$2; STEPS=${3:-5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_aunet.py --model $MODEL --steps $STEPS --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
echo "rc=$?"
cat $OUT/bench.json
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("$OUT/prof/p_kernel_stats.csv")))
steps=$STEPS+3
tot=0
for r in rows[:40]:
    n=re.sub(r"^void ","",r["Name"]).replace("(anonymous namespace)::","")[:100]
    c=int(r["Calls"]); t=float(r["TotalDurationNs"])/1e3
    tot+=t/steps
    print(f"{n:100s} {c/steps:6.1f}/step {t/c:8.2f} us avg {t/steps:8.1f} us/step")
print("sum (top rows) us/step", round(tot,1))
PY
