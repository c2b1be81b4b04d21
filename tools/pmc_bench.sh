#!/bin/bash
# HBM traffic of a bench step from the L2 memory-side counters: one rocprofv3 run per counter
# (FETCH_SIZE and WRITE_SIZE do not fit one pass).  usage: pmc_bench.sh <outdir>
set -e
OUT=$1
cd /tmp && export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $P -d $OUT/pass$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; exit 1; }
done
echo ok
