#!/bin/bash
# Starts the two ranks of tests/dp_worker.py by hand (debugging aid): bash tools/run_dp_pair.sh OUT_DIR
out=${1:-gpurun_out/dp_pair}
mkdir -p "$out"
MIS_DP_DEBUG=25 timeout -k 5 60 python -u tests/dp_worker.py 0 2 29655 "$out" > "$out/r0.log" 2>&1 &
p0=$!
MIS_DP_DEBUG=25 timeout -k 5 60 python -u tests/dp_worker.py 1 2 29655 "$out" > "$out/r1.log" 2>&1 &
p1=$!
wait $p0; echo "rank0 rc=$?"
wait $p1; echo "rank1 rc=$?"
