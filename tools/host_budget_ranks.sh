#!/bin/bash
# Host budget of the eager data-parallel step with N rank processes on one host at the same time: N copies of
# tools/host_step_time.py --dist (one-rank RCCL groups, tiny input: the GPU work is negligible, wall time per step is
# host time), each with 8 hardware queues as bench.py sets them.  Compare with the GPU time of the real step.
# usage: tools/host_budget_ranks.sh [N=8]   (output: gpurun_out/hostbudget_<N>_<rank>.txt)
n=${1:-8}
mkdir -p gpurun_out
pids=""
for r in $(seq 0 $((n - 1))); do
  GPU_MAX_HW_QUEUES=8 MASTER_PORT=$((29700 + r)) timeout -k 10 170 python3 tools/host_step_time.py --dist > gpurun_out/hostbudget_${n}_$r.txt 2>&1 &
  pids="$pids $!"
done
rc=0
for p in $pids; do wait $p || rc=1; done
grep -h "ms per step" gpurun_out/hostbudget_${n}_*.txt
exit $rc
