#!/bin/bash
# Two (or N) independent single-GPU bench processes on the same GPU at the same time: do they disturb each
# other's RESULTS?  (They have nothing in common but the hardware.)   tools/two_procs.sh [trials] [procs] [steps]
trials=${1:-3}; procs=${2:-2}; steps=${3:-150}
for t in $(seq 1 $trials); do
  pids=""
  for p in $(seq 1 $procs); do
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-roofline --steps $steps > /tmp/tp_$p.json 2> /tmp/tp_$p.err &
    pids="$pids $!"
  done
  for pid in $pids; do wait $pid; done
  for p in $(seq 1 $procs); do
    if [ -s /tmp/tp_$p.json ]; then python3 -c "import json; d=json.load(open('/tmp/tp_$p.json')); print('trial $t proc $p', d['value'], d['final_loss'])"
    else echo "trial $t proc $p FAILED: $(grep 'bench.py:' /tmp/tp_$p.err | cut -c1-120)"; fi
  done
done
