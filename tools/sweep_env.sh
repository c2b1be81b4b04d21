#!/bin/bash
# Bench value under settings of one environment knob:  tools/sweep_env.sh NAME "v1 v2 ..." [bench flags]
name=$1; vals=$2; shift 2
for v in $vals; do
  env $name=$v timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-roofline --steps 40 "$@" 2>/dev/null > /tmp/sw.json || { echo "$name=$v failed"; continue; }
  python3 -c "import json; d=json.load(open('/tmp/sw.json')); print('$name=$v', d['value'], d['ms_per_step'], d['final_loss'], flush=True)"
done
