#!/bin/bash
set -e
var=$1; shift
for rep in 1 2; do
for v in "$@"; do
  env $var=$v timeout -k 10 120 python bench.py --force-dist --steps 200 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/abd_${var}_$v.json 2> gpurun_out/abd_${var}_$v.err || { echo "$var=$v failed"; tail -3 gpurun_out/abd_${var}_$v.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/abd_${var}_$v.json').read().strip().splitlines()[-1]); print('dist $var=$v rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
done
done
