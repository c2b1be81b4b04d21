#!/bin/bash
# Two tools/bench_aunet.py processes on the same GPU at once, then one alone: the final losses must agree
# (every reduction is fixed-order).   tools/two_procs_model.sh [a_unet|rdam_unet]
model=${1:-a_unet}
run() { timeout -k 10 300 python3 tools/bench_aunet.py --model $model 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['final_loss'])"; }
run alone
run twin_a & run twin_b; wait
