#!/bin/bash
# Two copies of tools/up_diag.py on one GPU at the same time, per configuration; output under gpurun_out/.
mkdir -p gpurun_out
for cfg in "bf16 32 256" "f32 32 256" "bf16 128 64"; do
  set -- $cfg
  timeout -k 10 150 python3 tools/up_diag.py 1200 $1 $2 $3 > gpurun_out/updiag_${1}_${2}_a.txt 2>&1 &
  p1=$!
  timeout -k 10 150 python3 tools/up_diag.py 1200 $1 $2 $3 > gpurun_out/updiag_${1}_${2}_b.txt 2>&1 &
  p2=$!
  wait $p1 || exit 1
  wait $p2 || exit 1
  tail -1 gpurun_out/updiag_${1}_${2}_a.txt gpurun_out/updiag_${1}_${2}_b.txt
done
