#!/bin/bash
# Two training-step processes on one GPU, each repeating every bilinear forward launch and comparing the two results.
# MISAMD_LIBRARY=$PWD/bin_ab/libupdbg.so (tools/build_variant.sh updbg pool_upsample.hip -DMIS_UP_DEBUG=1) adds in-kernel checks.
mkdir -p gpurun_out
PROBE_UP_TWICE=1 PROBE_UP_WHERE=1 PROBE_FINAL_ONLY=1 timeout -k 10 200 python3 tools/probe_step.py 40 > gpurun_out/uprepro_a.txt 2>&1 &
p1=$!
PROBE_UP_TWICE=1 PROBE_UP_WHERE=1 PROBE_FINAL_ONLY=1 timeout -k 10 200 python3 tools/probe_step.py 40 > gpurun_out/uprepro_b.txt 2>&1 &
p2=$!
wait $p1 || exit 1
wait $p2 || exit 1
grep -h "up:\|WHERE\|up_dbg" gpurun_out/uprepro_a.txt gpurun_out/uprepro_b.txt | cut -c1-1500
