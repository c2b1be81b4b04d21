#!/bin/bash
# Every profile of the round in one GPU call; each step has its own time limit and the chain stops at the first failure.
# Results under gpurun_out/<tag>*; tools/make_profiles.py copies the summaries into profiles/.
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
# (1) the bench line, un-profiled
timeout -k 10 400 python3 bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
# (2) kernel tables: one stream (every kernel alone: its duration means something) and the default two streams
MIS_WGRAD_STREAM=0 bash tools/prof_bench.sh r3s > gpurun_out/r3s_table.txt 2>&1
bash tools/prof_bench.sh r3d > gpurun_out/r3d_table.txt 2>&1
# (3) L2 memory-side traffic, one counter per pass
mkdir -p gpurun_out/r3s/pmc && MIS_WGRAD_STREAM=0 bash tools/pmc_bench.sh $R/gpurun_out/r3s/pmc
# (4) matrix-core busy counters
MIS_WGRAD_STREAM=0 bash tools/pmc_mfma_bench.sh $R/gpurun_out/r3s/mfma > gpurun_out/r3s_mfma.txt 2>&1
# (5) the conv layers one by one
cd $R && timeout -k 10 200 python3 tools/bench_conv.py --lazy > gpurun_out/r3_conv_layers_isolated.txt 2>/dev/null
timeout -k 10 200 python3 tools/bench_bn.py > gpurun_out/r3_streaming_kernels_isolated.txt 2>/dev/null
echo done
