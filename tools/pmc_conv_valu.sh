#!/bin/bash
# Share of the SIMD cycles that the conv kernels of one layer spend issuing vector-ALU instructions:
#   bash tools/pmc_conv_valu.sh <outdir> <layer> ...   (one rocprofv3 --pmc pass per layer over tools/bench_conv.py)
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -d $OUT/$L -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --only $L --lazy --iters 2 > $OUT/$L.log 2>&1 || { echo "$L failed"; tail -3 $OUT/$L.log; }
  echo "== $L"
  python3 $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $OUT/$L -name "*.db") | grep -A 10 "conv3x3" | grep -v "^--"
done
