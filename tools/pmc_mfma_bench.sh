#!/bin/bash
# Matrix-core busy counters over two steps of the default bench command:
#   bash tools/pmc_mfma_bench.sh <outdir>   -> <outdir>/mfma.csv
OUT=$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $OUT/pass -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pass.log 2>&1 || { echo "pass failed"; tail -5 $OUT/pass.log; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/rocpd_mfma.py $(find $OUT/pass -name "*.db" | head -1) > $OUT/mfma.csv
cat $OUT/mfma.csv
