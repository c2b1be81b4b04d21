#!/bin/bash
# Matrix-core busy counters of the conv kernels, per layer of the benchmark network:
#   bash tools/pmc_mfma.sh <outdir> [layer-substring]
# Two rocprofv3 --pmc passes over tools/bench_conv.py (one kernel launch per layer and direction
# is enough for counters), results as rocpd sqlite files under <outdir>/passN, summary printed by
# tools/rocpd_mfma.py.
set -e
OUT=$1; L=${2:-}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
P2="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $P -d $OUT/pass$i -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py ${L:+--only $L} --lazy --iters 2 > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; }
done
python3 $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $OUT/pass1/*/*.db $OUT/pass2/*/*.db 2>/dev/null || python3 $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $OUT -name "*.db")
