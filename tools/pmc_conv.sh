#!/bin/bash
# PMC passes over the conv kernels of one layer (usage: pmc_conv.sh <layer-substring> <outdir>)
# one rocprofv3 run per counter set; results are rocpd sqlite files under <outdir>/passN
set -e
L=$1; OUT=$2
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA"
P3="FETCH_SIZE TCC_HIT_sum"
P4="WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P -d $OUT/pass$i -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --only $L --lazy --iters 3 > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; exit 1; }
done
echo ok
