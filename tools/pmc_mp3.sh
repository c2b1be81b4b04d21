#!/bin/bash
# SQ / cache counters of the 3x3 max-pool kernels: bash tools/pmc_mp3.sh <outdir> [bench_mp3 args]
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU TCC_HIT_sum TCC_MISS_sum"
P3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_REQ_sum"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $P -d $OUT/pass$i -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_mp3.py "$@" --iters 2 > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; }
done
python3 $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $OUT -name "*.db") | grep -A 12 maxpool3
