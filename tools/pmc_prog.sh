#!/bin/bash
# SQ counters of every kernel of a python tool: bash tools/pmc_prog.sh <outdir> <tool.py> [args] ; prints the kernels that match $PMC_GREP
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P -d $OUT/pass$i -o p -- python3 $GRAFT_REPO_ROOT/tools/"$@" > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; }
done
python3 $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $OUT -name "*.db") | grep -A 17 "${PMC_GREP:-kernel}"
