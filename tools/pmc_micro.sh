#!/bin/bash
# PMC passes over tools/microbench.py (each pass = its own rocprofv3 run, counters only).
# usage: tools/pmc_micro.sh <outdir> <case> [case ...]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="FETCH_SIZE GRBM_GUI_ACTIVE"
P4="WRITE_SIZE TCC_HIT TCC_MISS"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/$OUT/pass$i -- python3 $R/tools/microbench.py "$@" > $R/$OUT/pass$i.log 2>&1
done
ls -R $R/$OUT | head -30
