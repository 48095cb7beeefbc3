#!/bin/bash
# rocprofv3 --pmc passes (counters only, one pass per group, as MI355X_MICROARCH.md prescribes) over one python
# command, then a per-kernel summary.   usage: tools/pmc.sh <tag> <kernel-name-regex> <script.py> [args...]
# -> gpurun_out/pmc_<tag>/summary.txt
TAG=$1; PAT=$2; shift; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="FETCH_SIZE GRBM_GUI_ACTIVE"
P4="WRITE_SIZE TCC_HIT TCC_MISS"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$i -- python3 $R/"$@" > $OUT/pass$i.log 2>&1
done
python3 $R/tools/pmc_summary.py "$PAT" /tmp/pmc_${TAG}_* > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
