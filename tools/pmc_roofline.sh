#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes) and matrix-pipe
# counters of the dominant kernels (decoder conv3 forward / data-gradient / weight-gradient), plus
# kernel-trace stats of the full bench.   usage: tools/pmc_roofline.sh <outdir-under-repo>
OUT=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU"; do
  T=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$OUT/$T -- python3 $R/tools/run_micro_case.py dec3_up dec3_down dec3_wgrad > $R/$OUT/$T.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/$OUT/bench.log 2>&1
