#!/bin/bash
# SQ counters of one lab GEMM launch series: bash tools/lab/pmc_lab.sh <tag>   (env SHAPES / VARIANTS as rows3_lab.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_lab_$1; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $OUT/p1 -o pmc -- python tools/lab/rows3_lab.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA \
  --kernel-trace --output-format csv -d $OUT/p2 -o pmc -- python tools/lab/rows3_lab.py > $OUT/p2.log 2>&1
python tools/lab/pmc_lab.py $OUT/p1 $OUT/p2 > $OUT/summary.txt 2>&1
rm -rf $OUT/p1 $OUT/p2
cat $OUT/summary.txt
