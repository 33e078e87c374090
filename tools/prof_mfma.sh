#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/round; mkdir -p $OUT; rm -rf $OUT/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o pmc -- python bench.py --no-cpu-baseline --no-also --no-calibration --steps 3 --warmup 1 --no-tvis-table --probe-steps 0 > $OUT/pmc_mfma.log 2>&1
python tools/mfma_summary.py $OUT/pmc_mfma > $OUT/mfma.json
rm -rf $OUT/pmc_mfma
