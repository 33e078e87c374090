#!/bin/bash
# launches per step of the default bench under one environment switch:  bash tools/lab/count_launches.sh PDAE_GLUE 0 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  OUT=gpurun_out/count_$v; rm -rf $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --no-cpu-baseline --no-also --no-tvis-table --no-calibration --probe-steps 0 --steps 20 --warmup 5 > $OUT.log 2>&1
  t=$(find $OUT -name '*kernel_trace.csv' | head -1)
  TS_MIN=0 python tools/trace_summary.py "$t" 20 > gpurun_out/count_summary_$v.txt
  echo "$VAR=$v $(sed -n 2,3p gpurun_out/count_summary_$v.txt | tr '\n' ' ')"
  rm -rf $OUT
done
