#!/bin/bash
# A/B/C.. of one environment switch on the default bench, interleaved rounds in separate processes on ONE box:
#   bash tools/lab/abn.sh PDAE_LIB point_dae_amd/libpdae_hip.so tools/lab/lab_sc1.so tools/lab/lab_prio.so
# ARGS="--workload cfg2" STEPS=10 WARMUP=4: the same on another workload
cd "$GRAFT_REPO_ROOT"
VAR=$1; shift; ROUNDS=${ROUNDS:-2}; STEPS=${STEPS:-40}
for round in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    line=$(env $VAR=$v python bench.py --no-cpu-baseline --no-also --no-tvis-table --no-calibration --probe-steps 0 --steps $STEPS --warmup ${WARMUP:-10} $ARGS 2>/dev/null | grep '"metric"' | tail -1)
    echo "$VAR=$v $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>/dev/null)"
  done
done
