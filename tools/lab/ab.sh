#!/bin/bash
# A/B of one environment switch on the default bench, interleaved rounds in separate processes on ONE box:
#   bash tools/lab/ab.sh PDAE_WGRAD_SIDE 0 1
cd "$GRAFT_REPO_ROOT"
VAR=$1; A=$2; B=$3; ROUNDS=${ROUNDS:-2}
for round in $(seq 1 $ROUNDS); do
  for v in $A $B; do
    line=$(env $VAR=$v python bench.py --no-cpu-baseline --no-also --no-tvis-table --probe-steps 0 --steps 40 --warmup 10 2>/dev/null | grep '"metric"' | tail -1)
    echo "$VAR=$v $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>/dev/null)"
  done
done
