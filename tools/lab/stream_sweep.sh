#!/bin/bash
# LAB: stream-K grid size / slab count of the narrow row GEMMs against ms/step (a slab costs its producer a 5.5 MB write and
# its LayerNorm consumer a 5.5 MB read).  gpurun -- bash tools/lab/stream_sweep.sh
cd "$GRAFT_REPO_ROOT/point_dae_amd/csrc" && touch rows_gemm.hip && make -j8 FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DPDAE_LAB_PLAN" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
run() { echo "$1: $(env $1 python bench.py --no-cpu-baseline --no-also --no-tvis-table --probe-steps 0 --steps 40 2>/dev/null | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"; }
for rep in 1 2; do
  run "X=0"
  run "PDAE_STREAM_SMAX=3"
  run "PDAE_STREAM_SMAX=2"
  run "PDAE_STREAM_PMAX=768"
  run "PDAE_STREAM_PMAX=512"
  run "PDAE_STREAM_PMAX=768 PDAE_STREAM_SMAX=3"
done
